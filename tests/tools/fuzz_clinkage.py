#!/usr/bin/env python3
"""Long randomized sweep of hmk_clinkage_cluster against the oracle's ClinkageSequenceClusterer restatement: random sizes
(counts), mixed lengths, thresholds from dense to sparse, shift penalty, families of near-duplicates (big clusters, long
chains, many score ties), single- and two-"device" contexts.  Ids, list order and member order must all be equal.
Usage: python tests/tools/fuzz_clinkage.py [trials] [seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402
from oracle import c_oracle  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    mats = {k: np.asarray(v, dtype=np.int32) for k, v in json.load(fh)["matrices"].items()}
names = sorted(mats)
rng = np.random.default_rng(seed)
merges = crashes = 0
for trial in range(trials):
    M = mats["blosum62"] if trial % 3 else mats[names[int(rng.integers(len(names)))]]
    lo = int(rng.integers(6, 14))
    hi = int(min(32, lo + rng.integers(0, 9)))
    n = int(rng.integers(2, 4000))
    res, off = synth_peptides(int(rng.integers(1, 10 ** 6)), n, lo, hi)
    if trial % 2 == 0:   # families of near-duplicates
        peps = [res[off[k]:off[k + 1]].copy() for k in range(n)]
        for k in range(n // 4, n):
            src = peps[int(rng.integers(0, max(1, n // 4)))].copy()
            for _ in range(int(rng.integers(1, 4))):
                src[int(rng.integers(len(src)))] = rng.integers(0, 20)
            peps[k] = src
        peps = list({bytes(q): q for q in peps}.values())
        n = len(peps)
        res, off = hammock_amd.pack_sequences(peps)
    sizes = None
    if trial % 3 != 1:
        sizes = np.ones(n, dtype=np.int32)
        pick = rng.random(n) < 0.3
        sizes[pick] = 1 + rng.integers(0, 5, int(pick.sum()))
    L = np.diff(off.astype(np.int64))
    thr = int(round(L.mean() * 1.7)) + int(rng.integers(-9, 6))
    X = int(min(max(0, round(L.mean() / 4)), L.min() - 1))
    p = int(rng.choice([0, 0, -1, -2]))
    if os.environ.get("FUZZ_ONLY") and trial != int(os.environ["FUZZ_ONLY"]):   # replay one trial: the draws above keep the sequence
        continue
    st, ocid, oorder, orank, ostats = c_oracle.clinkage_cluster(M, res, off, sizes, X, p, thr, 16)
    ctx = hammock_amd.Context(M, device=[0, 0] if trial % 5 == 4 and not os.environ.get("FUZZ_ONE_DEVICE") else 0)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    if st == c_oracle.HMO_ERR_REFERENCE_WOULD_CRASH:   # the chain returns to a cluster that is still on its stack
        try:
            ctx.clinkage_cluster(X, p, thr)
            print(json.dumps({"FAIL": "no crash on the GPU path", "trial": trial, "n": n, "len": [lo, hi], "X": X, "p": p, "thr": thr}))
            sys.exit(1)
        except hammock_amd.ReferenceWouldCrash:
            crashes += 1
            continue
    assert st == 0, st
    cid, order, stats = ctx.clinkage_cluster(X, p, thr)
    if not (np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:n], orank)
            and stats.merges == ostats.merges):
        print(json.dumps({"FAIL": "clusters differ", "trial": trial, "n": n, "len": [lo, hi], "X": X, "p": p, "thr": thr}))
        sys.exit(1)
    merges += int(stats.merges)
    if trial % 20 == 19:
        print(f"trial {trial + 1}/{trials}: identical, {merges} merges so far", flush=True)
print(json.dumps({"trials": trials, "seed": seed, "identical": trials - crashes, "crash_parity": crashes, "merges": merges}))
