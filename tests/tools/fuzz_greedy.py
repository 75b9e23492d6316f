#!/usr/bin/env python3
"""Long randomized sweep of hmk_greedy_cluster against the oracle's literal greedy: random sizes (counts),
mixed lengths, thresholds around the reference default, cluster limits from tiny to large, shift penalty,
asymmetric matrices, including the inputs on which the reference throws (crash parity).
Usage: python tests/tools/fuzz_greedy.py [trials] [seed] [band]     (band: every input large enough for the prepared band, one length)"""
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
BAND = len(sys.argv) > 3 and sys.argv[3] == "band"
with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    blosum62 = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
rng = np.random.default_rng(seed)
crashes = ok_runs = 0
from collections import Counter
paths = Counter()
for trial in range(trials):
    M = blosum62.copy()
    if trial % 5 == 4:
        M += rng.integers(-1, 2, size=(24, 24)).astype(np.int32)      # asymmetric
    lo = int(rng.integers(6, 14))
    hi = int(min(32, lo + rng.integers(0, 9)))
    n = int(rng.integers(300, 5000)) if trial % 4 else int(rng.integers(5000, 30000))   # every fourth input is big enough for the band + device loop
    if BAND:
        n = int(rng.integers(16384, 36000))
        hi = lo if trial % 3 else hi          # mostly one length (a bucket reordered by score bound has no band)
    alphabet_seed = int(rng.integers(1, 10 ** 6))
    res, off = synth_peptides(alphabet_seed, n, lo, hi)
    if trial % 3 == 0:   # families of near-duplicates: dense neighbourhoods, big clusters
        peps = [res[off[k]:off[k + 1]].copy() for k in range(n)]
        for k in range(n // 3, n):
            src = peps[int(rng.integers(0, n // 3))].copy()
            for _ in range(int(rng.integers(1, 4))):
                src[int(rng.integers(len(src)))] = rng.integers(0, 20)
            peps[k] = src
        peps = list({bytes(q): q for q in peps}.values())
        n = len(peps)
        res, off = hammock_amd.pack_sequences(peps)
    sizes = np.ones(n, dtype=np.int32)
    pick = rng.random(n) < 0.3
    sizes[pick] = 1 + rng.integers(0, 50, int(pick.sum()))
    perm = c_oracle.sort_order(res, off, sizes, "size")
    peps = [res[off[k]:off[k + 1]] for k in perm]
    sizes = sizes[perm]
    res, off = hammock_amd.pack_sequences(peps)
    L = np.diff(off.astype(np.int64))
    thr = int(round(L.mean() * 1.7)) + int(rng.integers(-6, 5))
    X = int(min(max(1, round(L.mean() / 4)), L.min() - 1))
    p = int(rng.choice([0, 0, -1, -2]))
    maxc = int(max(1, rng.choice([2, 10, int(n * 0.025) + 1, n // 8 + 1])))
    if BAND:
        maxc = int(max(1, rng.choice([10, int(n * 0.025) + 1, n // 10, n // 5])))
    st, ocid, oorder, ostats = c_oracle.greedy_cluster(M, res, off, sizes, 0, X, p, thr, maxc, 16)
    mode = [None, "device", "host", "host"][trial % 4 if trial % 8 >= 4 else 0]   # half the trials: the default path
    if mode:
        os.environ["HMK_SECOND_LOOP"] = mode
    else:
        os.environ.pop("HMK_SECOND_LOOP", None)
    ctx = hammock_amd.Context(M, device=[0, 0] if trial % 7 == 6 else 0)   # now and then a two-"device" context
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    info = {"trial": trial, "n": n, "len": [lo, hi], "X": X, "p": p, "thr": thr, "maxc": maxc, "second_loop": mode}
    if os.environ.get("FUZZ_VERBOSE_FROM") and trial >= int(os.environ["FUZZ_VERBOSE_FROM"]):
        print(json.dumps({**info, "devices": 2 if trial % 7 == 6 else 1, "oracle_status": int(st)}), flush=True)
    if st == c_oracle.HMO_ERR_REFERENCE_WOULD_CRASH:
        try:
            ctx.greedy_cluster(X, p, thr, maxc)
            print(json.dumps({"FAIL": "no crash on the GPU path", **info}))
            sys.exit(1)
        except hammock_amd.ReferenceWouldCrash as e:
            if (e.case, e.index) != (ostats.crash_case, ostats.crash_index):
                print(json.dumps({"FAIL": "crash case differs", **info}))
                sys.exit(1)
        crashes += 1
    else:
        assert st == 0, st
        try:
            cid, order, gstats = ctx.greedy_cluster(X, p, thr, maxc)
        except Exception as e:   # (not a parity failure: say which input it was)
            print(json.dumps({"FAIL": "exception: " + repr(e), **info, "devices": 2 if trial % 7 == 6 else 1}))
            raise
        if not (np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:n], ostats.member_rank)):
            print(json.dumps({"FAIL": "clusters differ", **info}))
            sys.exit(1)
        ok_runs += 1
        ph = ctx.greedy_phases()
        paths[("device" if ph["loop_rounds"] else "host") + (", prepared band" if ph["band_bytes"] else "")] += 1
    if trial % 20 == 19:
        print(f"trial {trial + 1}/{trials}: {ok_runs} identical clusterings, {crashes} crash parities", flush=True)
print(json.dumps({"trials": trials, "seed": seed, "identical": ok_runs, "crash_parity": crashes, "second_loop_paths": dict(paths)}))
