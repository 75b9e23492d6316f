#!/usr/bin/env python3
"""CPU only: the C oracle against the LITERAL Python restatement of the reference, on many small random inputs with few
letters (score ties, dense neighbourhoods) -- greedy (LimitedGreedySequenceClusterer incl. its three NullPointerException
cases) and clinkage (ClinkageSequenceClusterer + CachedClusterScorer + DynamicMatrix + HashSet parts, incl. the inputs on
which the reference's chain returns to a stacked cluster: the literal form must then throw NoSuchElementException or return a
sequence twice).  The GPU fuzzers compare the product with the C oracle; this one keeps the C oracle honest.
Usage: python tests/tools/fuzz_oracles.py [trials] [seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402
from oracle import hammock_oracle as po  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    mats = {k: np.asarray(v, dtype=np.int32) for k, v in json.load(fh)["matrices"].items()}
names = sorted(mats)
LETTERS = "ARNDCQEGHILKMFPSTWYV"
rng = np.random.default_rng(seed)
greedy_ok = greedy_crash = clink_ok = clink_stale = 0
for trial in range(trials):
    M = mats[names[int(rng.integers(len(names)))]]
    if not (M == M.T).all():
        M = mats["blosum62"]
    n = int(rng.integers(2, 70))
    alphabet = int(rng.integers(2, 7))
    lo = int(rng.integers(5, 10)); hi = lo + int(rng.integers(0, 4))
    n = min(n, alphabet ** lo // 2)      # enough distinct peptides exist
    seen, peps = set(), []
    while len(peps) < n:
        q = rng.integers(0, alphabet, int(rng.integers(lo, hi + 1))).astype(np.uint8)
        if q.tobytes() not in seen:
            seen.add(q.tobytes()); peps.append(q)
    sizes = rng.integers(1, 4, n).astype(np.int32) if trial % 2 else None
    strings = ["".join(LETTERS[c] for c in q) for q in peps]
    L = np.array([len(q) for q in peps])
    X = int(min(rng.integers(0, 4), L.min() - 1)); p = -int(rng.integers(0, 3))
    res, off = c_oracle.pack(peps)
    # ---- clinkage ----
    thr = int(rng.integers(5, 30))
    st, cid, order, rank, stats = c_oracle.clinkage_cluster(M, res, off, sizes, X, p, thr, 1 + trial % 3)
    seqs = [po.UniqueSequence(s, {"no_label": int(sizes[k]) if sizes is not None else 1}) for k, s in enumerate(strings)]
    cl = po.ClinkageSequenceClusterer(po.ShiftedScorer(M.tolist(), p, X), thr, size_limit=1, n_threads=1 + trial % 4)
    try:
        result = cl.cluster(seqs)
        memberships = [id(s) for c in result for s in c.sequences]
        twice = len(memberships) != len(set(memberships))
    except po.NoSuchElement:
        result, twice = None, True
    info = {"trial": trial, "n": n, "alphabet": alphabet, "X": X, "p": p, "thr": thr}
    if st == c_oracle.HMO_ERR_REFERENCE_WOULD_CRASH:
        if not twice:
            print(json.dumps({"FAIL": "C oracle flags a stacked cluster, the literal form returns a clean clustering", **info})); sys.exit(1)
        clink_stale += 1
    else:
        if twice or st != 0:
            print(json.dumps({"FAIL": "the literal form throws / repeats a sequence, the C oracle does not flag it", "st": st, **info})); sys.exit(1)
        index_of = {id(s): k for k, s in enumerate(seqs)}
        pcid = np.full(n, -1, dtype=np.int32); prank = np.full(n, -1, dtype=np.int32)
        for c in result:
            for pos, s in enumerate(c.sequences):
                pcid[index_of[id(s)]] = c.id; prank[index_of[id(s)]] = pos
        if not (np.array_equal(cid, pcid) and order.tolist() == [c.id for c in result] and np.array_equal(rank, prank)):
            print(json.dumps({"FAIL": "clinkage: C oracle and literal form differ", **info})); sys.exit(1)
        clink_ok += 1
    # ---- greedy ----
    thr = int(rng.integers(5, 30)); maxc = int(rng.integers(1, max(2, n // 2)))
    st, cid, order, gstats = c_oracle.greedy_cluster(M, res, off, sizes, 0, X, p, thr, maxc, 1 + trial % 3)
    seqs = [po.UniqueSequence(s, {"no_label": int(sizes[k]) if sizes is not None else 1}) for k, s in enumerate(strings)]
    gl = po.LimitedGreedySequenceClusterer(po.ShiftedScorer(M.tolist(), p, X), thr, maxc, n_threads=1 + trial % 4)
    info = {"trial": trial, "n": n, "alphabet": alphabet, "X": X, "p": p, "thr": thr, "maxc": maxc}
    try:
        result = gl.cluster(seqs)
    except po.ReferenceWouldCrash as e:
        if st != c_oracle.HMO_ERR_REFERENCE_WOULD_CRASH or (gstats.crash_case, gstats.crash_index) != (e.case, e.index):
            print(json.dumps({"FAIL": "greedy: crash case differs", **info})); sys.exit(1)
        greedy_crash += 1
        continue
    index_of = {id(s): k for k, s in enumerate(seqs)}
    pcid = np.full(n, -1, dtype=np.int32); prank = np.full(n, -1, dtype=np.int32)
    for c in result:
        for pos, s in enumerate(c.sequences):
            pcid[index_of[id(s)]] = c.id; prank[index_of[id(s)]] = pos
    if st != 0 or not (np.array_equal(cid, pcid) and order.tolist() == [c.id for c in result] and np.array_equal(gstats.member_rank, prank)):
        print(json.dumps({"FAIL": "greedy: C oracle and literal form differ", "st": st, **info})); sys.exit(1)
    greedy_ok += 1
    if trial % 10 == 9:
        print(f"trial {trial + 1}/{trials}", flush=True)
print(json.dumps({"trials": trials, "seed": seed, "clinkage_identical": clink_ok, "clinkage_stale_stack": clink_stale,
                  "greedy_identical": greedy_ok, "greedy_crash_parity": greedy_crash}))
