#!/usr/bin/env python3
"""CPU fuzz of the windowed phase 1 of the host merge (hmk_greedy.cpp): dense, low-complexity inputs in the reference's default
order (neighbouring rows are neighbours in the graph: commits patch or invalidate the scans of the window's later rows all the
time), random cluster limits and thresholds, several thread counts and window sizes -- ids, list order, member order and the
phase-1 counters against the oracle's literal sequential loop (tests/test_host_greedy.py::run_both).

    python tests/tools/fuzz_phase1_windows.py [trials=120] [seed=1]
"""
import faulthandler
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import c_oracle  # noqa: E402
import test_host_greedy as T  # noqa: E402
import hammock_amd  # noqa: E402

faulthandler.dump_traceback_later(600, exit=True)   # a hang shows where
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c_oracle.lib()
coracle = c_oracle
with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
rng = np.random.default_rng(seed)
done = crash = 0
for trial in range(trials):
    n = int(rng.integers(50, 1600))
    alphabet = int(rng.integers(3, 7))   # (2 letters: every pair is a neighbour and the literal oracle takes minutes)
    lo = int(rng.integers(7, 13))
    hi = int(rng.integers(lo, 15))
    peps = T.random_peptides(rng, n, lo, hi, alphabet=alphabet)
    sizes = rng.integers(1, 4, size=len(peps)).astype(np.int32) if rng.random() < 0.5 else None
    res, off = coracle.pack(peps)
    order = ["size", "alphabetic", "input"][int(rng.integers(0, 3))]
    perm = coracle.sort_order(res, off, sizes, order) if order != "input" else np.arange(len(peps))
    peps = [peps[k] for k in perm]
    if sizes is not None:
        sizes = sizes[perm]
    X = int(rng.integers(0, 4))
    thr = int(rng.integers(8, 40))
    maxc = int(rng.integers(1, max(2, len(peps) // 2)))
    res, off = coracle.pack(peps)
    st, cid, order_o, stats = coracle.greedy_cluster(M, res, off, sizes, 0, X, 0, thr, maxc, 8)
    edges = T.oracle_edges(coracle, M, res, off, X, 0, thr, True)
    np.random.default_rng(trial).shuffle(edges)
    ctx = hammock_amd.Context(M, device=-1)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    for threads, window in ((1, 1), (2, 3), (3, 8), (4, 32), (8, 64), (5, 4096)):
        os.environ["HMK_PHASE1_THREADS"] = str(threads)
        os.environ["HMK_PHASE1_WINDOW"] = str(window)
        info = {"trial": trial, "n": len(peps), "alphabet": alphabet, "order": order, "X": X, "thr": thr, "maxc": maxc, "threads": threads, "window": window}
        if st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH:
            try:
                ctx.greedy_from_edges(edges, True, thr, maxc)
                print(json.dumps({"FAIL": "no crash parity", **info})); sys.exit(1)
            except hammock_amd.ReferenceWouldCrash as exc:
                if (exc.case, exc.index) != (stats.crash_case, stats.crash_index):
                    print(json.dumps({"FAIL": "crash case", **info})); sys.exit(1)
            crash += 1
        else:
            assert st == 0
            gcid, gorder, gstats = ctx.greedy_from_edges(edges, True, thr, maxc)
            ok = (np.array_equal(gcid, cid) and np.array_equal(gorder, order_o) and np.array_equal(ctx.member_rank[:len(cid)], stats.member_rank)
                  and (gstats.phase1_stop_index, gstats.phase1_clusters, gstats.phase1_orphans, gstats.n_multi)
                  == (stats.phase1_stop_index, stats.phase1_clusters, stats.phase1_orphans, stats.n_multi))
            if not ok:
                print(json.dumps({"FAIL": "differs from the oracle", **info})); sys.exit(1)
        done += 1
    faulthandler.cancel_dump_traceback_later()
    faulthandler.dump_traceback_later(600, exit=True)
    if trial % 10 == 9:
        print(f"trial {trial + 1}/{trials}: {done} runs identical to the oracle ({crash} crash-parity)", flush=True)
print(json.dumps({"trials": trials, "seed": seed, "runs": done, "crash_parity": crash, "all_identical": True}))
