#!/usr/bin/env python3
"""BASELINE config 5 input (10^6 synthetic 12-mers) end to end on ONE GPU: hmk_greedy_cluster
(all 5 x 10^11 pairs scored, CSR on the device, pinned D2H, host merge).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
M = load_blosum62()
t0 = time.perf_counter()
res, off = synth_peptides(1, n, 12)
t_gen = time.perf_counter() - t0
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off)
maxc = int(np.floor(n * 0.025 + 0.5))
t0 = time.perf_counter()
cid, order, st = ctx.greedy_cluster(3, 0, 20, maxc)
t1 = time.perf_counter() - t0
t0 = time.perf_counter()
cid2, order2, st2 = ctx.greedy_cluster(3, 0, 20, maxc)   # second call: buffers already sized
t2 = time.perf_counter() - t0
assert np.array_equal(cid, cid2) and np.array_equal(order, order2)
sizes = np.bincount(np.unique(cid, return_inverse=True)[1])
print(json.dumps({"n": n, "pairs": n * (n - 1) // 2, "edges": int(st.n_edges), "clusters": int(st.n_multi),
                  "result_list": int(st.n_result_clusters), "largest_cluster": int(sizes.max()),
                  "first_call_s": t1, "second_call_s": t2, "score_and_csr_ms": st2.neighbors_ms,
                  "merge_ms_incl_wait_for_d2h": st2.greedy_ms, "phase1_stop_index": int(st.phase1_stop_index),
                  "generate_s": t_gen}))
