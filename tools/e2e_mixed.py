#!/usr/bin/env python3
"""End-to-end greedy clustering of a mixed-length set (BASELINE config 4a input) through the C ABI:
wall time of hmk_greedy_cluster split into plan / scoring / merge, twice (second run re-plans with
another threshold, kernels already loaded)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402
from bench import load_blosum62  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
res, off = synth_peptides(1, n, 7, 20)
M = load_blosum62()
t0 = time.perf_counter()
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off)
t_set = time.perf_counter() - t0
for thr in (23, 24, 23):
    t = time.perf_counter()
    try:
        cid, order, st = ctx.greedy_cluster(3, -1, thr, int(n * 0.025 + 0.5))
        out = {"clusters": int(st.n_multi), "edges": int(st.n_edges), "neighbors_ms": st.neighbors_ms, "merge_ms": st.greedy_ms}
    except hammock_amd.ReferenceWouldCrash as e:
        out = {"reference_would_crash": [e.case, e.index]}
    out.update({"n": n, "thr": thr, "wall_s": time.perf_counter() - t, "kernel_ms": ctx.last_kernel_ms(), "set_sequences_s": t_set})
    print(json.dumps(out), flush=True)
