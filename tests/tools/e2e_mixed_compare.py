#!/usr/bin/env python3
"""BASELINE config 4a input (10^5 peptides of length 7..20, shift penalty -1) through the greedy path, GPU vs the CPU
restatement, identical membership asserted.  Usage: python tests/tools/e2e_mixed_compare.py [n] [threads]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402
from bench import load_blosum62  # noqa: E402
from oracle import c_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = load_blosum62()
res, off = synth_peptides(1, n, 7, 20)
rng = np.random.default_rng(1)
sizes = np.ones(n, dtype=np.int32)
sizes[::4] = 1 + rng.integers(0, 64, size=len(sizes[::4]))
perm = c_oracle.sort_order(res, off, sizes, "size")
peps = [res[off[k]:off[k + 1]] for k in perm]
sizes = sizes[perm]
res, off = hammock_amd.pack_sequences(peps)
X, p, thr, maxc = 3, -1, 23, int(np.floor(n * 0.025 + 0.5))
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
t0 = time.perf_counter()
try:
    cid, order, gst = ctx.greedy_cluster(X, p, thr, maxc)
    gpu = {"status": 0}
except hammock_amd.ReferenceWouldCrash as e:
    gpu = {"status": "reference_would_crash", "case": e.case, "index": e.index}
t_gpu = time.perf_counter() - t0
t0 = time.perf_counter()
st, ocid, oorder, ost = c_oracle.greedy_cluster(M, res, off, sizes, 0, X, p, thr, maxc, threads)
t_cpu = time.perf_counter() - t0
if st == 0:
    identical = gpu["status"] == 0 and bool(np.array_equal(cid, ocid) and np.array_equal(order, oorder))
else:
    identical = gpu.get("case") == ost.crash_case and gpu.get("index") == ost.crash_index
print(json.dumps({"workload": f"{n} synthetic peptides of length 7..20 with counts, BLOSUM62, X=3, p=-1, thr=23, maxClusters={maxc}",
                  "gpu_cluster_s_first_call": t_gpu, "cpu_port_cluster_s": t_cpu, "cpu_threads": threads, "oracle_status": int(st),
                  "cpu_sequenceScore_calls": int(ost.score_calls_phase1 + ost.score_calls_phase2), "identical": identical}))
assert identical
