#!/usr/bin/env python3
"""End-to-end figure of SURVEY.md 8(d): wall time of "sort + cluster" (Hammock.java:406-411) on the
BASELINE workload for the GPU path vs the CPU restatement of the reference algorithm, with identical
cluster membership asserted.  Usage: python tests/tools/e2e_compare.py [n] [threads]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
from oracle import c_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = load_blosum62()
res, off = synth_peptides(1, n, 12)
rng = np.random.default_rng(1)
sizes = np.ones(n, dtype=np.int32)
sizes[::4] = 1 + rng.integers(0, 64, size=len(sizes[::4]))   # counts: exercises the size order and the size tie-break
X, p, thr, maxc = 3, 0, 20, int(np.floor(n * 0.025 + 0.5))

ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res[:off[100]], offsets=off[:101])
ctx.neighbors_shifted(X, p, thr)                               # warm the context (module load)

t0 = time.perf_counter()
perm = np.lexsort(tuple(-res.reshape(n, 12)[:, k].astype(np.int64) for k in range(11, -1, -1)) + (-sizes.astype(np.int64),))
t_sort_np = time.perf_counter() - t0
# the reference's order: size desc, then sequence STRING desc (UniqueSequence.java:238-261); the letter
# order differs from the residue-index order, so sort by the decoded letters
letters = np.frombuffer(hammock_amd.AMINO_ACIDS.encode(), dtype=np.uint8)[res.reshape(n, 12)]
t0 = time.perf_counter()
perm = np.lexsort(tuple(-letters[:, k].astype(np.int64) for k in range(11, -1, -1)) + (-sizes.astype(np.int64),))
sres = np.ascontiguousarray(res.reshape(n, 12)[perm].reshape(-1))
ssizes = sizes[perm]
ctx.set_sequences(residues=sres, offsets=off, sizes=ssizes)
cid, order, gstats = ctx.greedy_cluster(X, p, thr, maxc)
t_gpu = time.perf_counter() - t0

t0 = time.perf_counter()
operm = c_oracle.sort_order(res, off, sizes, "size")
assert np.array_equal(operm, perm), "host ordering differs from the oracle's sortSequences"
ores = np.ascontiguousarray(res.reshape(n, 12)[operm].reshape(-1))
st, ocid, oorder, ostats = c_oracle.greedy_cluster(M, ores, off, sizes[operm], 0, X, p, thr, maxc, threads)
t_cpu = time.perf_counter() - t0
assert st == 0
identical = bool(np.array_equal(cid, ocid) and np.array_equal(order, oorder))
calls = int(ostats.score_calls_phase1 + ostats.score_calls_phase2)
print(json.dumps({"workload": f"{n} synthetic 12-mers with counts, BLOSUM62, X=3, p=0, thr=20, maxClusters={maxc}",
                  "gpu_sort_upload_cluster_s": t_gpu, "gpu_score_and_csr_ms": gstats.neighbors_ms,
                  "gpu_merge_ms_incl_wait_for_d2h": gstats.greedy_ms, "cpu_port_sort_cluster_s": t_cpu, "cpu_threads": threads,
                  "cpu_sequenceScore_calls": calls, "cpu_calls_per_s": calls / t_cpu,
                  "pair_space": n * (n - 1) // 2, "speedup_end_to_end": t_cpu / t_gpu,
                  "identical_membership": identical, "clusters": int(gstats.n_multi)}))
assert identical
