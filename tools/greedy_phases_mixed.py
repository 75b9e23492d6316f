#!/usr/bin/env python3
"""Per-phase milliseconds of hmk_greedy_cluster on the mixed-length workload of BASELINE config 4a (10^5 peptides of
length 7..20, X = 3, p = -1, thr = 23, maxClusters = 2,500): three consecutive calls of one context."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
res, off = synth_peptides(1, n, 7, 20)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
for call in range(3):
    t = time.perf_counter()
    cid, order, st = ctx.greedy_cluster(3, -1, 23, int(n * 0.025 + 0.5))
    w = (time.perf_counter() - t) * 1e3
    print(json.dumps({"n": n, "len": [7, 20], "call": call, "wall_ms": w, "clusters": int(st.n_multi), **ctx.greedy_phases()}))
