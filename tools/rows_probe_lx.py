#!/usr/bin/env python3
"""Times the plain neighbour pass of 1e5 uniform L-mers at max shift X for several thresholds (a threshold no pair reaches
= the read phase alone).  Usage: python tools/rows_probe_lx.py L X thr [thr ...]   (N=... in the environment: set size)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

L, X = int(sys.argv[1]), int(sys.argv[2])
thrs = [int(v) for v in sys.argv[3:]]
n = int(os.environ.get("N", "100000"))
res, off = synth_peptides(1, n, L)
dev = torch.device("cuda", 0)
cap = 1 << 28
d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
pairs = n * (n - 1) // 2
cells = L + 2 * X * L - X * (X + 1)
ideal = cells * pairs / (256 * 256 * 2.4e9) * 1e3
out = {"L": L, "X": X, "lds_ideal_ms": round(ideal, 4)}
for thr in thrs:
    ctx = hammock_amd.Context(load_blosum62(), device=0)
    ctx.set_sequences(residues=res, offsets=off)
    # as bench.py measures: 12 settle passes back to back, then 20 timed ones between events on the launch stream, one synchronise
    # (a synchronise after every pass let the GPU idle between passes: every figure 4-8 % slower than the same kernel in bench.py)
    stream = torch.cuda.current_stream(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for k in range(12 + 20):
        if k >= 12:
            evs[k - 12][0].record(stream)
        ctx.neighbors_shifted_dev(X, 0, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        if k >= 12:
            evs[k - 12][1].record(stream)
    torch.cuda.synchronize()
    ms = [0.0] * 4 + [a.elapsed_time(b) for a, b in evs]
    med = float(np.median(ms[4:]))
    out[thr] = {"edges": int(d_counts.sum().item()), "ms_median": round(med, 4), "ms_min": round(float(min(ms[4:])), 4),
                "frac": round(ideal / med, 3), "rows": int(ctx.last_plan().classes_rows)}
print(json.dumps(out))
