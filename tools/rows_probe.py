#!/usr/bin/env python3
"""Times the BASELINE pass (1e5 12-mers, X=3, p=0) at several thresholds: 20 = the workload (0.26 % of the pairs are hits),
60 = no hits at all (the read phase alone), 14 = six times the hits.  Usage: python tools/rows_probe.py [thr ...]"""
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

thrs = [int(v) for v in sys.argv[1:]] or [20, 60, 14]
n = int(os.environ.get("N", "100000"))
res, off = synth_peptides(1, n, 12)
dev = torch.device("cuda", 0)
cap = 1 << 27
d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
out = {}
for thr in thrs:
    ctx = hammock_amd.Context(load_blosum62(), device=0)
    ctx.set_sequences(residues=res, offsets=off)
    ms = []
    for _ in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.neighbors_shifted_dev(3, 0, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    out[thr] = {"edges": int(d_counts.sum().item()), "ms_median": float(np.median(ms[4:])), "ms_min": float(min(ms[4:])), "tiles": int(ctx.last_plan().n_tiles)}
print(json.dumps(out))
