#!/usr/bin/env python3
"""Mixed-length neighbour pass (BASELINE config 4a parameters) at several n: throughput against tile length --
what is left of the per-tile fixed cost (table build, prologue, flush)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

for n in (25000, 50000, 100000, 200000, 400000):
    res, off = synth_peptides(1, n, 7, 20)
    ctx = hammock_amd.Context(load_blosum62(), device=0)
    ctx.set_sequences(residues=res, offsets=off)
    dev = torch.device("cuda", 0)
    cap = 1 << 26
    d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
    ms = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.neighbors_shifted_dev(3, -1, 23, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    pairs = n * (n - 1) // 2
    print(n, round(min(ms), 3), "ms", f"{pairs / min(ms) * 1e3:.3e} pairs/s", ctx.last_plan().n_tiles, "tiles", flush=True)
