#!/usr/bin/env python3
"""A few plain neighbour passes of 1e5 uniform L-mers (for rocprofv3).  Usage: python3 tools/run_lx.py L X thr [passes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

L, X, thr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n = int(os.environ.get("N", "100000"))
res, off = synth_peptides(1, n, L)
dev = torch.device("cuda", 0)
cap = 1 << 28
d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
for _ in range(passes):
    ctx.neighbors_shifted_dev(X, 0, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
print(int(d_counts.sum().item()))
