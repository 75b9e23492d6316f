#!/usr/bin/env python3
"""Times the tiling variants of the hot neighbour kernel (HMK_HOT_VARIANT x
HMK_COLS_PER_TILE) on the BASELINE workload, interleaved rounds in one process
(cdna_hip_programming.md rule 24).  Usage: python tools/tune_hot.py [n] [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,6,7").split(",")]
cols = [int(v) for v in os.environ.get("COLS", "16384").split(",")]
M = load_blosum62()
res, off = synth_peptides(1, n, 12)
dev = torch.device("cuda", 0)
cap = (1 << 25)
d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev)
ctxs = {}
ref_edges = None
for v in variants:
    for c in cols:
        os.environ["HMK_HOT_VARIANT"] = str(v)
        os.environ["HMK_COLS_PER_TILE"] = str(c)
        ctx = hammock_amd.Context(M, device=0)
        ctx.set_sequences(residues=res, offsets=off)
        ctx.neighbors_shifted_dev(3, 0, 20, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        seg = cap // _native.HMK_EDGE_SHARDS
        cnts = d_counts.tolist()
        es = torch.sort(torch.cat([d_edges[q * seg:q * seg + int(k)] for q, k in enumerate(cnts)]))[0]
        if ref_edges is None:
            ref_edges = es
        assert torch.equal(es, ref_edges), f"variant {v} cols {c}: edge list differs from the first variant"
        ctxs[(v, c)] = (ctx, int(d_counts.sum().item()), ctx.last_plan().n_tiles)
times = {k: [] for k in ctxs}
for _ in range(rounds):
    for k, (ctx, _, _) in ctxs.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.neighbors_shifted_dev(3, 0, 20, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b))
pairs = n * (n - 1) // 2
names = {0: "R16C2", 1: "R8C2", 2: "R12C2", 3: "R8C4", 4: "R5C2", 5: "R7C2", 6: "R4C2", 7: "R6C2", 8: "R6C3", 9: "R6C1"}
for k in sorted(times, key=lambda k: np.median(times[k])):
    t = times[k]
    print(f"variant {k[0]} {names[k[0]]:6s} cols {k[1]:5d} tiles {ctxs[k][2]:6d} edges {ctxs[k][1]:9d} "
          f"median {np.median(t):7.3f} ms min {min(t):7.3f} ms  {pairs / np.median(t) / 1e6:7.1f} Gpairs/s")
