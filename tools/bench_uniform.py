#!/usr/bin/env python3
"""Neighbour pass on uniform-length sets other than 12 (e.g. Ph.D.-7 libraries are 7-mers), with the
reference's default parameters for each length: thr = round(1.7 L), X = round(L / 4), p = 0."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
for L in (7, 9, 12, 15, 20):
    thr, X = int(1.7 * L + 0.5), int(L / 4 + 0.5)
    res, off = synth_peptides(1, n, L)
    ctx = hammock_amd.Context(load_blosum62(), device=0)
    ctx.set_sequences(residues=res, offsets=off)
    cap = 1 << 26
    d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.neighbors_shifted_dev(X, 0, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    pairs = n * (n - 1) // 2
    st = ctx.last_plan()
    print(json.dumps({"n": n, "length": L, "X": X, "thr": thr, "kernel_ms": min(ms), "pairs_per_s": pairs / min(ms) * 1e3,
                      "edges": int(d_counts.sum().item()), "lookups_per_s": pairs * L / min(ms) * 1e3,
                      "classes": {"u8": st.classes_u8, "u16": st.classes_u16, "direct": st.classes_direct}}), flush=True)
