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
lengths = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [7, 9, 12, 15, 20]
LDS_PEAK = 256 * 256 * 2.4e9   # B/s: 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS)
dev = torch.device("cuda", 0)
for L in lengths:
    thr, X = int(1.7 * L + 0.5), min(int(L / 4 + 0.5), L - 1)   # Hammock.java:1409-1434 (Math.round is half-up)
    res, off = synth_peptides(1, n, L)
    ctx = hammock_amd.Context(load_blosum62(), device=0)
    ctx.set_sequences(residues=res, offsets=off)
    cap = 1 << 28   # 2 GB of edges: 7-mers at the default threshold have 1.06e8 of them
    d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
    ms = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.neighbors_shifted_dev(X, 0, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    pairs = n * (n - 1) // 2
    st = ctx.last_plan()
    cells = L + 2 * X * L - X * (X + 1)            # ShiftedScorer.java:67-77: cells a pair of two L-mers adds = bytes of LDS the row-packed kernel reads for it
    ideal_ms = cells * pairs / LDS_PEAK * 1e3
    k_ms = float(sorted(ms[4:])[len(ms[4:]) // 2])   # median after 4 untimed-in-effect passes (clock ramp)
    edges = int(d_counts.sum().item())
    print(json.dumps({"n": n, "length": L, "X": X, "thr": thr, "kernel_ms": round(k_ms, 4), "kernel_ms_min": round(min(ms), 4),
                      "pairs_per_s": pairs / k_ms * 1e3, "edges": edges, "hit_fraction": edges / pairs,
                      "cells_per_pair": cells, "lds_ideal_ms": round(ideal_ms, 4), "frac": round(ideal_ms / k_ms, 3),
                      "row_packed_classes": int(st.classes_rows),
                      "classes": {"u8": st.classes_u8, "u16": st.classes_u16, "direct": st.classes_direct}}), flush=True)
