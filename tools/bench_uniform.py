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
    # measured the way bench.py measures the BASELINE line: SETTLE untimed passes back to back (the clocks of an idle MI355X need
    # ~25 ms of load), then TIMED passes back to back, every one between two events on the launch stream, ONE synchronise at
    # the end.  (Until late in round 4 this tool synchronised after every pass: the GPU idled between passes and every figure came
    # out 4-8 % slower than the same kernel in bench.py -- 12-mers 2.54-2.67 ms against 2.46.)
    SETTLE, TIMED = 12, 20
    stream = torch.cuda.current_stream(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(TIMED)]
    for k in range(SETTLE + TIMED):
        if k >= SETTLE:
            evs[k - SETTLE][0].record(stream)
        ctx.neighbors_shifted_dev(X, 0, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        if k >= SETTLE:
            evs[k - SETTLE][1].record(stream)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    pairs = n * (n - 1) // 2
    st = ctx.last_plan()
    cells = L + 2 * X * L - X * (X + 1)            # ShiftedScorer.java:67-77: cells a pair of two L-mers adds = bytes of LDS the row-packed kernel reads for it
    ideal_ms = cells * pairs / LDS_PEAK * 1e3
    k_ms = float(sorted(ms)[len(ms) // 2])           # median of the timed passes
    edges = int(d_counts.sum().item())
    print(json.dumps({"n": n, "length": L, "X": X, "thr": thr, "kernel_ms": round(k_ms, 4), "kernel_ms_min": round(min(ms), 4),
                      "pairs_per_s": pairs / k_ms * 1e3, "edges": edges, "hit_fraction": edges / pairs,
                      "cells_per_pair": cells, "lds_ideal_ms": round(ideal_ms, 4), "frac": round(ideal_ms / k_ms, 3),
                      "row_packed_classes": int(st.classes_rows), "settle_passes": SETTLE, "timed_passes": TIMED,
                      "classes": {"u8": st.classes_u8, "u16": st.classes_u16, "direct": st.classes_direct}}), flush=True)
