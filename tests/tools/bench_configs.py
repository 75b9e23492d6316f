#!/usr/bin/env python3
"""Runs the BASELINE.json configs that fit one GPU (2, 3, 4a, 4b; 5 as a single-GPU shard) and
prints one JSON line each: kernel time, pairs/s, classes per lane path, sampled oracle parity.
Usage: python tests/tools/bench_configs.py [--quick]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
from oracle import c_oracle

quick = "--quick" in sys.argv
M = load_blosum62()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)


def neighbour_pass(name, n, lo, hi, X, p, thr, part=0, n_parts=1, reps=5):
    res, off = synth_peptides(1, n, lo, hi)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    cap = 1 << 28
    d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    ctx.neighbors_shifted_dev(X, p, thr, part, n_parts, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.neighbors_shifted_dev(X, p, thr, part, n_parts, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    st = ctx.last_plan()
    counts = d_counts.cpu().numpy()
    assert counts.max() <= cap // _native.HMK_EDGE_SHARDS
    seg = cap // _native.HMK_EDGE_SHARDS
    edges = torch.cat([d_edges[s * seg:s * seg + int(c)] for s, c in enumerate(counts)]).cpu().numpy().view(np.uint64)
    x, m, s = hammock_amd.edge_fields(edges)
    pick = np.random.default_rng(0).choice(len(edges), min(200000, len(edges)), replace=False)
    ok_st, want = c_oracle.score_pairs(M, res, off, m[pick], x[pick], 0, X, p)
    parity = bool(ok_st == 0 and np.array_equal(want, s[pick]) and (s >= thr).all())
    # complete rows: every neighbour of 20 random rows
    deg = np.bincount(np.concatenate([x, m]), minlength=n)
    rows_ok = True
    if n_parts == 1:
        for r in np.random.default_rng(1).choice(n, 20, replace=False):
            _, sc = c_oracle.score_pairs(M, res, off, np.arange(n, dtype=np.uint32), np.full(n, r, np.uint32), 0, X, p)
            sc[r] = -10 ** 6
            rows_ok &= int((sc >= thr).sum()) == deg[r]
    med = float(np.median(ms))
    print(json.dumps({"config": name, "n": n, "len": [lo, hi], "X": X, "p": p, "thr": thr, "part": [part, n_parts],
                      "pairs": int(st.pairs_scored), "edges": int(len(edges)), "kernel_ms_median": med,
                      "kernel_ms_min": float(min(ms)), "pairs_per_s": st.pairs_scored / (med * 1e-3),
                      "tiles": int(st.n_tiles), "classes": {"u8": st.classes_u8, "u16": st.classes_u16,
                                                            "direct": st.classes_direct},
                      "plan_and_first_pass_s": plan_s, "oracle_parity_sampled_edges": parity,
                      "oracle_parity_full_rows": bool(rows_ok)}), flush=True)


def local_block(name, n, lo, hi, rows, reps=3):
    """config 4b: LocalAlignmentScorer scores only (dense block, ordered pairs, seq1 = row)."""
    res, off = synth_peptides(1, n, lo, hi)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    best = kms = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ctx.score_block_local(0, rows, 0, n, -5, -1)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        kms = ctx.last_kernel_ms() if kms is None else min(kms, ctx.last_kernel_ms())
    st, want = c_oracle.score_block(M, res, off, np.arange(0, 64), np.arange(0, n), 1, -5, -1)
    print(json.dumps({"config": name, "n": n, "len": [lo, hi], "block": [rows, n], "ordered_pairs": rows * n,
                      "wall_s_incl_d2h": best, "kernel_ms": kms, "pairs_per_s_kernel": rows * n / (kms * 1e-3),
                      "cells_per_s_kernel": float(np.diff(off.astype(np.int64))[:rows].sum()) * float(np.diff(off.astype(np.int64)).sum()) / (kms * 1e-3),
                      "oracle_parity_first_64_rows": bool(st == 0 and np.array_equal(out[:64], want))}), flush=True)


neighbour_pass("2: 1e4 x 12, BLOSUM62", 10000, 12, 12, 3, 0, 20)
neighbour_pass("3: 1e5 x 12, BLOSUM62", 100000, 12, 12, 3, 0, 20)
neighbour_pass("4a: 1e5 x 7..20, ShiftedScorer p=-1", 100000 if not quick else 30000, 7, 20, 3, -1, 23)
local_block("4b: 1e5 x 7..20, LocalAlignmentScorer open -5 ext -1 (rows 0..4095 vs all)", 100000, 7, 20, 4096 if not quick else 512)
if not quick:
    neighbour_pass("5 (one of 8 shards): 1e6 x 12, BLOSUM62", 1000000, 12, 12, 3, 0, 20, part=0, n_parts=8, reps=2)
