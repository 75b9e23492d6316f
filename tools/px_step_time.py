#!/usr/bin/env python3
"""Steady-state step time of the pipelined score -> pack -> ship loop for a 1/P shard of the
BASELINE workload on ONE GPU (the collectives degenerate to device copies): shows what the
pack kernels and the host enqueue cost next to the scoring kernel, per block format."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd import dist as hd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
res, off = synth_peptides(1, n, 12)
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off)
dev = torch.device("cuda", 0)
PARTS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
for parts in PARTS:
    for fmt in ("rows", "edges"):
        px = hd.PipelinedExchange(ctx, 3, 0, 20, 0, 1, dev, fmt=fmt, shard=(0, parts))
        for _ in range(5):
            px.step()
        px.finish()
        K = 100
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for k in range(K):
            px.step(*ev[k])
        t_enq = time.perf_counter() - t0
        px.finish()
        t = time.perf_counter() - t0
        kms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        print(json.dumps({"n": n, "shard": f"0/{parts}", "fmt": fmt, "ms_per_step": t / K * 1e3, "kernel_ms": kms,
                          "host_enqueue_ms_per_step": t_enq / K * 1e3, "local_edges": px.local_total,
                          "block_bytes": px.bytes_per_step}), flush=True)
        del px
