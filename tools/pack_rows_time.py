#!/usr/bin/env python3
"""Isolated timing of the exchange-block builders on an idle GPU: hmk_pack_rows_dev (4-byte row
blocks) vs hmk_compact_edges_dev (8-byte edges) for a 1/P shard of the BASELINE workload."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd import _native as N  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
res, off = synth_peptides(1, n, 12)
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off)
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream(dev)
cap = ((int(n * (n - 1) // 2 * 6e-3 / parts) + (1 << 20)) // 16 + 1) * 16
e = torch.empty(cap, dtype=torch.int64, device=dev)
c = torch.zeros(N.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
ctx.neighbors_shifted_dev(3, 0, 20, 0, parts, e.data_ptr(), cap, c.data_ptr(), s.cuda_stream)
tot = int(c.sum().item())
head = torch.zeros(n + 2, dtype=torch.int32, device=dev)
adj = torch.zeros(tot + 64, dtype=torch.int32, device=dev)
blk = torch.zeros(tot + 64, dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for name, fn in (("rows", lambda: ctx.pack_rows_dev(e.data_ptr(), cap, c.data_ptr(), 20, head.data_ptr(), adj.data_ptr(),
                                                    adj.numel(), s.cuda_stream)),
                 ("edges", lambda: ctx.compact_edges_dev(e.data_ptr(), cap, c.data_ptr(), blk.data_ptr(), blk.numel(),
                                                         cnt.data_ptr(), s.cuda_stream))):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(s)
    for _ in range(50):
        fn()
    t1.record(s)
    t1.synchronize()
    print(json.dumps({"n": n, "shard": f"0/{parts}", "edges": tot, "builder": name, "us": t0.elapsed_time(t1) / 50 * 1e3}))
