#!/usr/bin/env python3
"""LocalAlignmentScorer (BASELINE config 4b) kernel-only throughput: striped register kernel vs the
literal LDS kernel (HMK_LOCAL_LITERAL=1), same block, parity against the oracle on 64 rows."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
from oracle import c_oracle

M = load_blosum62()
n, rows = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
res, off = synth_peptides(1, n, 7, 20)
L = np.diff(off.astype(np.int64))
st, want = c_oracle.score_block(M, res, off, np.arange(0, 64), np.arange(0, n), 1, -5, -1)
for mode in ("striped", "literal"):
    if mode == "literal":
        os.environ["HMK_LOCAL_LITERAL"] = "1"
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    kms = []
    for _ in range(3):
        out = ctx.score_block_local(0, rows, 0, n, -5, -1)
        kms.append(ctx.last_kernel_ms())
    k = min(kms)
    cells = float(L[:rows].sum()) * float(L.sum())
    print(json.dumps({"config": "4b LocalAlignmentScorer open -5 ext -1, 1e5 x 7..20", "kernel": mode,
                      "block": [rows, n], "ordered_pairs": rows * n, "kernel_ms": k,
                      "pairs_per_s": rows * n / (k * 1e-3), "dp_cells_per_s": cells / (k * 1e-3),
                      "parity_first_64_rows": bool(np.array_equal(out[:64], want))}), flush=True)

# all ORDERED pairs, thresholded (hmk_neighbors_local): the length-bucketed form
os.environ.pop("HMK_LOCAL_LITERAL", None)
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off)
edges, st = ctx.neighbors_local(-5, -1, 24, part=0, n_parts=8, capacity=1 << 26)
edges, st = ctx.neighbors_local(-5, -1, 24, part=0, n_parts=8, capacity=1 << 26)
x, m, s = hammock_amd.edge_fields(edges)
pick = np.random.default_rng(0).choice(len(edges), min(100000, len(edges)), replace=False)
ok, want = c_oracle.score_pairs(M, res, off, m[pick], x[pick], 1, -5, -1)
print(json.dumps({"config": "4b LocalAlignmentScorer open -5 ext -1, 1e5 x 7..20, all ordered pairs >= 24 (1 of 8 shards)",
                  "kernel": "k_neighbors_local", "ordered_pairs": int(st.pairs_scored), "edges": int(len(edges)),
                  "kernel_ms": st.kernel_ms, "pairs_per_s": st.pairs_scored / (st.kernel_ms * 1e-3),
                  "parity_sampled_edges": bool(ok == 0 and np.array_equal(want, s[pick]))}), flush=True)
