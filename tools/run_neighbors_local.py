#!/usr/bin/env python3
"""hmk_neighbors_local on BASELINE config 4's input (10^5 peptides of length 7..20, open -5, extend -1): ALL 10^10
ordered pairs, threshold 28 (0.37 % pass), three passes.  Prints one JSON line; also the rocprofv3 target."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

n = 100000
res, off = synth_peptides(1, n, 7, 20)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
ms = []
for _ in range(3):
    edges, st = ctx.neighbors_local(-5, -1, 28, capacity=1 << 26)
    ms.append(st.kernel_ms)
print(json.dumps({"config": "4b: 1e5 x 7..20, LocalAlignmentScorer open -5 ext -1, all ordered pairs, thr 28", "ordered_pairs": int(st.pairs_scored),
                  "edges": int(len(edges)), "kernel_ms": ms, "pairs_per_s": st.pairs_scored / (min(ms) * 1e-3)}))
