#!/usr/bin/env python3
"""hmk_score_block_local on an 8,192 x 10^5 block of BASELINE config 4's peptides (lengths 7..20, open -5, extend -1):
pairs/s of the dense form (the score matrix comes back to the host: the D2H copy is inside the call, so the figure is a floor)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

n = 100000
res, off = synth_peptides(1, n, 7, 20)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
out = []
for _ in range(3):
    t = time.perf_counter()
    sc = ctx.score_block_local(0, rows, 0, n, -5, -1)
    out.append(time.perf_counter() - t)
print(json.dumps({"block": [rows, n], "seconds": out, "pairs_per_s": rows * n / min(out), "checksum": int(sc.astype("int64").sum()),
                  "forms": {k: os.environ.get(k) for k in ("HMK_LOCAL_NO_PK", "HMK_LOCAL_SIGNED")}}))
