#!/usr/bin/env python3
"""Per-phase timing of hmk_greedy_cluster on a FASTA file with Hammock's greedy defaults (size order), e.g. the
reference's antibodies example:  python tools/greedy_phases_fasta.py tests/golden/antibodies.fa.gz"""
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from bench import load_blosum62  # noqa: E402

path = sys.argv[1]
opener = gzip.open if path.endswith(".gz") else open
counts, order_seen = {}, []
with opener(path, "rt") as fh:     # FileIOManager.loadUniqueSequencesFromFasta: >id|count|label, duplicates merged
    cnt = 1
    for line in fh:
        line = line.strip()
        if line.startswith(">"):
            f = line[1:].split("|")
            cnt = int(f[1]) if len(f) > 1 and f[1] else 1
        elif line:
            s = line.upper()
            if s not in counts:
                counts[s] = 0
                order_seen.append(s)
            counts[s] += cnt
seqs = sorted(order_seen, key=lambda s: (-counts[s], [-ord(c) for c in s]))   # size desc, then string desc
sizes = np.array([counts[s] for s in seqs], dtype=np.int32)
L = np.array([len(s) for s in seqs])
jr = lambda x: int(np.floor(x + 0.5))
thr, X, maxc = jr(L.mean() * 1.7), min(jr(L.mean() / 4), int(L.min()) - 1), jr(len(seqs) * 0.025)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(seqs, sizes=sizes)
for call in range(3):
    t = time.perf_counter()
    cid, order, st = ctx.greedy_cluster(X, 0, thr, maxc)
    line = {"input": os.path.basename(path), "n": len(seqs), "thr": thr, "X": X, "max_clusters": maxc, "call": call,
            "wall_ms": (time.perf_counter() - t) * 1e3, "clusters": int(st.n_multi), "edges": int(st.n_edges),
            "phase1_stop_index": int(st.phase1_stop_index)}
    line.update(ctx.greedy_phases())
    print(json.dumps(line), flush=True)
