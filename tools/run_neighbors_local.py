#!/usr/bin/env python3
"""One shard (1/8) of hmk_neighbors_local on BASELINE config 4b, twice -- for rocprofv3 runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62

res, off = synth_peptides(1, 100000, 7, 20)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
for _ in range(2):
    edges, st = ctx.neighbors_local(-5, -1, 24, part=0, n_parts=8, capacity=1 << 26)
print(len(edges), st.kernel_ms)
