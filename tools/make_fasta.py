#!/usr/bin/env python3
"""Synthetic FASTA for hammock-hip runs: make_fasta.py N out.fa [min_len max_len] [--counts]
(the generator of BASELINE's synthetic configs; --counts gives every fourth peptide a count of 1..64 and two labels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hammock_amd.synth import synth_peptides

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, path = int(args[0]), args[1]
lo, hi = (int(args[2]), int(args[3])) if len(args) >= 4 else (12, 12)
res, off = synth_peptides(1, n, lo, hi)
A = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
letters = A[res].tobytes()
counts = "--counts" in sys.argv
rng = np.random.default_rng(1)
with open(path, "wb") as f:
    out = []
    for i in range(n):
        head = b">%d" % i
        if counts and i % 4 == 0:
            head += b"|%d|%s" % (1 + int(rng.integers(0, 64)), b"A" if i % 8 else b"B")
        out.append(head + b"\n" + letters[off[i]:off[i + 1]] + b"\n")
    f.write(b"".join(out))
