#!/usr/bin/env python3
"""Long randomized sweep of the LocalAlignmentScorer kernels (packed / tagged / plain / literal tiers are
chosen by the matrix and penalty ranges) against the oracle: dense blocks and thresholded ordered pairs.
Usage: python tests/tools/fuzz_local.py [trials] [seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402
from oracle import c_oracle  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    matrices = {k: np.asarray(v, dtype=np.int32) for k, v in json.load(fh)["matrices"].items()}
names = sorted(matrices)
rng = np.random.default_rng(seed)
blocks = passes = 0
for trial in range(trials):
    if trial % 3 == 0:
        M = matrices[names[int(rng.integers(len(names)))]].copy()
    elif trial % 3 == 1:
        M = rng.integers(-31, 32, size=(24, 24)).astype(np.int32)     # tagged-max range, asymmetric
    else:
        M = rng.integers(-127, 128, size=(24, 24)).astype(np.int32)   # plain striped kernel range
    lo = int(rng.integers(1, 20))
    hi = int(min(32, lo + rng.integers(0, 20)))
    n = int(rng.integers(100, 400))
    if hi <= 8:
        n = min(n, sum(20 ** L for L in range(lo, hi + 1)) // 2)
    res, off = synth_peptides(int(rng.integers(1, 10 ** 6)), n, lo, hi)
    go = -int(rng.integers(0, 40))
    ge = -int(rng.integers(0, 40))
    if trial % 8 == 7:
        go, ge = int(rng.integers(1, 4)), -1   # positive penalty: literal kernel for blocks
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    idx = np.arange(n, dtype=np.uint32)
    st, want = c_oracle.score_block(M, res, off, idx, idx, 1, go, ge)
    got = ctx.score_block_local(0, n, 0, n, go, ge)
    info = {"trial": trial, "lo": lo, "hi": hi, "n": n, "open": go, "ext": ge}
    if not np.array_equal(got, want):
        print(json.dumps({"FAIL": "block", **info}))
        sys.exit(1)
    blocks += 1
    if go <= 0 and ge <= 0:
        thr = int(np.quantile(want, float(rng.choice([0.5, 0.9, 0.99]))))
        edges, _ = ctx.neighbors_local(go, ge, thr)
        mm, xx = np.meshgrid(idx, idx, indexing="ij")
        keep = (want >= thr) & (mm != xx)
        if not np.array_equal(np.sort(edges), np.sort(hammock_amd.pack_edges(xx[keep], mm[keep], want[keep]))):
            print(json.dumps({"FAIL": "neighbors_local", "thr": thr, **info}))
            sys.exit(1)
        passes += 1
    if trial % 25 == 24:
        print(f"trial {trial + 1}/{trials}: {blocks} blocks, {passes} thresholded passes equal", flush=True)
print(json.dumps({"trials": trials, "seed": seed, "blocks_equal": blocks, "neighbors_local_equal": passes}))
