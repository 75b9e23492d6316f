#!/usr/bin/env python3
"""Long randomized sweep of hmk_neighbors_shifted against the oracle (the GPU test runs 48 trials of the
same generator; this runs as many as asked): matrices (shipped, random symmetric, asymmetric, extreme),
length ranges up to 32, max shift, shift penalty of either sign, thresholds from the score distribution,
heavy (tryptophan-rich) peptides that straddle the 8-bit row-bound limit.
Usage: python tests/tools/fuzz_neighbors.py [trials] [seed] [first_trial]   (trials before first_trial only advance the
random stream; HMK_FUZZ_VERBOSE=1 prints every trial's parameters before it runs)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hammock_amd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402
from oracle import c_oracle  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
verbose = os.environ.get("HMK_FUZZ_VERBOSE") == "1"
with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    matrices = {k: np.asarray(v, dtype=np.int32) for k, v in json.load(fh)["matrices"].items()}
names = sorted(matrices)
rng = np.random.default_rng(seed)


def oracle_edges(M, res, off, X, p, thr):
    n = len(off) - 1
    symmetric = bool((M == M.T).all())
    out = []
    cols = np.arange(n, dtype=np.uint32)
    for r0 in range(0, n, 512):
        rows = np.arange(r0, min(n, r0 + 512), dtype=np.uint32)
        st, sc = c_oracle.score_block(M, res, off, cols, rows, 0, X, p)
        assert st == 0
        mm, xx = np.meshgrid(cols, rows, indexing="ij")
        keep = (sc >= thr) & ((xx < mm) if symmetric else (xx != mm))
        out.append(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep]))
    return np.sort(np.concatenate(out)) if out else np.zeros(0, np.uint64)


used = {"u8": 0, "u16": 0, "direct": 0}
for trial in range(trials):
    kind = trial % 5
    if kind == 0:
        M = matrices[names[int(rng.integers(len(names)))]].copy()
    elif kind == 1:
        A = rng.integers(-8, 16, size=(24, 24)).astype(np.int32)
        M = np.minimum(A, A.T)
    elif kind == 2:
        M = matrices["blosum62"].copy() + rng.integers(-2, 3, size=(24, 24)).astype(np.int32)
    elif kind == 3:
        M = rng.integers(-120, 121, size=(24, 24)).astype(np.int32)
        M = np.minimum(M, M.T) if trial % 2 else M
    else:
        M = matrices["blosum62"].copy() * int(rng.integers(1, 4))
    lo = int(rng.integers(1, 20))
    hi = int(min(32, lo + rng.integers(0, 20)))
    n = int(rng.integers(150, 700))
    n = min(n, sum(20 ** L for L in range(lo, min(hi, 8) + 1)) // 2) if hi <= 8 else n   # enough distinct peptides exist
    res, off = synth_peptides(int(rng.integers(1, 10 ** 6)), n, lo, hi)
    peps = [res[off[k]:off[k + 1]].copy() for k in range(n)]
    if kind in (0, 4):   # heavy residues in some peptides: both sides of the row-bound limit
        heavy = hammock_amd.encode("WCHYF")
        for k in rng.choice(n, n // 6, replace=False):
            m = rng.random(len(peps[k])) < rng.random()
            peps[k][m] = heavy[rng.integers(0, 5, int(m.sum()))]
        peps = list({bytes(q): q for q in peps}.values())
        n = len(peps)
    res, off = hammock_amd.pack_sequences(peps)
    lens = np.diff(off.astype(np.int64))
    X = int(rng.integers(0, min(int(lens.min()), 9)))
    p = int(rng.integers(-6, 3))
    i = rng.integers(0, n, 4000).astype(np.uint32)
    j = rng.integers(0, n, 4000).astype(np.uint32)
    q = float(rng.choice([0.0, 0.5, 0.9, 0.99, 0.999]))
    if trial < first:
        continue
    st, sc = c_oracle.score_pairs(M, res, off, i, j, 0, X, p)
    thr = int(np.quantile(sc, q))
    if verbose:
        print(json.dumps({"trial": trial, "kind": kind, "lo": lo, "hi": hi, "n": n, "X": X, "p": p, "thr": thr,
                          "Mmin": int(M.min()), "Mmax": int(M.max())}), flush=True)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    edges, stats = ctx.neighbors_shifted(X, p, thr)
    want = oracle_edges(M, res, off, X, p, thr)
    ok = np.array_equal(np.sort(edges), want)
    used["u8"] += stats.classes_u8
    used["u16"] += stats.classes_u16
    used["direct"] += stats.classes_direct
    if not ok:
        print(json.dumps({"FAIL": trial, "kind": kind, "lo": lo, "hi": hi, "n": n, "X": X, "p": p, "thr": thr,
                          "got": len(edges), "want": len(want)}), flush=True)
        sys.exit(1)
    if trial % 25 == 24:
        print(f"trial {trial + 1}/{trials} ok, classes so far {used}", flush=True)
print(json.dumps({"trials": trials, "seed": seed, "all_equal": True, "classes": used}))
