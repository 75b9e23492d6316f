"""CPU fuzz of phase 1 on a PREPARED band (hmk_greedy.cpp; HMK_PHASE1_HOST_BAND builds the BandPack on the host): random band sizes,
far-list lengths 1-5, orders, counts, thresholds, cluster limits -- ids, list order, member order, stop index and crash parity
against the oracle.  usage: fuzz_phase1_band.py <inputs>"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hammock_amd
from oracle import c_oracle as coracle
from conftest import random_peptides
from test_host_greedy import oracle_edges
M = np.asarray(json.load(open(os.path.join(ROOT, "tests", "golden", "matrices.json")))["matrices"]["blosum62"], dtype=np.int32)
bad = 0; crash = 0; N = int(sys.argv[1])
for seed in range(N):
    rng = np.random.default_rng(90000 + seed)
    n = int(rng.integers(2, 900))
    os.environ["HMK_PHASE1_HOST_BAND"] = f"{int(rng.integers(1, n + 5))},{int(rng.integers(1, 6))}"
    peps = random_peptides(rng, n, int(rng.integers(7, 13)), 12, alphabet=int(rng.integers(2, 7)))
    sizes = rng.integers(1, 5, size=len(peps)).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, ["size", "alphabetic", "input"][seed % 3])
    peps = [peps[k] for k in perm]
    if sizes is not None: sizes = sizes[perm]
    res, off = coracle.pack(peps)
    X, p, thr, maxc = int(rng.integers(1, 4)), -int(rng.integers(0, 2)), int(rng.integers(8, 30)), int(rng.integers(1, 120))
    st, cid, order, stats = coracle.greedy_cluster(M, res, off, sizes, 0, X, p, thr, maxc, 1)
    edges = oracle_edges(coracle, M, res, off, X, p, thr, True)
    ctx = hammock_amd.Context(M, device=-1)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    try:
        g = ctx.greedy_from_edges(edges, True, thr, maxc)
        ok = st == 0 and np.array_equal(g[0], cid) and np.array_equal(g[1], order) and np.array_equal(ctx.member_rank[:len(cid)], stats.member_rank) and g[2].phase1_stop_index == stats.phase1_stop_index
    except hammock_amd.ReferenceWouldCrash as e:
        crash += 1
        ok = st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH and (e.case, e.index) == (stats.crash_case, stats.crash_index)
    if not ok:
        bad += 1; print("MISMATCH seed", seed, os.environ["HMK_PHASE1_HOST_BAND"], n, X, p, thr, maxc, flush=True)
print(json.dumps({"inputs": N, "mismatches": bad, "crash_parity_cases": crash}))
