"""CPU tests of the product's host-side logic: the exact greedy merge
(hammock_amd/csrc/hmk_greedy.cpp through hmk_greedy_from_edges) fed with an
edge list the ORACLE scorer produced, against the oracle's literal greedy; and
that libhammock_hip.so loads and exports every symbol of include/hammock_hip.h.
No GPU compute is called here."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, random_peptides
from oracle import hammock_oracle as po

import hammock_amd
from hammock_amd import _native as N


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "hammock_hip.h")) as fh:
        declared = set(re.findall(r"\b(hmk_[a-z_0-9]+)\s*\(", fh.read()))
    assert declared == set(N.SYMBOLS)
    for name in declared:
        assert hasattr(N.lib, name), name
    assert N.lib.hmk_abi_version() == 4


def test_no_gpu_fails_loudly(blosum62):
    """No CPU fallback: scoring on a host-only context is an error, not a result."""
    ctx = hammock_amd.Context(blosum62, device=-1)
    ctx.set_sequences(["WVTAPRSLPVLP", "RSPIVRQLPSLP"])
    with pytest.raises(hammock_amd.DeviceError):
        ctx.score_pairs_shifted([0], [1], 3, 0)
    with pytest.raises(hammock_amd.DeviceError):
        ctx.greedy_cluster(3, 0, 20, 1)
    with pytest.raises(hammock_amd.DeviceError):
        ctx.neighbors_shifted(3, 0, 20)


def test_argument_validation(blosum62):
    ctx = hammock_amd.Context(blosum62, device=-1)
    with pytest.raises(ValueError):
        ctx.set_sequences(residues=np.array([24], dtype=np.uint8), offsets=np.array([0, 1], dtype=np.uint32))
    with pytest.raises(ValueError):
        ctx.set_sequences(["A" * 33])
    with pytest.raises(hammock_amd.FileFormatException):
        hammock_amd.encode("ACDJ")
    with pytest.raises(ValueError):
        hammock_amd.Context(np.full((24, 24), 5000), device=-1)


def oracle_edges(coracle, M, res, off, X, p, thr, symmetric):
    n = len(off) - 1
    ii, jj = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32), indexing="ij")
    keep = ii < jj if symmetric else ii != jj
    x, m = ii[keep], jj[keep]
    st, sc = coracle.score_pairs(M, res, off, m, x, 0, X, p)  # score(seq1 = m, seq2 = x)
    assert st == 0
    hit = sc >= thr
    return hammock_amd.pack_edges(x[hit], m[hit], sc[hit])


def run_both(coracle, M, peps, sizes, X, p, thr, maxc):
    res, off = coracle.pack(peps)
    st, cid, order, stats = coracle.greedy_cluster(M, res, off, sizes, 0, X, p, thr, maxc, 1)
    symmetric = bool((M == M.T).all())
    edges = oracle_edges(coracle, M, res, off, X, p, thr, symmetric)
    ctx = hammock_amd.Context(M, device=-1)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    if st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH:
        with pytest.raises(hammock_amd.ReferenceWouldCrash) as ei:
            ctx.greedy_from_edges(edges, symmetric, thr, maxc)
        assert (ei.value.case, ei.value.index) == (stats.crash_case, stats.crash_index)
        return None
    assert st == 0
    rng = np.random.default_rng(0)
    rng.shuffle(edges)  # edge order must not matter
    gcid, gorder, gstats = ctx.greedy_from_edges(edges, symmetric, thr, maxc)
    assert np.array_equal(gcid, cid)
    assert np.array_equal(gorder, order)
    assert np.array_equal(ctx.member_rank[:len(cid)], stats.member_rank)   # Cluster.getSequences() insertion order
    assert gstats.phase1_stop_index == stats.phase1_stop_index
    assert gstats.phase1_clusters == stats.phase1_clusters
    assert gstats.phase1_orphans == stats.phase1_orphans
    assert gstats.n_multi == stats.n_multi
    return cid


@pytest.mark.parametrize("seed", range(6))
def test_greedy_from_edges_matches_oracle(blosum62, coracle, seed):
    rng = np.random.default_rng(100 + seed)
    peps = random_peptides(rng, 300, 9 if seed % 2 else 12, 12, alphabet=4 + seed % 3)
    sizes = rng.integers(1, 5, size=len(peps)).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [peps[k] for k in perm]
    if sizes is not None:
        sizes = sizes[perm]
    run_both(coracle, blosum62, peps, sizes, 2 + seed % 2, -(seed % 2), 16 + seed, 8 + 3 * seed)


def test_greedy_from_edges_asymmetric_matrix(blosum62, coracle):
    """The loader does not enforce symmetry (FileIOManager.java:46-81): with an
    asymmetric matrix score(a, b) != score(b, a) for equal lengths and the
    greedy needs the directed edges."""
    rng = np.random.default_rng(5)
    M = blosum62.copy()
    M[np.triu_indices(24, 1)] += rng.integers(-2, 3, size=276).astype(np.int32)
    assert not (M == M.T).all()
    peps = random_peptides(rng, 200, 10, 12, alphabet=5)
    run_both(coracle, M, peps, None, 3, 0, 18, 10)


def test_greedy_from_edges_crash_cases(blosum62, coracle):
    far = [coracle.encode(s) for s in ["WWWWWWWW", "CCCCCCCC", "PPPPPPPP", "GGGGGGGG"]]
    assert run_both(coracle, blosum62, far, None, 2, 0, 30, 3) is None          # case 1
    assert run_both(coracle, blosum62, far[:1], None, 2, 0, 30, 3) is None      # case 2
    three = [coracle.encode(s) for s in ["WWWWWWWW", "WWWWWWWF", "CCCCCCCC"]]
    assert run_both(coracle, blosum62, three, None, 2, 0, 30, 3) is None        # case 3
    assert run_both(coracle, blosum62, three, None, 2, 0, 30, 1).tolist() == [0, 0, 2]
    assert run_both(coracle, blosum62, three, None, 2, 0, 30, 0).tolist() == [0, 1, 2]


def test_greedy_from_edges_musi(blosum62, coracle):
    seqs = po.load_unique_sequences_from_fasta(os.path.join(GOLDEN, "musi.fa"))
    thr, X, maxc = po.greedy_defaults(seqs)
    po.sort_sequences(seqs, "size")
    peps = [coracle.encode(s.get_sequence_string()) for s in seqs]
    cid = run_both(coracle, blosum62, peps, None, X, 0, thr, maxc)
    assert int((np.bincount(cid) > 1).sum()) == 61


@pytest.mark.parametrize("asym", [False, True])
def test_greedy_from_edges_large_uses_precheck_and_inbox(blosum62, coracle, asym):
    """> 512 leftovers: the threaded monotone pre-check and (symmetric scores) the inbox form of the
    second loop run; the result must still equal the oracle's literal sequential greedy."""
    rng = np.random.default_rng(9 if asym else 8)
    M = blosum62.copy()
    if asym:
        M[np.triu_indices(24, 1)] += rng.integers(-1, 2, size=276).astype(np.int32)
    peps = random_peptides(rng, 3000, 12, 12, alphabet=6)   # low complexity: dense graph, clusters keep growing
    sizes = rng.integers(1, 4, size=len(peps)).astype(np.int32)
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [peps[k] for k in perm]
    cid = run_both(coracle, M, peps, sizes[perm], 3, 0, 24, 60)
    assert cid is not None and int((np.bincount(cid) > 1).sum()) == 60


@pytest.mark.parametrize("threads", [1, 2, 5, 8])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_greedy_phase1_window_scans_match_oracle(blosum62, coracle, monkeypatch, seed, threads):
    """Phase 1 scans a window of upcoming positions on several threads and commits the steps in order, invalidating the
    scans a commit can have changed (hmk_greedy.cpp, firstPhase).  Dense, low-complexity inputs make those conflicts the
    rule -- most rows of a window are neighbours of each other, absorb each other's candidates and join the same clusters --
    and the cluster limit is high enough that phase 1 runs over most of the list.  Ids, list order, member order and the
    phase-1 counters must equal the oracle's literal sequential loop for every thread count."""
    monkeypatch.setenv("HMK_PHASE1_THREADS", str(threads))
    if threads == 5:
        monkeypatch.setenv("HMK_PHASE1_WINDOW", "3")       # positions a thread scans ahead per round (default 16)
    if threads == 2:
        monkeypatch.setenv("HMK_GREEDY_TIMING", "1")       # the timeline on stderr changes nothing
    rng = np.random.default_rng(4000 + seed)
    n = [900, 1500, 2500][seed]
    peps = random_peptides(rng, n, 12, 12, alphabet=[3, 4, 5][seed])
    sizes = rng.integers(1, 4, size=len(peps)).astype(np.int32) if seed != 1 else None
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [peps[k] for k in perm]
    cid = run_both(coracle, blosum62, peps, None if sizes is None else sizes[perm], 3, 0, [30, 26, 24][seed], [400, 500, 700][seed])
    assert cid is not None


@pytest.mark.parametrize("band", ["40,1", "150,2", "400,8", "100000,3"])
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_greedy_phase1_on_the_prepared_band_matches_oracle(blosum62, coracle, monkeypatch, seed, band):
    """Phase 1 on a PREPARED band (BandPack, hmk_internal.h: near rows, the rows' best far candidates, transposed lists of the
    first two) keeps the feasible clusters of every later band row incrementally instead of recounting whole rows.
    HMK_PHASE1_HOST_BAND=rows,far_t builds the band on the host from the whole graph -- the literal statement of what the
    device's k_band_* kernels produce -- so the loop can be checked on the CPU against the oracle's sequential loop: bands that
    end before, inside and after phase 1 (the whole-row loop takes over at the band's last row), far lists of 1-3 candidates
    (absorbed candidates force the on-demand fetches), counts (size order and size tie-breaks), dense low-complexity inputs."""
    monkeypatch.setenv("HMK_PHASE1_HOST_BAND", band)
    rng = np.random.default_rng(5000 + seed)
    n = [700, 1200, 1800, 2500][seed]
    peps = random_peptides(rng, n, 12, 12, alphabet=[3, 4, 5, 6][seed])
    sizes = rng.integers(1, 4, size=len(peps)).astype(np.int32) if seed != 1 else None
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, "size" if seed % 2 == 0 else "alphabetic")
    peps = [peps[k] for k in perm]
    cid = run_both(coracle, blosum62, peps, None if sizes is None else sizes[perm], 3, 0, [30, 26, 24, 22][seed], [300, 500, 700, 60][seed])
    assert cid is not None


@pytest.mark.parametrize("seed", range(40))
def test_greedy_phase1_on_the_prepared_band_small_inputs(blosum62, coracle, monkeypatch, seed):
    """The same on many small inputs, crash-parity outcomes included (the three NullPointerExceptions of
    LimitedGreedySequenceClusterer.java:97/104/108 must come out of the incremental loop with the oracle's case and index)."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(2, 220))
    monkeypatch.setenv("HMK_PHASE1_HOST_BAND", f"{int(rng.integers(1, n + 5))},{int(rng.integers(1, 5))}")
    peps = random_peptides(rng, n, 9 if seed % 2 else 12, 12, alphabet=3 + seed % 4)
    sizes = rng.integers(1, 5, size=len(peps)).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, ["size", "alphabetic", "input"][seed % 3])
    peps = [peps[k] for k in perm]
    run_both(coracle, blosum62, peps, None if sizes is None else sizes[perm], 2 + seed % 2, -(seed % 2), 14 + seed % 12, int(rng.integers(1, 40)))


# --------------------------------------------------------------------------------------
# the product's nearest-neighbour chain (hmk_clinkage.cpp) on the CPU: edges from the oracle scorer
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_clinkage_from_edges_matches_oracle(blosum62, coracle, seed):
    """hmk_clinkage_from_edges (host-only context, no GPU): the product's chain over the thresholded graph -- candidate
    lists kept sorted by id, merged by intersection with the smaller score, HashSet order emulated -- against the oracle's
    ClinkageSequenceClusterer restatement, which scores cluster pairs member by member."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(2, 500))
    peps = random_peptides(rng, n, 9 if seed % 2 else 12, 12, alphabet=4 + seed % 3)
    sizes = rng.integers(1, 5, size=n).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    X, p, thr = 2 + seed % 2, -(seed % 2), 12 + 2 * seed
    st, ocid, oorder, orank, ostats = coracle.clinkage_cluster(blosum62, res, off, sizes, X, p, thr, 2)
    assert st == 0
    edges = oracle_edges(coracle, blosum62, res, off, X, p, thr, True)
    np.random.default_rng(0).shuffle(edges)
    ctx = hammock_amd.Context(blosum62, device=-1)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    cid, order, stats = ctx.clinkage_from_edges(edges)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:n], orank)
    assert (stats.merges, stats.searches) == (ostats.merges, ostats.searches)


@pytest.mark.parametrize("version", [7, 6])
@pytest.mark.parametrize("seed", range(5))
def test_clinkage_from_edges_java7_hashset_order(blosum62, coracle, seed, version):
    """hmk_set_java_hashset(7 / 6): the product's chain with the pre-Java-8 HashSet iteration order (the reference is a Java
    1.7 project) against the oracle in the same mode -- ids, list order, member order, merge and search counts."""
    rng = np.random.default_rng(1300 + seed)
    n = int(rng.integers(2, 600))
    peps = random_peptides(rng, n, 9 if seed % 2 else 12, 12, alphabet=3 + seed % 3)
    sizes = rng.integers(1, 5, size=n).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    X, p, thr = 2 + seed % 2, -(seed % 2), 12 + 2 * seed
    coracle.set_java_hashset(version)
    try:
        st, ocid, oorder, orank, ostats = coracle.clinkage_cluster(blosum62, res, off, sizes, X, p, thr, 2)
    finally:
        coracle.set_java_hashset(8)
    assert st == 0
    edges = oracle_edges(coracle, blosum62, res, off, X, p, thr, True)
    ctx = hammock_amd.Context(blosum62, device=-1)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    ctx.set_java_hashset(version)
    cid, order, stats = ctx.clinkage_from_edges(edges)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:n], orank)
    assert (stats.merges, stats.searches) == (ostats.merges, ostats.searches)
    with pytest.raises(ValueError):
        ctx.set_java_hashset(5)


def test_clinkage_from_edges_chain_returns_to_a_stacked_cluster(matrices, coracle):
    """tests/test_oracle.py::test_clinkage_chain_returns_to_a_stacked_cluster through the product's chain: the input on
    which the reference throws NoSuchElementException and the one on which it returns sequences in two clusters are both
    refused with HMK_ERR_REFERENCE_WOULD_CRASH; at a threshold without the tie the four peptides cluster as in the oracle."""
    M = matrices["blosum75"]
    four = ["TTKFVE", "DTKFVE", "QTKFVE", "ETKFVE"]
    for strings in (four, four + ["WWWWWW", "CCCCCC", "WWWWWC"]):
        res, off = coracle.pack(strings)
        ctx = hammock_amd.Context(M, device=-1)
        ctx.set_sequences(residues=res, offsets=off)
        with pytest.raises(hammock_amd.ReferenceWouldCrash, match="still on its stack"):
            ctx.clinkage_from_edges(oracle_edges(coracle, M, res, off, 2, -2, 19, True))
    res, off = coracle.pack(four)
    ctx = hammock_amd.Context(M, device=-1)
    ctx.set_sequences(residues=res, offsets=off)
    cid, order, stats = ctx.clinkage_from_edges(oracle_edges(coracle, M, res, off, 2, -2, 25, True))
    st, ocid, oorder, orank, ostats = coracle.clinkage_cluster(M, res, off, None, 2, -2, 25, 1)
    assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder) and stats.merges == ostats.merges == 1


def test_clinkage_from_edges_errors(blosum62, coracle):
    ctx = hammock_amd.Context(blosum62, device=-1)
    with pytest.raises(hammock_amd.ReferenceWouldCrash):
        ctx.clinkage_from_edges(np.zeros(0, dtype=np.uint64))                 # empty input: NoSuchElementException parity
    ctx.set_sequences(["WVTAPRSLPVLP", "WVTAPRSLPVLA"])
    with pytest.raises(ValueError):
        ctx.clinkage_from_edges(hammock_amd.pack_edges([0], [2], [50]))       # m == n
    with pytest.raises(hammock_amd.DeviceError):
        ctx.clinkage_cluster(3, 0, 20)                                         # no GPU behind this context, no CPU fallback


def test_greedy_phase1_window_fuzz_small():
    """tests/tools/fuzz_phase1_windows.py in small: dense inputs in size / alphabetic / input order, random thresholds and cluster
    limits, 1-8 threads and windows of 1-4,096 positions; every run identical to the oracle's sequential loop (ids, list order,
    member order, phase-1 counters, crash parity).  The commits of a window patch the later rows' scans (join: minimum updated;
    new cluster: feasible iff the absorbed sequence is a neighbour too -- from the row's window lists or a search of the row)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_phase1_windows.py"), "40", "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '"all_identical": true' in r.stdout
