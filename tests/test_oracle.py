"""CPU tests of the oracle itself: known answers, agreement of
the two independent restatements (C and pure Python), crash parity, ordering."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, random_peptides
from oracle import hammock_oracle as po

AA = po.AMINO_ACIDS


def to_str(p):
    return "".join(AA[int(x)] for x in p)


def per_shift_sums(M, s1, s2, X, p):
    """Per-shift list exactly as SURVEY 8(c) tabulates it (shift of the shorter)."""
    sc = po.ShiftedScorer(M, p, X)
    a, b = po.UniqueSequence(s1), po.UniqueSequence(s2)
    A, B = a.sequence, b.sequence
    shorter, longer = (B, A) if len(A) >= len(B) else (A, B)
    d = len(longer) - len(shorter)
    out = []
    for s in range(-X, X + d + 1):
        if s <= 0:
            v = sum(M[shorter[i - s]][longer[i]] for i in range(len(shorter) + s))
        else:
            v = sum(M[shorter[i]][longer[i + s]] for i in range(min(len(shorter), len(longer) - s)))
        v += d * p + (-s * 2 * p if s < 0 else 0) + ((s - d) * 2 * p if s > d else 0)
        out.append(v)
    assert max(out) == sc.sequence_score(a, b)
    return out


def test_known_answers_shifted(known_answers, blosum62, coracle):
    M = blosum62.tolist()
    for row in known_answers["shifted_blosum62"]:
        st, score, _ = coracle.shifted_score(blosum62, row["seq1"], row["seq2"], row["X"], row["p"])
        assert st == 0 and score == row["score"], row
        py = po.ShiftedScorer(M, row["p"], row["X"]).sequence_score(
            po.UniqueSequence(row["seq1"]), po.UniqueSequence(row["seq2"]))
        assert py == row["score"], row
        if "per_shift" in row:
            assert per_shift_sums(M, row["seq1"], row["seq2"], row["X"], row["p"]) == row["per_shift"]


def test_known_answers_local(known_answers, blosum62, coracle):
    M = blosum62.tolist()
    for row in known_answers["local_blosum62_open-5_ext-1"]:
        assert coracle.local_score(blosum62, row["seq1"], row["seq2"], -5, -1) == row["score"], row
        sc = po.LocalAlignmentScorer(M, -5, -1)
        assert sc.sequence_score(po.UniqueSequence(row["seq1"]), po.UniqueSequence(row["seq2"])) == row["score"]
        if "swapped" in row:  # order dependence of the direction-matrix rule
            assert coracle.local_score(blosum62, row["seq2"], row["seq1"], -5, -1) == row["swapped"]


def test_shift_too_big(blosum62, coracle):
    # ShiftedScorer.java:59-62: maxShift >= shorter length -> DataException
    st, _, _ = coracle.shifted_score(blosum62, "ACDEFGH", "CDEFGHIKLM", 7, 0)
    assert st == coracle.HMO_ERR_SHIFT_TOO_BIG
    st, _, _ = coracle.shifted_score(blosum62, "ACDEFGH", "CDEFGHIKLM", 6, 0)
    assert st == 0
    with pytest.raises(po.DataException):
        po.ShiftedScorer(blosum62.tolist(), 0, 7).sequence_score(po.UniqueSequence("ACDEFGH"),
                                                                 po.UniqueSequence("CDEFGHIKLM"))


@pytest.mark.parametrize("mat", ["blosum62", "pam250", "blosum30", "mcla71"])
def test_c_vs_python_scorers_random(matrices, coracle, mat):
    rng = np.random.default_rng(7)
    M = matrices[mat]
    Ml = M.tolist()
    peps = random_peptides(rng, 60, 7, 20, alphabet=24)
    for X, p in [(0, 0), (3, 0), (3, -1), (6, -4), (1, 2)]:
        sc = po.ShiftedScorer(Ml, p, X)
        for _ in range(150):
            i, j = rng.integers(0, len(peps), 2)
            a, b = peps[i], peps[j]
            st, score, shift = coracle.shifted_score(M, a, b, X, p)
            ua, ub = po.UniqueSequence(to_str(a)), po.UniqueSequence(to_str(b))
            pscore, pshift = sc.score_with_shift(ua, ub)
            assert st == 0 and (score, shift) == (pscore, pshift)
    for go, ge in [(-5, -1), (-10, -2), (-3, -3), (0, 0)]:
        sc = po.LocalAlignmentScorer(Ml, go, ge)
        for _ in range(100):
            i, j = rng.integers(0, len(peps), 2)
            a, b = peps[i], peps[j]
            assert coracle.local_score(M, a, b, go, ge) == sc.sequence_score(
                po.UniqueSequence(to_str(a)), po.UniqueSequence(to_str(b)))


def test_shifted_properties(blosum62, coracle):
    """SURVEY 7.5 property tests (symmetric M): symmetry, X=0 = plain sum, monotone in p."""
    rng = np.random.default_rng(11)
    peps = random_peptides(rng, 40, 7, 20)
    for _ in range(300):
        i, j = rng.integers(0, len(peps), 2)
        a, b = peps[i], peps[j]
        X = int(rng.integers(0, 7))
        _, s_ab, _ = coracle.shifted_score(blosum62, a, b, X, -1)
        _, s_ba, _ = coracle.shifted_score(blosum62, b, a, X, -1)
        assert s_ab == s_ba
        _, s_p0, _ = coracle.shifted_score(blosum62, a, b, X, 0)
        _, s_p2, _ = coracle.shifted_score(blosum62, a, b, X, -2)
        assert s_p0 >= s_ab >= s_p2
        if len(a) == len(b):
            _, s0, _ = coracle.shifted_score(blosum62, a, b, 0, -3)
            assert s0 == int(sum(blosum62[x, y] for x, y in zip(b, a)))


def test_local_never_above_gotoh(blosum62, coracle):
    """ref <= 3-matrix Gotoh; equality is NOT required."""
    def gotoh(a, b, go, ge):
        NEG = -10 ** 9
        n, m = len(a), len(b)
        H = [[0] * (m + 1) for _ in range(n + 1)]
        E = [[NEG] * (m + 1) for _ in range(n + 1)]
        F = [[NEG] * (m + 1) for _ in range(n + 1)]
        best = 0
        for i in range(1, n + 1):
            for j in range(1, m + 1):
                E[i][j] = max(E[i][j - 1] + ge, H[i][j - 1] + go)
                F[i][j] = max(F[i - 1][j] + ge, H[i - 1][j] + go)
                H[i][j] = max(0, H[i - 1][j - 1] + int(blosum62[a[i - 1], b[j - 1]]), E[i][j], F[i][j])
                best = max(best, H[i][j])
        return best
    rng = np.random.default_rng(5)
    peps = random_peptides(rng, 50, 7, 20)
    lower = 0
    for _ in range(400):
        i, j = rng.integers(0, len(peps), 2)
        r = coracle.local_score(blosum62, peps[i], peps[j], -5, -1)
        g = gotoh(peps[i], peps[j], -5, -1)
        assert r <= g
        lower += r < g
    assert lower > 0  # the direction-matrix rule does differ from Gotoh


def _greedy_both(coracle, M, peps, sizes, X, p, thr, maxc, n_threads=1, scorer=0):
    res, off = coracle.pack(peps)
    st, cid, order, stats = coracle.greedy_cluster(M, res, off, sizes, scorer, X, p, thr, maxc, n_threads)
    seqs = [po.UniqueSequence(to_str(q), {"no_label": int(sizes[k]) if sizes is not None else 1})
            for k, q in enumerate(peps)]
    sc = po.ShiftedScorer(M.tolist(), p, X) if scorer == 0 else po.LocalAlignmentScorer(M.tolist(), X, p)
    cl = po.LimitedGreedySequenceClusterer(sc, thr, maxc, n_threads=3)
    try:
        result = cl.cluster(seqs)
    except po.ReferenceWouldCrash as e:
        assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH
        assert (stats.crash_case, stats.crash_index) == (e.case, e.index)
        return st, None, stats
    assert st == 0
    index_of = {id(s): k for k, s in enumerate(seqs)}
    py_cid = np.full(len(peps), -1, dtype=np.int32)
    py_rank = np.full(len(peps), -1, dtype=np.int32)
    for c in result:
        for pos, s in enumerate(c.sequences):
            py_cid[index_of[id(s)]] = c.id
            py_rank[index_of[id(s)]] = pos
    assert np.array_equal(py_cid, cid)
    assert np.array_equal(py_rank, stats.member_rank)   # Cluster.getSequences() insertion order, both restatements
    assert [c.id for c in result] == order.tolist()
    assert cl.stats["score_calls_phase1"] == stats.score_calls_phase1
    assert cl.stats["score_calls_phase2"] == stats.score_calls_phase2
    assert cl.stats["phase1_stop_index"] == stats.phase1_stop_index
    return st, cid, stats


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_greedy_c_vs_python_random(blosum62, coracle, seed):
    rng = np.random.default_rng(seed)
    # low-complexity alphabet -> dense neighbourhoods, real clusters
    peps = random_peptides(rng, 220, 9, 12, alphabet=4 + seed)
    sizes = rng.integers(1, 6, size=len(peps)).astype(np.int32)
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [peps[k] for k in perm]
    sizes = sizes[perm]
    thr = 14 + 2 * seed
    st, cid, stats = _greedy_both(coracle, blosum62, peps, sizes, 2, -1 if seed % 2 else 0, thr, 12)
    if st == 0:
        assert stats.n_multi > 0


def test_greedy_thread_count_independent(blosum62, coracle):
    rng = np.random.default_rng(3)
    peps = random_peptides(rng, 400, 12, 12, alphabet=5)
    res, off = coracle.pack(peps)
    ref = None
    for t in (1, 2, 8):
        st, cid, order, stats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 20, 25, t)
        assert st == 0
        cur = (cid.tolist(), order.tolist(), stats.score_calls_phase1, stats.score_calls_phase2)
        if ref is None:
            ref = cur
        assert cur == ref


def test_greedy_local_scorer(blosum62, coracle):
    rng = np.random.default_rng(9)
    peps = random_peptides(rng, 120, 7, 14, alphabet=5)
    st, cid, stats = _greedy_both(coracle, blosum62, peps, None, -5, -1, 18, 10, scorer=1)
    assert st == 0


def test_crash_parity(blosum62, coracle):
    """The three NullPointerException rows of SURVEY.md section 3.2."""
    far = ["WWWWWWWW", "CCCCCCCC", "PPPPPPPP", "GGGGGGGG"]  # mutually far below thr
    # case 1: clusters empty, x has no later neighbour
    st, _, stats = _greedy_both(coracle, blosum62, [coracle.encode(s) for s in far], None, 2, 0, 30, 3)
    assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH and stats.crash_case == 1 and stats.crash_index == 0
    # case 2: clusters empty and x is the last element (single sequence)
    st, _, stats = _greedy_both(coracle, blosum62, [coracle.encode("WWWWWWWW")], None, 2, 0, 30, 3)
    assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH and stats.crash_case == 2
    # case 3: a cluster exists, nothing feasible, x is the last element
    seqs = ["WWWWWWWW", "WWWWWWWF", "CCCCCCCC"]
    st, _, stats = _greedy_both(coracle, blosum62, [coracle.encode(s) for s in seqs], None, 2, 0, 30, 3)
    assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH and stats.crash_case == 3 and stats.crash_index == 1
    # maxClusters reached before the last element: no crash, last stays a singleton
    st, cid, stats = _greedy_both(coracle, blosum62, [coracle.encode(s) for s in seqs], None, 2, 0, 30, 1)
    assert st == 0 and cid.tolist() == [0, 0, 2]
    # empty input and maxClusters == 0: loop never runs
    st, cid, order, stats = coracle.greedy_cluster(blosum62, *coracle.pack([]), None, 0, 2, 0, 30, 3, 1)
    assert st == 0 and len(cid) == 0
    st, cid, stats = _greedy_both(coracle, blosum62, [coracle.encode(s) for s in seqs], None, 2, 0, 30, 0)
    assert st == 0 and cid.tolist() == [0, 1, 2]


def test_musi_matches_survey_provisional(known_answers, blosum62, coracle):
    """The C oracle reproduces every provisional MUSI figure of SURVEY.md 8(c)."""
    exp = known_answers["musi_greedy_provisional"]
    seqs = po.load_unique_sequences_from_fasta(os.path.join(GOLDEN, "musi.fa"))
    assert len(seqs) == exp["n"]
    thr, X, maxc = po.greedy_defaults(seqs)
    assert (thr, X, maxc) == (exp["threshold"], exp["max_shift"], exp["max_clusters"])
    po.sort_sequences(seqs, "size")
    strings = [s.get_sequence_string() for s in seqs]
    assert strings[:5] == exp["first_five"]
    res, off = coracle.pack(strings)
    n = len(strings)
    for k, want in enumerate(exp["neighbours_ge20_first_three"]):
        others = np.array([j for j in range(n) if j != k])
        st, sc = coracle.score_pairs(blosum62, res, off, others, np.full(n - 1, k), 0, X, 0)
        assert st == 0 and int((sc >= thr).sum()) == want
    st, cid, order, stats = coracle.greedy_cluster(blosum62, res, off, None, 0, X, 0, thr, maxc, 2)
    assert st == 0
    assert stats.phase1_stop_index == exp["phase1_stop_index"]
    assert stats.phase1_clusters == exp["phase1_clusters"] and stats.phase1_orphans == exp["phase1_orphans"]
    assert stats.score_calls_phase1 == exp["score_calls_phase1"]
    assert stats.score_calls_phase2 == exp["score_calls_phase2"]
    sizes = np.bincount(cid)
    sizes = np.sort(sizes[sizes > 1])[::-1]
    assert len(sizes) == exp["final_clusters"]
    assert int((np.bincount(cid) == 1).sum()) == exp["final_singletons"]
    assert sizes[:10].tolist() == exp["ten_largest_unique_sizes"]
    with open(os.path.join(GOLDEN, "musi_greedy_oracle.json")) as fh:
        pin = json.load(fh)
    assert pin["order"] == strings and pin["cluster_id"] == cid.tolist()


def test_sort_orders(coracle):
    rng = np.random.default_rng(2)
    peps = random_peptides(rng, 300, 7, 12, alphabet=24)
    sizes = rng.integers(1, 4, size=len(peps)).astype(np.int32)
    res, off = coracle.pack(peps)
    for order in ("size", "alphabetic", "input"):
        perm = coracle.sort_order(res, off, sizes, order)
        seqs = [po.UniqueSequence(to_str(p), {"x": int(sizes[k])}) for k, p in enumerate(peps)]
        idx = {id(s): k for k, s in enumerate(seqs)}
        po.sort_sequences(seqs, order)
        assert [idx[id(s)] for s in seqs] == perm.tolist()


def test_java_random_shuffle_known_values():
    # java.util.Random(42): first nextInt() values are a widely published sequence
    r = po.JavaRandom(42)
    assert [r.next(32) for _ in range(3)] == [-1170105035, 234785527, -1360544799]
    r = po.JavaRandom(42)
    assert [r.next_int(10) for _ in range(5)] == [0, 3, 8, 4, 0]


def test_loaders_manual_example():
    seqs = po.load_unique_sequences_from_fasta(os.path.join(GOLDEN, "manual_example.fa"))
    with open(os.path.join(GOLDEN, "manual_example_expected.json")) as fh:
        exp = json.load(fh)["sequences"]
    assert [[s.get_sequence_string(), s.labels_map] for s in seqs] == exp
    tab = po.load_unique_sequences_from_table(os.path.join(GOLDEN, "manual_example.tsv"))
    assert [[s.get_sequence_string(), s.labels_map] for s in tab] == exp


def test_synth_generator(coracle):
    res, off = coracle.synth(1, 1000, 12)
    assert len(res) == 12000 and res.max() < 20
    assert len({res[off[k]:off[k + 1]].tobytes() for k in range(1000)}) == 1000
    res2, off2 = coracle.synth(1, 500, 7, 20)
    L = np.diff(off2.astype(np.int64))
    assert L.min() >= 7 and L.max() <= 20 and len(set(L.tolist())) == 14


# --------------------------------------------------------------------------------------
# clinkage mode (ClinkageSequenceClusterer + CachedClusterScorer + DynamicMatrix)
# --------------------------------------------------------------------------------------
def _clinkage_python(M, peps, sizes, X, p, thr, n_threads=1, size_limit=1):
    seqs = [po.UniqueSequence(to_str(q), {"no_label": int(sizes[k]) if sizes is not None else 1}) for k, q in enumerate(peps)]
    cl = po.ClinkageSequenceClusterer(po.ShiftedScorer(M.tolist(), p, X), thr, size_limit=size_limit, n_threads=n_threads)
    result = cl.cluster(seqs)
    index_of = {id(s): k for k, s in enumerate(seqs)}
    cid = np.full(len(peps), -1, dtype=np.int32)
    rank = np.full(len(peps), -1, dtype=np.int32)
    for c in result:
        for pos, s in enumerate(c.sequences):
            cid[index_of[id(s)]] = c.id
            rank[index_of[id(s)]] = pos
    return cid, [c.id for c in result], rank, cl.stats


@pytest.mark.parametrize("seed", range(8))
def test_clinkage_c_vs_literal_python(blosum62, coracle, seed):
    """The C form (own memo of cluster scores, HashSet order emulated) against the literal Python restatement
    (CachedClusterScorer + DynamicMatrix + HashSet parts) on random inputs with real merges."""
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(2, 260))
    peps = random_peptides(rng, n, 8 if seed % 2 else 12, 12, alphabet=3 + seed % 4)
    sizes = rng.integers(1, 4, size=n).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    X, p, thr = seed % 4, -(seed % 2), 10 + 3 * seed
    st, cid, order, rank, stats = coracle.clinkage_cluster(blosum62, res, off, sizes, X, p, thr, 1 + seed % 3)
    assert st == 0
    pcid, porder, prank, pstats = _clinkage_python(blosum62, peps, sizes, X, p, thr)
    assert np.array_equal(cid, pcid) and order.tolist() == porder and np.array_equal(rank, prank)
    assert stats.merges == pstats["merges"] and stats.searches == pstats["searches"]
    if seed < 3:
        assert stats.merges > 0


@pytest.mark.parametrize("version", [7, 6])
@pytest.mark.parametrize("seed", range(6))
def test_clinkage_java7_hashset_order_c_vs_literal_python(blosum62, coracle, seed, version):
    """The reference is a Java 1.7 project (nbproject/project.properties:45-46); chain starts and the returned list's order
    are HashSet iteration orders (ClinkageSequenceClusterer.java:70,118-123), which Java 8 changed.  Both oracles emulate the
    older orders on request (7: JDK 7u6+, 6: JDK 6 / early 7): they must agree with each other in those modes too."""
    rng = np.random.default_rng(700 + seed)
    n = int(rng.integers(2, 300))
    peps = random_peptides(rng, n, 8 if seed % 2 else 12, 12, alphabet=3 + seed % 4)
    sizes = rng.integers(1, 4, size=n).astype(np.int32) if seed % 3 else None
    res, off = coracle.pack(peps)
    X, p, thr = seed % 4, -(seed % 2), 10 + 3 * seed
    coracle.set_java_hashset(version)
    po.JAVA_HASHSET = version
    try:
        st, cid, order, rank, stats = coracle.clinkage_cluster(blosum62, res, off, sizes, X, p, thr, 1)
        assert st == 0
        pcid, porder, prank, pstats = _clinkage_python(blosum62, peps, sizes, X, p, thr)
    finally:
        coracle.set_java_hashset(8)
        po.JAVA_HASHSET = 8
    assert np.array_equal(cid, pcid) and order.tolist() == porder and np.array_equal(rank, prank)
    assert stats.merges == pstats["merges"] and stats.searches == pstats["searches"]


def test_java7_hashmap_order_known_values():
    """Hand-derived pins of the pre-Java-8 HashMap: hash(h) = h ^ h>>>20 ^ h>>>12, then h ^ h>>>7 ^ h>>>4 (java.util.HashMap,
    JDK 6/7).  For small non-negative h only the second step acts: hash(553 + 1) = 554 ^ (554 >>> 7 = 4) ^ (554 >>> 4 = 34) = 524
    -> bucket 12 of 16; hash(555) = 525 -> 13; hash(556) = 556 ^ 4 ^ 34 = 522 -> 10; hash(557) = 523 -> 11.  So a set holding ids
    1..4 iterates as 3, 4, 1, 2 (buckets 10, 11, 12, 13), where Java 8 (hash = h: 554.. & 15 = 10, 11, 12, 13 for ids 1..4)
    iterates in insertion order 1, 2, 3, 4."""
    assert po.JavaHashSet7._hash7(554) == 524 and po.JavaHashSet7._hash7(555) == 525
    assert po.JavaHashSet7._hash7(556) == 522 and po.JavaHashSet7._hash7(557) == 523
    class C:  # noqa: E701
        def __init__(self, i): self.id = i
    for version, want in ((7, [3, 4, 1, 2]), (6, [3, 4, 1, 2]), (8, [1, 2, 3, 4])):
        po.JAVA_HASHSET = version
        try:
            s = po._cluster_set()
        finally:
            po.JAVA_HASHSET = 8
        for i in (1, 2, 3, 4):
            s.add(C(i))
        assert [c.id for c in s] == want, version
    # 13 entries into a table of 16 (threshold 12): variant 6 resizes after the 13th insert, variant 7 only once a 13th entry
    # meets an occupied bucket -- with consecutive ids the 13th lands in an empty bucket, so 7 keeps 16 buckets where 6 has 32
    po.JAVA_HASHSET = 8
    s7, s6 = po.JavaHashSet7(lambda c: 553 + c.id, lambda c: c.id, 7), po.JavaHashSet7(lambda c: 553 + c.id, lambda c: c.id, 6)
    for i in range(1, 14):
        s7.add(C(i)); s6.add(C(i))
    assert len(s6.table) == 32 and len(s7.table) in (16, 32)


def test_clinkage_independent_of_threads_and_cache(blosum62):
    """DESIGN.md (round 1) argued that the reference's clinkage result depends on -t through stale DynamicMatrix
    entries.  The literal restatement says otherwise for a single pool thread working through the parts in order:
    only the very first clusterScore call takes the addEmpty path (whose re-used row could keep stale columns), every
    later row comes from add(index, row, value) or join(), which overwrite the whole column.  So the cache is a
    transparent memo: same clusters for every part count and with the cache bypassed (sizeLimit = infinity)."""
    rng = np.random.default_rng(77)
    for it in range(12):
        n = int(rng.integers(20, 140))
        peps = random_peptides(rng, n, 9, 12, alphabet=3 + it % 3)
        sizes = rng.integers(1, 5, size=n).astype(np.int32)
        thr = 12 + 2 * (it % 5)
        ref = _clinkage_python(blosum62, peps, sizes, 2, 0, thr, 1, 1)
        for t, sl in ((2, 1), (4, 1), (16, 1), (1, 10 ** 9), (4, 10 ** 9)):
            cur = _clinkage_python(blosum62, peps, sizes, 2, 0, thr, t, sl)
            assert np.array_equal(cur[0], ref[0]) and cur[1] == ref[1] and np.array_equal(cur[2], ref[2]), (it, t, sl)


def test_clinkage_known_small_case(blosum62, coracle):
    """Hand-checkable: three near-identical peptides and one stranger at a high threshold.  Ids: singletons k + 1,
    the first merge gets n + 2 (currentId is n + 1 after the initial loop and is incremented BEFORE use, :97)."""
    peps = ["WVTAPRSLPVLP", "WVTAPRSLPVLA", "WVTAPRSLPVLG", "GSWVVDISNVED"]
    res, off = coracle.pack(peps)
    st, cid, order, rank, stats = coracle.clinkage_cluster(blosum62, res, off, None, 3, 0, 40, 1)
    assert st == 0 and stats.merges == 2
    assert cid.tolist() == [7, 7, 7, 4] and sorted(order.tolist()) == [4, 7]
    assert sorted(rank[:3].tolist()) == [0, 1, 2]
    st, *_ = coracle.clinkage_cluster(blosum62, *coracle.pack([]), None, 0, 0, 10, 1)
    assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH   # NoSuchElementException, ClinkageSequenceClusterer.java:118


STACKED_AGAIN = ["TTKFVE", "DTKFVE", "QTKFVE", "ETKFVE"]          # BLOSUM75, X = 2, p = -2, thr = 19: found by fuzzing
STACKED_AGAIN_PLUS = STACKED_AGAIN + ["WWWWWW", "CCCCCC", "WWWWWC"]


def test_clinkage_chain_returns_to_a_stacked_cluster(matrices, coracle):
    """A tie (score, then Cluster.size(), then the smaller id) can send the reference's nearest-neighbour chain back to a
    cluster that is still on its stack, below stack[-2]; ClinkageSequenceClusterer.java:96-113 pushes it again, merges the
    upper copy and later takes the stale Cluster object below as `top`.  The literal restatement shows what follows:
    with nothing else left the active set runs empty and :118 throws NoSuchElementException; with other clusters around
    the run "succeeds" with sequences that belong to two clusters (mergedSequences aliases top's own list, :105).
    Neither is a clustering: the C oracle flags the first return to a stacked cluster."""
    M = matrices["blosum75"]
    def literal(strings):
        seqs = [po.UniqueSequence(s, {"no_label": 1}) for s in strings]
        cl = po.ClinkageSequenceClusterer(po.ShiftedScorer(M.tolist(), -2, 2), 19, size_limit=1, n_threads=1)
        return seqs, cl.cluster(seqs)
    with pytest.raises(po.NoSuchElement):
        literal(STACKED_AGAIN)
    seqs, result = literal(STACKED_AGAIN_PLUS)
    memberships = [id(s) for c in result for s in c.sequences]
    assert len(memberships) > len(seqs) and len(set(memberships)) == len(seqs)     # every sequence is there, three of them twice
    for strings in (STACKED_AGAIN, STACKED_AGAIN_PLUS):
        res, off = coracle.pack(strings)
        for threads in (1, 3):
            st, *_ = coracle.clinkage_cluster(M, res, off, None, 2, -2, 19, threads)
            assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH
    # the same four peptides at a threshold that keeps TTKFVE (24 to all others) out: no tie on the way, an ordinary result
    # (QTKFVE + ETKFVE at 27; DTKFVE stays alone: 24 to QTKFVE)
    res, off = coracle.pack(STACKED_AGAIN)
    st, cid, order, rank, stats = coracle.clinkage_cluster(M, res, off, None, 2, -2, 25, 1)
    assert st == 0 and stats.merges == 1 and cid.tolist() == [1, 2, 6, 6]


# --------------------------------------------------------------------------------------
# hand-traced vectors (tests/golden/hand_traces.md): the only pin a box without a JVM can add
# --------------------------------------------------------------------------------------
def test_hand_traced_greedy_vectors_both_oracles(coracle):
    """Every case of tests/golden/hand_traces.json was traced BY HAND through LimitedGreedySequenceClusterer.java:39-120 (one
    table row per loop iteration in hand_traces.md); both restatements must reproduce ids, list order and member order,
    or the reference's NullPointerException with its branch and index."""
    from conftest import hand_traces
    ht, M = hand_traces()
    X, p = ht["max_shift"], ht["shift_penalty"]
    for case in ht["greedy"]:
        peps = [coracle.encode(s) for s in case["sequences"]]
        sizes = np.asarray(case["sizes"], dtype=np.int32)
        for threads in (1, 3):
            st, cid, stats = _greedy_both(coracle, M, peps, sizes, X, p, case["threshold"], case["max_clusters"], n_threads=threads)
            exp = case["expect"]
            if exp["status"] == "crash":
                assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH, case["name"]
                assert (stats.crash_case, stats.crash_index) == (exp["crash_case"], exp["crash_index"]), case["name"]
                continue
            assert st == 0, case["name"]
            res, off = coracle.pack(peps)
            st, cid, order, stats = coracle.greedy_cluster(M, res, off, sizes, 0, X, p, case["threshold"], case["max_clusters"], threads)
            assert cid.tolist() == exp["cluster_id"], case["name"]
            assert order.tolist() == exp["result_order"], case["name"]
            assert np.asarray(stats.member_rank)[:len(peps)].tolist() == exp["member_rank"], case["name"]


def test_hand_traced_local_alignment_both_oracles(coracle):
    """The four LocalAlignmentScorer DP tables of hand_traces.json (filled in by hand from LocalAlignmentScorer.java:31-86, every
    cell in hand_traces.md): open + extend along a gap, DIAGONAL over UP over LEFT on ties, a zero that keeps its direction, the
    gap through a DIAGONAL cell that pays gapOpen again (10 where Gotoh gives 11), and the argument-order dependence."""
    from conftest import hand_traces
    ht, M = hand_traces()
    for case in ht["local"]:
        go, ge = case["gap_open"], case["gap_extend"]
        assert coracle.local_score(M, case["seq1"], case["seq2"], go, ge) == case["score"], case["name"]
        sc = po.LocalAlignmentScorer(M.tolist(), go, ge)
        assert sc.sequence_score(po.UniqueSequence(case["seq1"]), po.UniqueSequence(case["seq2"])) == case["score"], case["name"]
        # the table's own largest cell is the score (a typo check of the fixture)
        assert max(cell[3] for row in case["cells"] for cell in row) == case["score"], case["name"]
    by = {c["name"]: c for c in ht["local"]}
    assert (by["L3a"]["seq1"], by["L3a"]["seq2"]) == (by["L3b"]["seq2"], by["L3b"]["seq1"]) and by["L3a"]["score"] != by["L3b"]["score"]


def test_hand_traced_clinkage_vectors_both_oracles(coracle):
    """The clinkage chains of hand_traces.json, traced by hand through ClinkageSequenceClusterer.java:43-124 for the HashSet
    iteration orders of Java 8 and of JDK 7u6+ (hand_traces.md works the bucket indices out)."""
    from conftest import hand_traces
    ht, M = hand_traces()
    X, p = ht["max_shift"], ht["shift_penalty"]
    for case in ht["clinkage"]:
        peps = [coracle.encode(s) for s in case["sequences"]]
        sizes = np.asarray(case["sizes"], dtype=np.int32)
        res, off = coracle.pack(peps)
        for version, key in ((8, "expect_java8"), (7, "expect_java7")):
            exp = case[key]
            coracle.set_java_hashset(version)
            po.JAVA_HASHSET = version
            try:
                st, cid, order, rank, stats = coracle.clinkage_cluster(M, res, off, sizes, X, p, case["threshold"], 1)
                pcid, porder, prank, _ = _clinkage_python(M, peps, sizes, X, p, case["threshold"])
            finally:
                coracle.set_java_hashset(8)
                po.JAVA_HASHSET = 8
            assert st == 0
            for got_cid, got_order, got_rank, who in ((cid, order.tolist(), rank, "C"), (pcid, porder, prank, "python")):
                assert np.asarray(got_cid).tolist() == exp["cluster_id"], (case["name"], version, who)
                assert list(got_order) == exp["result_order"], (case["name"], version, who)
                assert np.asarray(got_rank).tolist() == exp["member_rank"], (case["name"], version, who)
