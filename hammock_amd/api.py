"""Python host-side mirror of the reference's interfaces for the greedy path,
backed by libhammock_hip.so through ctypes.  Names, argument order and error
behaviour follow the Java classes (paths relative to
src/cz/krejciadam/hammock/ of the reference):

  UniqueSequence                UniqueSequence.java:19
  Cluster                       Cluster.java:21
  ShiftedScorer                 ShiftedScorer.java:12     (sequenceScore on the GPU)
  LocalAlignmentScorer          LocalAlignmentScorer.java:10
  HipGreedySequenceClusterer    drop-in for LimitedGreedySequenceClusterer.java:17
                                at Hammock.java:403

``Context`` is the thin 1:1 wrapper of the C ABI.  Nothing here computes a
score on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

AMINO_ACIDS = "ARNDCQEGHILKMFPSTWYVBZX*"  # UniqueSequence.java:23-26
_NAME_TO_NUM = {c: i for i, c in enumerate(AMINO_ACIDS)}


class HammockException(Exception):
    """HammockException.java"""


class DataException(HammockException):
    """DataException.java -- e.g. "Shift too big" (ShiftedScorer.java:59-62)."""


class FileFormatException(HammockException):
    """FileFormatException.java"""


class DeviceError(HammockException):
    """HIP failure or no usable gfx950 device.  There is no CPU fallback."""


class ReferenceWouldCrash(HammockException):
    """The reference throws NullPointerException here
    (LimitedGreedySequenceClusterer.java:97/104/108, reported at Hammock.java:153-157)."""

    def __init__(self, msg, case=0, index=-1):
        super().__init__(msg)
        self.case = case
        self.index = index


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def encode(sequence: str) -> np.ndarray:
    """UniqueSequence.java:46-57: case-folded letters -> residue indices."""
    out = np.empty(len(sequence), dtype=np.uint8)
    for k, ch in enumerate(sequence.upper()):
        if ch not in _NAME_TO_NUM:
            raise FileFormatException(f"Error, character {sequence[k]} is not a valid letter from the amino acid alphabet code.")
        out[k] = _NAME_TO_NUM[ch]
    return out


def pack_sequences(seqs):
    """list of str / uint8 arrays -> (residues uint8, offsets uint32[n+1])."""
    arrs = [encode(s) if isinstance(s, str) else np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    off = np.zeros(len(arrs) + 1, dtype=np.uint32)
    if arrs:
        off[1:] = np.cumsum([len(a) for a in arrs])
    res = np.concatenate(arrs).astype(np.uint8) if arrs else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(res), off


def edge_fields(edges: np.ndarray):
    """packed uint64 edges -> (x, m, score) arrays (HMK_EDGE_* of hammock_hip.h)."""
    e = np.asarray(edges, dtype=np.uint64)
    x = (e >> np.uint64(40)).astype(np.uint32)
    m = ((e >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.uint32)
    s = (e & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16).astype(np.int32)
    return x, m, s


def pack_edges(x, m, score) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    m = np.asarray(m, dtype=np.uint64)
    s = np.asarray(score, dtype=np.int64).astype(np.int16).view(np.uint16).astype(np.uint64)
    return (x << np.uint64(40)) | (m << np.uint64(16)) | s


class Context:
    """1:1 wrapper of hmk_ctx.  device >= 0: HIP ordinal; -1: host-only."""

    def __init__(self, matrix, device=0):
        """device: a HIP ordinal, -1 (host only), or a list of ordinals (hmk_create_multi: the first is the root)."""
        self._h = C.c_void_p()
        self.matrix = np.ascontiguousarray(np.asarray(matrix, dtype=np.int32).reshape(24, 24))
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            st = N.lib.hmk_create_multi(_ptr(self.matrix, C.c_int32), devs, len(device), C.byref(self._h))
            self.devices = [int(d) for d in device]
            device = self.devices[0] if self.devices else -1
        else:
            st = N.lib.hmk_create(_ptr(self.matrix, C.c_int32), int(device), C.byref(self._h))
            self.devices = [int(device)] if int(device) >= 0 else []
        if st:
            self._h = C.c_void_p()
            self._raise(st, None)
        self.device = device
        self.n = 0

    # -- errors -----------------------------------------------------------------
    def _raise(self, st, stats=None):
        msg = N.lib.hmk_last_error(self._h if self._h else None)
        msg = msg.decode() if msg else f"hmk status {st}"
        if st == N.HMK_ERR_SHIFT_TOO_BIG:
            raise DataException(msg)
        if st == N.HMK_ERR_REFERENCE_WOULD_CRASH:
            raise ReferenceWouldCrash(msg, getattr(stats, "crash_case", 0), getattr(stats, "crash_index", -1))
        if st in (N.HMK_ERR_DEVICE, N.HMK_ERR_OOM):
            raise DeviceError(msg)
        if st == N.HMK_ERR_CAPACITY:
            raise BufferError(msg)
        raise ValueError(msg)

    def last_kernel_ms(self):
        """device time of the probe kernel(s) of the last score_pairs_* / score_block_* call"""
        return float(N.lib.hmk_last_kernel_ms(self._h))

    def close(self):
        if getattr(self, "_h", None):
            N.lib.hmk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- sequences ----------------------------------------------------------------
    def set_sequences(self, seqs=None, sizes=None, residues=None, offsets=None):
        if residues is None:
            residues, offsets = pack_sequences(seqs)
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        n = len(offsets) - 1
        sp = None
        if sizes is not None:
            sizes = np.ascontiguousarray(sizes, dtype=np.int32)
            sp = _ptr(sizes, C.c_int32)
        st = N.lib.hmk_set_sequences(self._h, _ptr(residues, C.c_uint8), _ptr(offsets, C.c_uint32), sp, n)
        if st:
            self._raise(st)
        self.n = n
        return self

    # -- scorers --------------------------------------------------------------------
    def _pairs(self, fn, i, j, a, b):
        i = np.ascontiguousarray(i, dtype=np.uint32)
        j = np.ascontiguousarray(j, dtype=np.uint32)
        if i.shape != j.shape:
            raise ValueError("i and j must have the same shape")
        out = np.empty(i.size, dtype=np.int32)
        st = fn(self._h, _ptr(i, C.c_uint32), _ptr(j, C.c_uint32), i.size, int(a), int(b), _ptr(out, C.c_int32))
        if st:
            self._raise(st)
        return out.reshape(i.shape)

    def score_pairs_shifted(self, i, j, max_shift, shift_penalty):
        return self._pairs(N.lib.hmk_score_pairs_shifted, i, j, max_shift, shift_penalty)

    def score_with_shift(self, i, j, max_shift, shift_penalty):
        """-> (score, shift) arrays: AligningSequenceScorer.scoreWithShift for every pair."""
        i = np.ascontiguousarray(i, dtype=np.uint32)
        j = np.ascontiguousarray(j, dtype=np.uint32)
        score = np.empty(i.size, dtype=np.int32)
        shift = np.empty(i.size, dtype=np.int32)
        st = N.lib.hmk_score_with_shift(self._h, _ptr(i, C.c_uint32), _ptr(j, C.c_uint32), i.size, int(max_shift),
                                        int(shift_penalty), _ptr(score, C.c_int32), _ptr(shift, C.c_int32))
        if st:
            self._raise(st)
        return score.reshape(i.shape), shift.reshape(i.shape)

    def score_pairs_local(self, i, j, gap_open, gap_extend):
        return self._pairs(N.lib.hmk_score_pairs_local, i, j, gap_open, gap_extend)

    def _block(self, fn, r0, r1, c0, c1, a, b):
        out = np.empty((max(r1 - r0, 0), max(c1 - c0, 0)), dtype=np.int32)
        st = fn(self._h, r0, r1, c0, c1, int(a), int(b), _ptr(out, C.c_int32))
        if st:
            self._raise(st)
        return out

    def score_block_shifted(self, r0, r1, c0, c1, max_shift, shift_penalty):
        return self._block(N.lib.hmk_score_block_shifted, r0, r1, c0, c1, max_shift, shift_penalty)

    def score_block_local(self, r0, r1, c0, c1, gap_open, gap_extend):
        return self._block(N.lib.hmk_score_block_local, r0, r1, c0, c1, gap_open, gap_extend)

    # -- neighbour graph ----------------------------------------------------------------
    def neighbors_shifted(self, max_shift, shift_penalty, threshold, part=0, n_parts=1, capacity=None):
        """-> (edges uint64[n_edges], NeighborStats)"""
        stats = N.NeighborStats()
        n_edges = C.c_uint64(0)
        cap = int(capacity) if capacity is not None else 1 << 20
        while True:
            buf = np.empty(max(cap, 1), dtype=np.uint64)
            st = N.lib.hmk_neighbors_shifted(self._h, int(max_shift), int(shift_penalty), int(threshold), part, n_parts,
                                             _ptr(buf, C.c_uint64), cap, C.byref(n_edges), C.byref(stats))
            if st == N.HMK_ERR_CAPACITY and capacity is None and int(n_edges.value) > cap:
                cap = int(n_edges.value)
                continue
            if st:
                self._raise(st)
            return buf[:n_edges.value].copy(), stats

    def neighbors_local(self, gap_open, gap_extend, threshold, part=0, n_parts=1, capacity=None):
        """LocalAlignmentScorer, all ordered pairs >= threshold -> (edges uint64[n_edges], NeighborStats)"""
        stats = N.NeighborStats()
        n_edges = C.c_uint64(0)
        cap = int(capacity) if capacity is not None else 1 << 20
        while True:
            buf = np.empty(max(cap, 1), dtype=np.uint64)
            st = N.lib.hmk_neighbors_local(self._h, int(gap_open), int(gap_extend), int(threshold), part, n_parts,
                                           _ptr(buf, C.c_uint64), cap, C.byref(n_edges), C.byref(stats))
            if st == N.HMK_ERR_CAPACITY and capacity is None and int(n_edges.value) > cap:
                cap = int(n_edges.value)
                continue
            if st:
                self._raise(st)
            return buf[:n_edges.value].copy(), stats

    def neighbors_shifted_dev(self, max_shift, shift_penalty, threshold, part, n_parts, d_edges_ptr, capacity,
                              d_counts_ptr, stream=0):
        st = N.lib.hmk_neighbors_shifted_dev(self._h, int(max_shift), int(shift_penalty), int(threshold), part, n_parts,
                                             C.c_void_p(d_edges_ptr), int(capacity), C.c_void_p(d_counts_ptr),
                                             C.c_void_p(stream))
        if st:
            self._raise(st)

    def compact_edges_dev(self, d_edges_ptr, capacity, d_counts_ptr, d_out_ptr, out_capacity, d_total_ptr, stream=0):
        st = N.lib.hmk_compact_edges_dev(self._h, C.c_void_p(d_edges_ptr), int(capacity), C.c_void_p(d_counts_ptr),
                                         C.c_void_p(d_out_ptr), int(out_capacity), C.c_void_p(d_total_ptr),
                                         C.c_void_p(stream))
        if st:
            self._raise(st)

    def pack_rows_dev(self, d_edges_ptr, capacity, d_counts_ptr, threshold, d_row_start_ptr, d_adj_ptr, adj_capacity, stream=0):
        st = N.lib.hmk_pack_rows_dev(self._h, C.c_void_p(d_edges_ptr), int(capacity), C.c_void_p(d_counts_ptr), int(threshold),
                                     C.c_void_p(d_row_start_ptr), C.c_void_p(d_adj_ptr), int(adj_capacity), C.c_void_p(stream))
        if st:
            self._raise(st)

    def unpack_rows_dev(self, d_row_start_ptr, d_adj_ptr, threshold, d_edges_out_ptr, out_capacity, stream=0):
        st = N.lib.hmk_unpack_rows_dev(self._h, C.c_void_p(d_row_start_ptr), C.c_void_p(d_adj_ptr), int(threshold),
                                       C.c_void_p(d_edges_out_ptr), int(out_capacity), C.c_void_p(stream))
        if st:
            self._raise(st)

    def last_plan(self):
        stats = N.NeighborStats()
        st = N.lib.hmk_neighbors_last_plan(self._h, C.byref(stats))
        if st:
            self._raise(st)
        return stats

    # -- greedy ---------------------------------------------------------------------------
    def greedy_cluster(self, max_shift, shift_penalty, threshold, max_clusters):
        """-> (cluster_id int32[n], result_order int32[n_result], GreedyStats)"""
        cid = np.full(max(self.n, 1), -1, dtype=np.int32)
        order = np.full(max(self.n, 1), -1, dtype=np.int32)
        self.member_rank = np.zeros(max(self.n, 1), dtype=np.int32)
        stats = N.GreedyStats()
        st = N.lib.hmk_greedy_cluster(self._h, int(max_shift), int(shift_penalty), int(threshold), int(max_clusters),
                                      _ptr(cid, C.c_int32), _ptr(order, C.c_int32),
                                      _ptr(self.member_rank, C.c_int32), C.byref(stats))
        if st:
            self._raise(st, stats)
        return cid[:self.n], order[:stats.n_result_clusters], stats

    def reserve(self, n_sequences):
        """hmk_reserve: sizes the buffers of a clustering call on n_sequences ahead of time (optional; the two a call needs
        last are obtained on a thread of their own and the call's CSR step waits for them)."""
        st = N.lib.hmk_reserve(self._h, int(n_sequences))
        if st:
            self._raise(st)

    def set_java_hashset(self, version):
        """hmk_set_java_hashset: 8 (default, Java 8+), 7 (JDK 7u6+) or 6 (JDK 6 / 7 before 7u6) -- whose HashSet iteration
        order the clinkage calls emulate for the chain starts and the returned list."""
        st = N.lib.hmk_set_java_hashset(self._h, int(version))
        if st:
            self._raise(st)

    def clinkage_cluster(self, max_shift, shift_penalty, threshold):
        """hmk_clinkage_cluster -> (cluster_id int32[n], result_order int32[n_result], ClinkageStats); member_rank in
        self.member_rank.  Sequences in LOAD order (clinkage mode does not sort)."""
        cid = np.full(max(self.n, 1), -1, dtype=np.int32)
        order = np.full(max(self.n, 1), -1, dtype=np.int32)
        self.member_rank = np.zeros(max(self.n, 1), dtype=np.int32)
        stats = N.ClinkageStats()
        st = N.lib.hmk_clinkage_cluster(self._h, int(max_shift), int(shift_penalty), int(threshold), _ptr(cid, C.c_int32),
                                        _ptr(order, C.c_int32), _ptr(self.member_rank, C.c_int32), C.byref(stats))
        if st == N.HMK_ERR_REFERENCE_WOULD_CRASH:
            raise ReferenceWouldCrash(N.lib.hmk_last_error(self._h).decode(), 0, -1)
        if st:
            self._raise(st)
        return cid[:self.n], order[:stats.n_result_clusters], stats

    def clinkage_from_edges(self, edges):
        """hmk_clinkage_from_edges: the nearest-neighbour chain on a given edge list (works on a host-only context)."""
        edges = np.ascontiguousarray(edges, dtype=np.uint64)
        cid = np.full(max(self.n, 1), -1, dtype=np.int32)
        order = np.full(max(self.n, 1), -1, dtype=np.int32)
        self.member_rank = np.zeros(max(self.n, 1), dtype=np.int32)
        stats = N.ClinkageStats()
        st = N.lib.hmk_clinkage_from_edges(self._h, _ptr(edges, C.c_uint64), edges.size, _ptr(cid, C.c_int32),
                                           _ptr(order, C.c_int32), _ptr(self.member_rank, C.c_int32), C.byref(stats))
        if st == N.HMK_ERR_REFERENCE_WOULD_CRASH:
            raise ReferenceWouldCrash(N.lib.hmk_last_error(self._h).decode(), 0, -1)
        if st:
            self._raise(st)
        return cid[:self.n], order[:stats.n_result_clusters], stats

    def greedy_phases(self):
        """hmk_greedy_last_phases: per-phase milliseconds of the last greedy_cluster / greedy_from_edges_dev call."""
        ph = N.GreedyPhases()
        st = N.lib.hmk_greedy_last_phases(self._h, C.byref(ph))
        if st:
            self._raise(st)
        return ph.as_dict()

    def greedy_from_edges(self, edges, symmetric, threshold, max_clusters):
        edges = np.ascontiguousarray(edges, dtype=np.uint64)
        cid = np.full(max(self.n, 1), -1, dtype=np.int32)
        order = np.full(max(self.n, 1), -1, dtype=np.int32)
        stats = N.GreedyStats()
        self.member_rank = np.zeros(max(self.n, 1), dtype=np.int32)
        st = N.lib.hmk_greedy_from_edges(self._h, _ptr(edges, C.c_uint64), edges.size, int(bool(symmetric)),
                                         int(threshold), int(max_clusters), _ptr(cid, C.c_int32),
                                         _ptr(order, C.c_int32), _ptr(self.member_rank, C.c_int32), C.byref(stats))
        if st:
            self._raise(st, stats)
        return cid[:self.n], order[:stats.n_result_clusters], stats

    def greedy_from_edges_dev(self, d_edges_ptr, n_edges, symmetric, max_clusters):
        """hmk_greedy_from_edges_dev: packed edges in device memory -> clusters (CSR built on the device)."""
        cid = np.full(max(self.n, 1), -1, dtype=np.int32)
        order = np.full(max(self.n, 1), -1, dtype=np.int32)
        stats = N.GreedyStats()
        self.member_rank = np.zeros(max(self.n, 1), dtype=np.int32)
        st = N.lib.hmk_greedy_from_edges_dev(self._h, C.c_void_p(d_edges_ptr), int(n_edges), int(bool(symmetric)),
                                             int(max_clusters), _ptr(cid, C.c_int32), _ptr(order, C.c_int32),
                                             _ptr(self.member_rank, C.c_int32), C.byref(stats))
        if st:
            self._raise(st, stats)
        return cid[:self.n], order[:stats.n_result_clusters], stats


# -----------------------------------------------------------------------------------------
# mirrors of the reference classes
# -----------------------------------------------------------------------------------------
class UniqueSequence:
    """UniqueSequence.java:19-171."""

    def __init__(self, sequence: str, labelsMap=None):
        self.labelsMap = dict(labelsMap) if labelsMap is not None else {"no_label": 1}  # :65-68
        self.sequence = encode(sequence)

    def size(self):  # :82-88
        return int(sum(self.labelsMap.values()))

    def getSequence(self):
        return self.sequence

    def getSequenceString(self):  # :103-109
        return "".join(AMINO_ACIDS[int(i)] for i in self.sequence)

    def getLabelsMap(self):
        return self.labelsMap

    def __eq__(self, other):  # :143-153
        return isinstance(other, UniqueSequence) and np.array_equal(self.sequence, other.sequence)

    def __hash__(self):
        return hash(self.sequence.tobytes())

    def __repr__(self):
        return f"UniqueSequence({self.getSequenceString()!r}, {self.labelsMap})"


class Cluster:
    """Cluster.java:21-204 (member list, id, size)."""

    def __init__(self, sequences, id):
        self.sequences = list(sequences)
        self.id = int(id)
        self._size = sum(s.size() for s in self.sequences)

    def insert(self, sequence):  # :50-63
        if sequence in self.sequences:
            raise DataException(f"Trying to insert unique sequence {sequence.getSequenceString()} into cluster "
                                f"{self.id}, which already contains this sequence. ")
        self.sequences.append(sequence)
        self._size += sequence.size()

    def insertAll(self, sequences):  # :70-74
        for s in list(sequences):
            self.insert(s)

    def getSequences(self):
        return self.sequences

    def getId(self):
        return self.id

    def size(self):  # :156-158
        return self._size

    def getUniqueSize(self):  # :113-115
        return len(self.sequences)

    def __repr__(self):
        return f"Cluster(id={self.id}, unique={self.getUniqueSize()}, size={self._size})"


class _GpuScorer:
    def __init__(self, scoringMatrix, device=0):
        self.scoringMatrix = np.asarray(scoringMatrix, dtype=np.int32).reshape(24, 24)
        self._ctx = Context(self.scoringMatrix, device)

    def _score(self, seq1, seq2):
        self._ctx.set_sequences([seq1.sequence, seq2.sequence])
        return int(self._pairs([0], [1])[0])


class ShiftedScorer(_GpuScorer):
    """ShiftedScorer.java:12-114: ShiftedScorer(scoringMatrix, shiftPenalty, maxShift)."""

    def __init__(self, scoringMatrix, shiftPenalty, maxShift, device=0):
        super().__init__(scoringMatrix, device)
        self.shiftPenalty = int(shiftPenalty)
        self.maxShift = int(maxShift)

    def _pairs(self, i, j):
        return self._ctx.score_pairs_shifted(i, j, self.maxShift, self.shiftPenalty)

    def sequenceScore(self, seq1, seq2):  # :98-100; throws DataException (:59-62)
        return self._score(seq1, seq2)

    def scoreWithShift(self, seq1, seq2):  # :48-95 -> (score, shift), AligningScorerResult
        self._ctx.set_sequences([seq1.sequence, seq2.sequence])
        score, shift = self._ctx.score_with_shift([0], [1], self.maxShift, self.shiftPenalty)
        return int(score[0]), int(shift[0])


class LocalAlignmentScorer(_GpuScorer):
    """LocalAlignmentScorer.java:10-155: (scoringMatrix, gapOpenPenalty, gapExtendPenalty)."""

    def __init__(self, scoringMatrix, gapOpenPenalty, gapExtendPenalty, device=0):
        super().__init__(scoringMatrix, device)
        self.gapOpenPenalty = int(gapOpenPenalty)
        self.gapExtendPenalty = int(gapExtendPenalty)

    def _pairs(self, i, j):
        return self._ctx.score_pairs_local(i, j, self.gapOpenPenalty, self.gapExtendPenalty)

    def sequenceScore(self, seq1, seq2):  # :27-29
        return self._score(seq1, seq2)


class HipClinkageSequenceClusterer:
    """Drop-in for ClinkageSequenceClusterer(sequenceScorer, threshold) (ClinkageSequenceClusterer.java:29-33): same
    ``cluster(List<UniqueSequence>) -> List<Cluster>`` contract (:43-124) -- exact complete linkage, cluster ids as the
    reference assigns them (singletons index + 1, merged clusters n + 2, ... in merge order), the returned list in the
    iteration order of the reference's HashSet, members in getSequences() order."""

    def __init__(self, sequenceScorer: "ShiftedScorer", threshold):
        if not isinstance(sequenceScorer, ShiftedScorer):
            raise TypeError("the GPU clinkage path takes a ShiftedScorer (Hammock.java:458)")
        self.sequenceScorer = sequenceScorer
        self.threshold = int(threshold)
        self.stats = None

    def cluster(self, sequences):
        sc = self.sequenceScorer
        ctx = sc._ctx
        ctx.set_sequences([s.sequence for s in sequences], sizes=[s.size() for s in sequences])
        cid, order, stats = ctx.clinkage_cluster(sc.maxShift, sc.shiftPenalty, self.threshold)
        self.stats = stats
        members = {}
        for k, c in enumerate(cid.tolist()):
            members.setdefault(c, []).append(k)
        result = []
        for c in order.tolist():
            ks = sorted(members[c], key=lambda k: int(ctx.member_rank[k]))
            result.append(Cluster([sequences[k] for k in ks], c))
        return result


class HipGreedySequenceClusterer:
    """Drop-in for LimitedGreedySequenceClusterer(sequenceScorer, threshold, maxClusters)
    (LimitedGreedySequenceClusterer.java:22) -- same constructor shape, same
    ``cluster(List<UniqueSequence>) -> List<Cluster>`` contract (:39-69): clusters with
    more than one member first (in creation order, id = seed index), then the
    remaining singletons."""

    def __init__(self, sequenceScorer: ShiftedScorer, threshold, maxClusters):
        if not isinstance(sequenceScorer, ShiftedScorer):
            raise TypeError("the GPU greedy path takes a ShiftedScorer (Hammock.java:402)")
        self.sequenceScorer = sequenceScorer
        self.threshold = int(threshold)
        self.maxClusters = int(maxClusters)
        self.stats = None

    def cluster(self, sequences):
        sc = self.sequenceScorer
        ctx = sc._ctx
        ctx.set_sequences([s.sequence for s in sequences], sizes=[s.size() for s in sequences])
        cid, order, stats = ctx.greedy_cluster(sc.maxShift, sc.shiftPenalty, self.threshold, self.maxClusters)
        self.stats = stats
        members = {}
        for k, c in enumerate(cid.tolist()):
            members.setdefault(c, []).append(k)
        result = []
        for c in order.tolist():
            ks = members[c]
            ks.sort(key=lambda k: int(ctx.member_rank[k]))  # Cluster.getSequences() insertion order
            cl = Cluster([sequences[ks[0]]], c)
            for k in ks[1:]:
                cl.insert(sequences[k])
            result.append(cl)
        return result
