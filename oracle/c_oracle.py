"""ctypes binding of the C oracle (oracle/hammock_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()``
and ``bench.py``'s cpu_baseline leg.  ``hammock_amd`` never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhammock_oracle.so")

HMO_OK = 0
HMO_ERR_BAD_ARG = 1
HMO_ERR_SHIFT_TOO_BIG = 2
HMO_ERR_OOM = 4
HMO_ERR_REFERENCE_WOULD_CRASH = 5
SCORER_SHIFTED = 0
SCORER_LOCAL = 1
ORDER = {"size": 0, "alphabetic": 1, "input": 2}


class GreedyStats(C.Structure):
    _fields_ = [
        ("score_calls_phase1", C.c_uint64),
        ("score_calls_phase2", C.c_uint64),
        ("phase1_stop_index", C.c_int32),
        ("phase1_clusters", C.c_int32),
        ("phase1_orphans", C.c_int32),
        ("crash_case", C.c_int32),
        ("crash_index", C.c_int32),
        ("n_result_clusters", C.c_int32),
        ("n_multi", C.c_int32),
    ]


class ClinkageStats(C.Structure):
    _fields_ = [("score_calls", C.c_uint64), ("merges", C.c_int32), ("searches", C.c_int32),
                ("n_result_clusters", C.c_int32), ("reserved", C.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "hammock_oracle.c")
    hdr = os.path.join(_HERE, "hammock_oracle.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        p8 = C.POINTER(C.c_uint8)
        p32 = C.POINTER(C.c_int32)
        pu32 = C.POINTER(C.c_uint32)
        L.hmo_shifted_score.argtypes = [p32, p8, C.c_int, p8, C.c_int, C.c_int, C.c_int, p32, p32]
        L.hmo_shifted_score.restype = C.c_int
        L.hmo_local_score.argtypes = [p32, p8, C.c_int, p8, C.c_int, C.c_int, C.c_int]
        L.hmo_local_score.restype = C.c_int32
        L.hmo_score_pairs.argtypes = [p32, p8, pu32, pu32, pu32, C.c_uint64, C.c_int, C.c_int, C.c_int, p32]
        L.hmo_score_pairs.restype = C.c_int
        L.hmo_greedy_cluster.argtypes = [p32, p8, pu32, p32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, p32, p32, p32, C.POINTER(GreedyStats)]
        L.hmo_greedy_cluster.restype = C.c_int
        L.hmo_clinkage_cluster.argtypes = [p32, p8, pu32, p32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, p32, p32, p32,
                                           C.POINTER(ClinkageStats)]
        L.hmo_clinkage_cluster.restype = C.c_int
        L.hmo_sort_order.argtypes = [p8, pu32, p32, C.c_uint32, C.c_int, pu32]
        L.hmo_sort_order.restype = C.c_int
        L.hmo_synth.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, p8, pu32]
        L.hmo_synth.restype = C.c_int
        L.hmo_encode_residue.argtypes = [C.c_char]
        L.hmo_encode_residue.restype = C.c_int
        _lib = L
    return _lib


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def as_matrix(M):
    M = np.ascontiguousarray(np.asarray(M, dtype=np.int32).reshape(24, 24))
    return M


def encode(s):
    L = lib()
    out = np.empty(len(s), dtype=np.uint8)
    for k, ch in enumerate(s):
        r = L.hmo_encode_residue(ch.encode("ascii"))
        if r < 0:
            raise ValueError(f"invalid residue {ch!r}")
        out[k] = r
    return out


def pack(seqs):
    """list of str or uint8 arrays -> (res uint8, off uint32[n+1])."""
    arrs = [encode(s) if isinstance(s, str) else np.asarray(s, dtype=np.uint8) for s in seqs]
    off = np.zeros(len(arrs) + 1, dtype=np.uint32)
    if arrs:
        off[1:] = np.cumsum([len(a) for a in arrs])
    res = np.concatenate(arrs).astype(np.uint8) if arrs else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(res), off


def shifted_score(M, s1, s2, max_shift, shift_penalty):
    """-> (status, score, shift)"""
    L = lib()
    M = as_matrix(M)
    a = encode(s1) if isinstance(s1, str) else np.ascontiguousarray(s1, dtype=np.uint8)
    b = encode(s2) if isinstance(s2, str) else np.ascontiguousarray(s2, dtype=np.uint8)
    score = C.c_int32(0)
    shift = C.c_int32(0)
    st = L.hmo_shifted_score(_p(M, C.c_int32), _p(a, C.c_uint8), len(a), _p(b, C.c_uint8), len(b),
                             max_shift, shift_penalty, C.byref(score), C.byref(shift))
    return st, score.value, shift.value


def local_score(M, s1, s2, gap_open, gap_extend):
    L = lib()
    M = as_matrix(M)
    a = encode(s1) if isinstance(s1, str) else np.ascontiguousarray(s1, dtype=np.uint8)
    b = encode(s2) if isinstance(s2, str) else np.ascontiguousarray(s2, dtype=np.uint8)
    return L.hmo_local_score(_p(M, C.c_int32), _p(a, C.c_uint8), len(a), _p(b, C.c_uint8), len(b),
                             gap_open, gap_extend)


def score_pairs(M, res, off, i, j, scorer, a, b):
    L = lib()
    M = as_matrix(M)
    i = np.ascontiguousarray(i, dtype=np.uint32)
    j = np.ascontiguousarray(j, dtype=np.uint32)
    out = np.empty(len(i), dtype=np.int32)
    st = L.hmo_score_pairs(_p(M, C.c_int32), _p(res, C.c_uint8), _p(off, C.c_uint32), _p(i, C.c_uint32),
                           _p(j, C.c_uint32), len(i), scorer, a, b, _p(out, C.c_int32))
    return st, out


def score_block(M, res, off, rows, cols, scorer, a, b):
    """score(seq1=rows[r], seq2=cols[c]) for every (r, c) -> int32 [len(rows), len(cols)]."""
    rows = np.asarray(rows, dtype=np.uint32)
    cols = np.asarray(cols, dtype=np.uint32)
    ii = np.repeat(rows, len(cols))
    jj = np.tile(cols, len(rows))
    st, out = score_pairs(M, res, off, ii, jj, scorer, a, b)
    return st, out.reshape(len(rows), len(cols))


def greedy_cluster(M, res, off, size, scorer, a, b, threshold, max_clusters, n_threads=1):
    """-> (status, cluster_id[n], result_order[n_result], stats); stats.member_rank[n] = position of every
    sequence inside Cluster.getSequences() of its cluster (insertion order)"""
    L = lib()
    M = as_matrix(M)
    n = len(off) - 1
    cid = np.full(max(n, 1), -1, dtype=np.int32)
    order = np.full(max(n, 1), -1, dtype=np.int32)
    stats = GreedyStats()
    rank = np.zeros(max(n, 1), dtype=np.int32)
    sp = None
    if size is not None:
        size = np.ascontiguousarray(size, dtype=np.int32)
        sp = _p(size, C.c_int32)
    st = L.hmo_greedy_cluster(_p(M, C.c_int32), _p(res, C.c_uint8), _p(off, C.c_uint32), sp, n, scorer, a, b,
                              threshold, max_clusters, n_threads, _p(cid, C.c_int32), _p(order, C.c_int32),
                              _p(rank, C.c_int32), C.byref(stats))
    stats.member_rank = rank[:n]
    return st, cid[:n], order[:stats.n_result_clusters], stats


def set_java_hashset(version):
    """8 (default) / 7 / 6: the HashSet iteration order the clinkage restatement emulates (hammock_oracle.h)."""
    lib().hmo_set_java_hashset(int(version))


def clinkage_cluster(M, res, off, size, max_shift, shift_penalty, threshold, n_threads=1):
    """-> (status, cluster_id[n], result_order[n_result], member_rank[n], stats)"""
    L = lib()
    M = as_matrix(M)
    n = len(off) - 1
    cid = np.full(max(n, 1), -1, dtype=np.int32)
    order = np.full(max(n, 1), -1, dtype=np.int32)
    rank = np.zeros(max(n, 1), dtype=np.int32)
    stats = ClinkageStats()
    sp = None
    if size is not None:
        size = np.ascontiguousarray(size, dtype=np.int32)
        sp = _p(size, C.c_int32)
    st = L.hmo_clinkage_cluster(_p(M, C.c_int32), _p(res, C.c_uint8), _p(off, C.c_uint32), sp, n, max_shift, shift_penalty,
                                threshold, n_threads, _p(cid, C.c_int32), _p(order, C.c_int32), _p(rank, C.c_int32),
                                C.byref(stats))
    return st, cid[:n], order[:stats.n_result_clusters], rank[:n], stats


def sort_order(res, off, size, order):
    L = lib()
    n = len(off) - 1
    perm = np.empty(max(n, 1), dtype=np.uint32)
    sp = None
    if size is not None:
        size = np.ascontiguousarray(size, dtype=np.int32)
        sp = _p(size, C.c_int32)
    st = L.hmo_sort_order(_p(res, C.c_uint8), _p(off, C.c_uint32), sp, n, ORDER[order], _p(perm, C.c_uint32))
    if st:
        raise RuntimeError(f"hmo_sort_order status {st}")
    return perm[:n]


def synth(seed, n, len_lo, len_hi=None):
    L = lib()
    if len_hi is None:
        len_hi = len_lo
    res = np.empty(n * len_hi, dtype=np.uint8)
    off = np.empty(n + 1, dtype=np.uint32)
    st = L.hmo_synth(seed, n, len_lo, len_hi, _p(res, C.c_uint8), _p(off, C.c_uint32))
    if st:
        raise RuntimeError(f"hmo_synth status {st}")
    return np.ascontiguousarray(res[:off[n]]), off
