"""ctypes binding of libhammock_hip.so (include/hammock_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C hammock_amd/csrc``.  There is no fallback of any kind: if the shared
object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhammock_hip.so")

HMK_OK = 0
HMK_ERR_BAD_ARG = 1
HMK_ERR_SHIFT_TOO_BIG = 2
HMK_ERR_DEVICE = 3
HMK_ERR_OOM = 4
HMK_ERR_REFERENCE_WOULD_CRASH = 5
HMK_ERR_CAPACITY = 6
HMK_ERR_NO_SEQUENCES = 7
HMK_EDGE_SHARDS = 64
HMK_MAX_LEN = 32

# every symbol include/hammock_hip.h declares
SYMBOLS = [
    "hmk_abi_version", "hmk_last_kernel_ms", "hmk_create", "hmk_create_multi", "hmk_device_count", "hmk_destroy", "hmk_last_error", "hmk_set_sequences",
    "hmk_score_pairs_shifted", "hmk_score_with_shift", "hmk_score_pairs_local", "hmk_score_block_shifted", "hmk_score_block_local",
    "hmk_neighbors_shifted", "hmk_neighbors_local", "hmk_neighbors_shifted_dev", "hmk_compact_edges_dev", "hmk_pack_rows_dev", "hmk_unpack_rows_dev",
    "hmk_neighbors_last_plan",
    "hmk_greedy_cluster", "hmk_greedy_from_edges", "hmk_greedy_from_edges_dev", "hmk_greedy_last_phases",
    "hmk_clinkage_cluster", "hmk_clinkage_from_edges", "hmk_set_java_hashset", "hmk_reserve",
]


class NeighborStats(C.Structure):
    _fields_ = [
        ("n_edges", C.c_uint64), ("pairs_scored", C.c_uint64), ("n_tiles", C.c_uint32),
        ("symmetric", C.c_uint32), ("classes_u8", C.c_uint32), ("classes_u16", C.c_uint32),
        ("classes_direct", C.c_uint32), ("classes_rows", C.c_uint32), ("kernel_ms", C.c_double),
    ]


class GreedyStats(C.Structure):
    _fields_ = [
        ("n_edges", C.c_uint64), ("phase1_stop_index", C.c_int32), ("phase1_clusters", C.c_int32),
        ("phase1_orphans", C.c_int32), ("crash_case", C.c_int32), ("crash_index", C.c_int32),
        ("n_result_clusters", C.c_int32), ("n_multi", C.c_int32), ("reserved", C.c_int32),
        ("neighbors_ms", C.c_double), ("greedy_ms", C.c_double),
    ]


class ClinkageStats(C.Structure):
    _fields_ = [("n_edges", C.c_uint64), ("merges", C.c_int32), ("searches", C.c_int32), ("n_result_clusters", C.c_int32),
                ("reserved", C.c_int32), ("neighbors_ms", C.c_double), ("chain_ms", C.c_double)]


class GreedyPhases(C.Structure):
    """hmk_greedy_phases: where the time of the last greedy call went (milliseconds)."""
    _fields_ = [(k, C.c_double) for k in ("plan_ms", "score_ms", "csr_ms", "wait_rows_ms", "phase1_ms", "precheck_ms", "exchange_ms",
                                          "device_loop_ms", "host_precheck_ms", "sequential_ms", "total_ms")] + \
               [("cand_entries", C.c_uint64), ("band_bytes", C.c_uint64), ("loop_rounds", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hammock_amd/csrc`.  hammock_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so
    # (same SONAME, libamdhip64.so.7).  If torch is importable, load it FIRST so that
    # our NEEDED libamdhip64.so.7 resolves to the copy torch uses; two copies in one
    # process cannot both open the GPU ("no ROCm-capable device is detected").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
    p_i32, p_u32, p_u64, p_u8 = (C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint8))
    L.hmk_abi_version.restype = i32
    L.hmk_last_kernel_ms.argtypes = [vp]
    L.hmk_last_kernel_ms.restype = C.c_double
    L.hmk_create.argtypes = [p_i32, i32, C.POINTER(vp)]
    L.hmk_create_multi.argtypes = [p_i32, C.POINTER(C.c_int), i32, C.POINTER(vp)]
    L.hmk_device_count.argtypes = [vp]
    L.hmk_destroy.argtypes = [vp]
    L.hmk_destroy.restype = None
    L.hmk_last_error.argtypes = [vp]
    L.hmk_last_error.restype = C.c_char_p
    L.hmk_set_sequences.argtypes = [vp, p_u8, p_u32, p_i32, u32]
    L.hmk_score_pairs_shifted.argtypes = [vp, p_u32, p_u32, u64, i32, i32, p_i32]
    L.hmk_score_pairs_local.argtypes = [vp, p_u32, p_u32, u64, i32, i32, p_i32]
    L.hmk_score_with_shift.argtypes = [vp, p_u32, p_u32, u64, i32, i32, p_i32, p_i32]
    L.hmk_score_block_shifted.argtypes = [vp, u32, u32, u32, u32, i32, i32, p_i32]
    L.hmk_score_block_local.argtypes = [vp, u32, u32, u32, u32, i32, i32, p_i32]
    L.hmk_neighbors_shifted.argtypes = [vp, i32, i32, i32, u32, u32, p_u64, u64, p_u64, C.POINTER(NeighborStats)]
    L.hmk_neighbors_local.argtypes = [vp, i32, i32, i32, u32, u32, p_u64, u64, p_u64, C.POINTER(NeighborStats)]
    L.hmk_neighbors_shifted_dev.argtypes = [vp, i32, i32, i32, u32, u32, vp, u64, vp, vp]
    L.hmk_compact_edges_dev.argtypes = [vp, vp, u64, vp, vp, u64, vp, vp]
    L.hmk_pack_rows_dev.argtypes = [vp, vp, u64, vp, i32, vp, vp, u64, vp]
    L.hmk_unpack_rows_dev.argtypes = [vp, vp, vp, i32, vp, u64, vp]
    L.hmk_neighbors_last_plan.argtypes = [vp, C.POINTER(NeighborStats)]
    L.hmk_greedy_cluster.argtypes = [vp, i32, i32, i32, i32, p_i32, p_i32, p_i32, C.POINTER(GreedyStats)]
    L.hmk_greedy_from_edges_dev.argtypes = [vp, vp, u64, i32, i32, p_i32, p_i32, p_i32, C.POINTER(GreedyStats)]
    L.hmk_greedy_from_edges.argtypes = [vp, p_u64, u64, i32, i32, i32, p_i32, p_i32, p_i32, C.POINTER(GreedyStats)]
    L.hmk_greedy_last_phases.argtypes = [vp, C.POINTER(GreedyPhases)]
    L.hmk_set_java_hashset.argtypes = [vp, i32]
    L.hmk_reserve.argtypes = [vp, u32]
    L.hmk_clinkage_cluster.argtypes = [vp, i32, i32, i32, p_i32, p_i32, p_i32, C.POINTER(ClinkageStats)]
    L.hmk_clinkage_from_edges.argtypes = [vp, p_u64, u64, p_i32, p_i32, p_i32, C.POINTER(ClinkageStats)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("hmk_destroy", "hmk_last_error", "hmk_last_kernel_ms"):
            fn.restype = i32
    return L


lib = _load()
