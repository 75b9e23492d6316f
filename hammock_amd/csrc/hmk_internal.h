// hmk_internal.h -- structures shared by the host API, the host greedy merge
// and the HIP kernels of libhammock_hip.so.  Not part of the public ABI.
#ifndef HMK_INTERNAL_H
#define HMK_INTERNAL_H

#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/hammock_hip.h"

namespace hmk {

// ---- neighbour kernel plan ---------------------------------------------------
// Sequences are bucketed by length ("sorted order"); a tile is R rows x a run
// of columns inside ONE (row length, column length) class, so everything that
// depends on the two lengths is uniform over the workgroup.

enum { PATH_U8 = 0, PATH_U16 = 1, PATH_DIRECT = 2, PATH_ROWS = 3 };   // PATH_ROWS: launch groups only (8-bit lanes, k_neighbors_rows.hip)

struct TileClass {
    uint8_t la, lb;    // row / column sequence length
    uint8_t nd;        // diagonals (shifts) = 2X + |la - lb| + 1
    uint8_t case_b;    // 1: the COLUMN sequence is the shorter one (S of ShiftedScorer.java:51-57)
    uint8_t x;         // max shift
    uint8_t path;      // PATH_*
    uint8_t nw;        // dwords per table entry (1, 2, 4 or 8)
    uint8_t pad;
    int32_t g;         // lane value = g + score  (g = 128 - thr or 32768 - thr)
    int32_t d;         // |la - lb|
    uint32_t cinit[8]; // initial accumulator dwords: per lane g + penalty(s) - bias * cells(s)
};

struct Tile {
    uint32_t row0, nrows;  // rows [row0, row0 + nrows) in sorted order, nrows <= R
    uint32_t col0, ncols;  // columns [col0, col0 + ncols)
    uint32_t cls;          // index into the TileClass array
    uint32_t diag;         // 1: rows and columns come from the same bucket: keep col > row only
    uint32_t pad0, pad1;
};

struct NeighborParams {
    const uint8_t *res_sorted;   // [n][lpad] residues in sorted order, zero padded
    const uint32_t *perm;        // sorted position -> caller index
    const uint8_t *mb;           // biased matrix bytes [576]: M + bias
    const TileClass *classes;
    const Tile *tiles;
    uint64_t *edges;             // HMK_EDGE_SHARDS segments of cap_per_shard
    unsigned long long *counts;  // [HMK_EDGE_SHARDS]
    uint64_t cap_per_shard;
    uint32_t n_tiles;
    uint32_t lpad;               // 16 or 32
    uint32_t symmetric;          // 1: emit (min, max) caller indices
    uint32_t row_is_m;           // 1: the tile's ROW is seq1 (= m of the edge), LocalAlignmentScorer tiles
    uint32_t perm_identity;      // 1: sorted position == caller index (one length bucket, no reordering): skip the perm loads
    // optional (may be null): the rows' degrees, counted while the edges are written (the CSR build's first pass, fused into the
    // scoring of a clustering call) with fire-and-forget atomics.  Zeroed by the caller.
    uint32_t *deg;       // total degrees -- or, with deg_m_offset = n, upper counts (the edge's smaller end) in deg[0, n) and lower counts in deg[n, 2n)
    uint32_t deg_m_offset;   // the larger end m of a symmetric edge counts into deg[deg_m_offset + m] (0: one counter per row)
};

// one directed neighbour: sequenceScore(seq1 = m, seq2 = x) = s for the row x it is stored under
struct Nbr {
    uint32_t m;
    int32_t s;
    uint32_t id() const { return m; }
    int32_t score() const { return s; }
};
// the same in 4 bytes: m << 8 | (s - base), usable when every stored score is within base .. base + 255.
// The merge only compares scores, so the common offset does not matter; half the bytes to copy from the
// device and to walk on the host.
struct NbrPacked {
    uint32_t v;
    uint32_t id() const { return v >> 8; }
    int32_t score() const { return (int32_t)(v & 0xFFu); }
};

// A cluster a leftover sequence could still join after phase 1: every member of cluster c is a neighbour of the
// sequence; mn = the lowest of those scores; covered counts, during the second loop, the members that joined
// later and are neighbours too (starts at 0).
struct GreedyCand { int32_t c, mn, covered; };

// A run of packed edges in device memory: min(*count, cap) entries at `edges`.  The CSR kernels take a short list of
// them: the HMK_EDGE_SHARDS segments of one neighbour pass, or those plus the blocks gathered from other devices.
struct EdgeSeg { const uint64_t *edges; const unsigned long long *count; uint64_t cap; };
constexpr uint32_t HMK_MAX_SEGS = HMK_EDGE_SHARDS + 16;
struct EdgeSegs { EdgeSeg s[HMK_MAX_SEGS]; uint32_t n; };

// Where the adjacency rows of a clustering call live on the device side: one piece (this device's CSR), or -- a multi-device call --
// one piece per device, device d holding the rows [d * rows_per, (d + 1) * rows_per) (its start[] / up[] are indexed by the row itself,
// rows it does not own are empty).  The second loop's kernels run on the root and read a joiner's row where it lives.
constexpr uint32_t HMK_MAX_DEVICES = 16;
struct RowPieces {
    const uint64_t *start[HMK_MAX_DEVICES];
    const uint32_t *up[HMK_MAX_DEVICES];
    const void *adj[HMK_MAX_DEVICES];
    uint32_t rows_per;   // 0: one piece
};

// Optional device-side pre-check of the second loop (hmk_cluster.cpp provides it when the adjacency is still resident on
// the GPU): given cluster_of[n] (-1 = none), the clusters' member counts and the leftover list, fill the candidate
// CSR (cand_start[nl + 1], cand[]).  Returns false if it could not (the merge then fetches the whole adjacency and runs
// its threaded host version).
using GreedyPrecheck = std::function<bool(const int32_t *cluster_of, const std::vector<int32_t> &usize,
                                          const std::vector<uint32_t> &leftover,
                                          std::vector<uint32_t> &cand_start, std::vector<GreedyCand> &cand)>;

// The band of a clustering call as the device prepares it for phase 1 (firstPhase, LimitedGreedySequenceClusterer.java:77-120;
// hmk_cluster.cpp, k_band_* in k_edges.hip): what a step of the loop needs of row x < rows, split by where the neighbour lies.
//   near        the neighbours with an id below `rows` (band rows themselves): near[near_start[x] .. near_start[x + 1]), the
//               first near_up[x] of them with an id above x.  Entries are id << 8 | (score - base), as in the packed adjacency.
//   near_top    the best NEAR_T of the row's near neighbours ABOVE it by the same key as far_top, best first, ~0u = no more: the first
//               one still free is the best near candidate of :93; only when all listed ones have been taken (and the row has more)
//               does the host scan the near row's leading section.  May be null (then it always scans).
//   far_top     the row's best FAR neighbours (id >= rows) by the reference's key (score, Cluster.size(), smaller id --
//               ClinkageSequenceClusterer.java:166-173, :275-289), best first, far_t per row, ~0u = no more; far_more[x] != 0:
//               the row has far neighbours beyond the far_t listed.  A far sequence changes state in phase 1 only by being
//               absorbed (:99-101, :108-110), so the first entry that is still free IS the best far candidate of :93.
//   tr          for the row's first TR_PER_ROW far candidates c, slot u = TR_PER_ROW * x + t: the LATER band rows that have both x
//               and c as neighbours, tr[tr_start[u] .. + tr_cnt[u]), entries band row << 8 | (min(score(x, row), score(c, row)) -
//               base): the rows the new cluster {x, c} is feasible for, with its score (ClinkageClusterScorer.java:36-48) --
//               intersected on the device (k_band_isect).  tr_cnt[u] == ~0u: not prepared (a row or list beyond the kernel's
//               table / room); then, and for a far candidate beyond the first TR_PER_ROW, c's whole list comes through
//               GreedyHooks::far_row and the host filters it against x's row.
struct BandPack {
    static constexpr uint32_t TR_PER_ROW = 4;   // (2: a dozen lists per 25,000 steps had to be fetched on demand at 10^6 -- and a small copy issued while
                                                // the pass runs waits milliseconds for a slot)
    static constexpr uint32_t NEAR_T = 4;       // near_top entries per row
    uint32_t rows = 0, far_t = 0;
    const uint32_t *near_start = nullptr, *near_up = nullptr, *near = nullptr;
    const uint32_t *far_top = nullptr, *near_top = nullptr;
    const uint8_t *far_more = nullptr;
    const uint32_t *tr_cnt = nullptr, *tr_start = nullptr, *tr = nullptr;
};

// test / diagnostic options of the host merge (the library reads its environment switches once per call, hmk_ctx.h Switches)
struct GreedyOptions {
    int phase1_threads = 0;   // > 0: HMK_PHASE1_THREADS (any input, that many threads for the window scans)
    int phase1_window = 0;    // > 0: HMK_PHASE1_WINDOW (positions per window)
    bool timing = false;      // HMK_GREEDY_TIMING: the phases' times on stderr
    // HMK_PHASE1_HOST_BAND=rows[,far_t] (greedy_from_edges only, symmetric scores): phase 1 runs on a BandPack that is built on
    // the HOST from the whole graph -- the literal statement of what the device's k_band_* kernels must produce, and the CPU test
    // of the incremental phase 1 against the oracle
    int host_band_rows = 0, host_band_far_t = 0;
};

// Hooks of the host merge for a caller that still has the adjacency on the device (hmk_cluster.cpp):
//   precheck    see GreedyPrecheck (may be empty)
//   band_pack   the prepared band (blocks until it is on the host), or null: phase 1 then reads whole rows (need_rows)
//   far_row     band neighbours of far sequence `id` (entries band row << 8 | score - base), fetched from the device on demand
//   band_far    the far part of band row x (entries id << 8 | score - base), on demand
//   need_rows   only some rows may be in host memory yet: need_rows(k, from) returns R > k once start[from .. R] and
//               adj[start[from] .. start[R]) are valid on the host (it fetches more rows from the device if it has to; adj_base
//               says where adj[0] would lie).  Phase 1 asks row by row -- from 0, or from the band's row limit once the
//               prepared band has served the rows below it; with the device pre-check the second loop needs no rows at all.
//               Empty = everything is there already.
// Optional device-side run of the whole second loop (hmk_cluster.cpp, k_loop_*): given the state after phase 1 --
// cluster_of[n], per cluster slot its member count, Cluster.size() and id, the leftover list -- fill join_slot[q] =
// the slot leftover q joins or -1.  Returns false if it did not run (the merge then uses precheck / its host loop).
using GreedyDeviceLoop = std::function<bool(const int32_t *cluster_of, const std::vector<int32_t> &usize,
                                            const std::vector<int64_t> &csize, const std::vector<int32_t> &cids,
                                            const std::vector<uint32_t> &leftover, std::vector<int32_t> &join_slot)>;

struct GreedyTimes { double phase1_ms, host_precheck_ms, sequential_ms; };   // host wall time of the merge's parts
struct GreedyHooks {
    GreedyPrecheck precheck;
    GreedyDeviceLoop device_loop;
    std::function<const BandPack *()> band_pack;
    std::function<bool(uint32_t, std::vector<uint32_t> &)> far_row, band_far;
    std::function<uint32_t(uint32_t, uint32_t)> need_rows;   // a return value <= k means the rows could not be had: the merge stops
    std::function<const void *()> adj_base;        // with need_rows: where adj[] is now (the host buffer may move when it grows)
    GreedyTimes *times = nullptr;
};
constexpr int HMK_INTERNAL_ROWS_FAILED = -1;   // greedy_from_csr*: need_rows failed (the caller knows why)

// CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota where there is one (a container on a
// 256-thread host may own 8 of them: std::thread::hardware_concurrency() reports the host's).  At least 1.  (hmk_greedy.cpp)
unsigned usable_cpus();

// host greedy merge (hmk_greedy.cpp)
// symmetric_scores: adj holds every edge under both ends with the same score (symmetric matrix)
// upper: NULL, or per row the number of leading entries whose id is above the row's own (the row is laid out
//        "upper neighbours first"); lets the join propagation skip the neighbours that are already decided
// hooks: NULL or the caller's device-side helpers (see GreedyHooks)
int greedy_from_csr(uint32_t n, const int32_t *sizes, const uint64_t *start, const Nbr *adj, const uint32_t *upper,
                    const GreedyHooks *hooks, bool symmetric_scores, int max_clusters,
                    int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st,
                    std::string *err, const GreedyOptions &opt = GreedyOptions());
int greedy_from_csr_packed(uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrPacked *adj,
                           const uint32_t *upper, const GreedyHooks *hooks, bool symmetric_scores, int max_clusters, int32_t *cluster_id, int32_t *result_order, int32_t *member_rank,
                           hmk_greedy_stats *st, std::string *err, const GreedyOptions &opt = GreedyOptions());
int greedy_from_edges(uint32_t n, const int32_t *sizes, const uint64_t *edges, uint64_t n_edges,
                      bool symmetric, int threshold, int max_clusters, int32_t *cluster_id,
                      int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st, std::string *err,
                      const GreedyOptions &opt = GreedyOptions());

// exact complete linkage by nearest-neighbour chain on a symmetric CSR adjacency (hmk_clinkage.cpp)
// hashset_version: 8 (Java 8+), 7 (JDK 7u6+) or 6 (JDK 6 / early 7): whose java.util.HashSet iteration order picks the chain
// starts and orders the returned list
int clinkage_from_csr(int hashset_version, uint32_t n, const int32_t *sizes, const uint64_t *start, const Nbr *adj, int32_t *cluster_id,
                      int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err);
int clinkage_from_csr_packed(int hashset_version, uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrPacked *adj, int32_t *cluster_id,
                             int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err);

}  // namespace hmk

#endif
