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
    // optional (may be null): the CSR the greedy tail builds from these edges is PLACED while the edges are written.
    // deg_up[x] counts row x's "upper" entries (symmetric: the neighbours with a larger index; else all of them), deg_lo[m]
    // row m's lower ones (symmetric only); both zeroed uint32[n].  The value an edge's atomicAdd returns is its place inside
    // the row's section: rank[2 * slot] / rank[2 * slot + 1] for the edge stored in slot = segment * cap_per_shard +
    // position.  The scatter that follows needs no atomics and no pass that counts degrees.  rank == null: deg[] just counts
    // the rows' total degrees (both ends of a symmetric edge), with fire-and-forget atomics -- at 10^6 the 2.5 x 10^9 returning
    // atomics cost the pass 10 % and the scatter there is bound by its random writes, not by its atomics.
    uint32_t *deg;       // counting mode: total degrees -- or, with deg_m_offset = n, upper counts in deg[0, n) and lower counts in deg[n, 2n)
    uint32_t deg_m_offset;   // counting mode: the larger end m of an edge counts into deg[deg_m_offset + m] (0: one counter per row)
    uint32_t shard_base;     // a tile writes into segment shard_base + tile % shard_mod (the whole pass: 0 and HMK_EDGE_SHARDS; a clustering
    uint32_t shard_mod;      // call scores its band tiles and the others at the same time, each into segments of their own)
    uint32_t band_mod;       // > 0: ONE launch scores band tiles (Tile::pad0) and the others; a band tile writes into segment tile % band_mod,
                             // the others into shard_base + tile % shard_mod (shard_base >= band_mod)
    uint32_t *band_counter;  // ... and every band tile's workgroup adds 1 here when its edges are out (band_tile_done, hmk_device.h)
    uint32_t *deg_up;    // placing mode: the rows' upper counters ...
    uint32_t *deg_lo;    // ... and lower counters (symmetric only)
    uint32_t *rank;      // placing mode (else null)
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

// One entry of a join-propagation list (k_greedy_prop): candidate entry k (index into the cand array) belongs to a later
// leftover that is a neighbour of the joining sequence, with that pair's score.
struct GreedyProp { uint32_t k; int32_t score; };

// A run of packed edges in device memory: min(*count, cap) entries at `edges`.  The CSR kernels take a short list of
// them: the HMK_EDGE_SHARDS segments of one neighbour pass, or those plus the blocks gathered from other devices.
struct EdgeSeg { const uint64_t *edges; const unsigned long long *count; uint64_t cap; };
constexpr uint32_t HMK_MAX_SEGS = HMK_EDGE_SHARDS + 16;
struct EdgeSegs { EdgeSeg s[HMK_MAX_SEGS]; uint32_t n; };

// Optional device-side pre-check of the second loop (hmk_cluster.cpp provides it when the adjacency is still resident on
// the GPU): given cluster_of[n] (-1 = none), the clusters' member counts and the leftover list, fill the candidate
// CSR (cand_start[nl + 1], cand[]) and, if want_prop, the join-propagation lists (prop_start[cand.size() + 1], prop[];
// *have_prop says whether they were produced).  Returns false if it could not (the merge then fetches the whole
// adjacency and runs its threaded host version).
using GreedyPrecheck = std::function<bool(const int32_t *cluster_of, const std::vector<int32_t> &usize,
                                          const std::vector<uint32_t> &leftover, bool want_prop,
                                          std::vector<uint32_t> &cand_start, std::vector<GreedyCand> &cand,
                                          std::vector<uint32_t> &prop_start, std::vector<GreedyProp> &prop, bool *have_prop)>;

// Hooks of the host merge for a caller that still has the adjacency on the device (hmk_cluster.cpp):
//   precheck    see GreedyPrecheck (may be empty)
//   need_rows   only a prefix of the rows may be in host memory yet: need_rows(k) returns R > k once start[0 .. R] and
//               adj[0 .. start[R]) are valid on the host (it fetches more rows from the device if it has to).  Phase 1
//               asks row by row; with the device pre-check and propagation lists the second loop needs no rows at all.
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
    std::function<uint32_t(uint32_t)> need_rows;   // a return value <= k means the rows could not be had: the merge stops
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
                    std::string *err);
int greedy_from_csr_packed(uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrPacked *adj,
                           const uint32_t *upper, const GreedyHooks *hooks, bool symmetric_scores, int max_clusters, int32_t *cluster_id, int32_t *result_order, int32_t *member_rank,
                           hmk_greedy_stats *st, std::string *err);
int greedy_from_edges(uint32_t n, const int32_t *sizes, const uint64_t *edges, uint64_t n_edges,
                      bool symmetric, int threshold, int max_clusters, int32_t *cluster_id,
                      int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st, std::string *err);

// exact complete linkage by nearest-neighbour chain on a symmetric CSR adjacency (hmk_clinkage.cpp)
// hashset_version: 8 (Java 8+), 7 (JDK 7u6+) or 6 (JDK 6 / early 7): whose java.util.HashSet iteration order picks the chain
// starts and orders the returned list
int clinkage_from_csr(int hashset_version, uint32_t n, const int32_t *sizes, const uint64_t *start, const Nbr *adj, int32_t *cluster_id,
                      int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err);
int clinkage_from_csr_packed(int hashset_version, uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrPacked *adj, int32_t *cluster_id,
                             int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err);

}  // namespace hmk

#endif
