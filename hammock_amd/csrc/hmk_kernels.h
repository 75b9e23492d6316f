// hmk_kernels.h -- launchers of the HIP kernels (k_neighbors.hip, k_local.hip, k_edges.hip, k_pairs.hip), used by the host side (hmk_pass.cpp, hmk_cluster.cpp, hmk_multi.cpp, hmk_api.cpp).
#ifndef HMK_KERNELS_H
#define HMK_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "hmk_internal.h"

namespace hmk {

// rows per tile the SWAR kernel instantiation (lbmax, nw) was built with
int swar_rows_per_tile(int lbmax, int nw, bool exact);
// smallest instantiated column-length capacity that holds columns of length lb
int swar_lbmax_for(int lb);

hipError_t launch_neighbors_swar(int lbmax, int nw, bool exact, const NeighborParams &P,
                                 uint32_t tile_base, uint32_t n_tiles, hipStream_t s);
// row-packed kernels (k_neighbors_rows.hip): 8 rows per 8-byte table entry, one accumulator pair per shift.
// exact: a set of one length lb (its own instantiation); else the capacity form, rows_cap_for(lb) >= lb.
bool rows_kernel_available(int X, int la, int lb, bool exact);
int rows_cap_for(int lb);
int rows_per_tile_rows(int X, int d, int cap, bool exact);   // rows per tile of that instantiation
hipError_t launch_neighbors_rows(int X, int d, int cap, bool exact, const NeighborParams &P, uint32_t tile_base,
                                 uint32_t n_tiles, hipStream_t s);
hipError_t warm_neighbors_rows_module();
hipError_t launch_neighbors_direct(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles,
                                   const int32_t *d_matrix, int max_shift, int shift_penalty, int threshold,
                                   hipStream_t s);
// scorer 0: ShiftedScorer(a = maxShift, b = shiftPenalty); 1: LocalAlignmentScorer(a = gapOpen, b = gapExtend).
// pi == nullptr: dense block mode, pair k = (r0 + k / width, c0 + k % width).
hipError_t launch_probe_spin(unsigned long long *when, long long ticks, hipStream_t s);   // when[0] / [1]: start / end, 100 MHz wall clock
hipError_t launch_pairs(int scorer, const uint8_t *res32, const uint8_t *len, const int32_t *d_matrix,
                        const uint32_t *pi, const uint32_t *pj, uint64_t n_pairs, uint32_t r0, uint32_t c0,
                        uint32_t width, int a, int b, int32_t *out, int32_t *out_shift, hipStream_t s);

// LocalAlignmentScorer all ordered pairs, thresholded (tiles of one (row length, column length) class, lpad 32)
// force_signed / force_unpacked: the test switches HMK_LOCAL_SIGNED / HMK_LOCAL_NO_PK
hipError_t launch_neighbors_local(int lbmax, bool enc, bool force_signed, bool force_unpacked, const NeighborParams &P, uint32_t tile_base,
                                  uint32_t n_tiles, const int32_t *d_matrix, int gap_open, int gap_extend, int threshold, hipStream_t s);

// the same pass with the literal DP: any gap penalties, any matrix range
hipError_t launch_neighbors_local_literal(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, const int32_t *d_matrix,
                                          int gap_open, int gap_extend, int threshold, hipStream_t s);

// edge segments -> CSR (start[row_limit + 1], adj[]) on the device; rows at and beyond row_limit are left out
// (row_limit = n: the whole graph).  deg: zeroed uint32[row_limit]; cursor: zeroed uint32[2 row_limit], whose first half
// holds the rows' upper-neighbour counts ("up[]") after the scatter; score_range: device int[3] = {min score, max
// score, invalid edges}
// (segments first .. first + count - 1 of a pass: all of them by default)
EdgeSegs shard_segments(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t first = 0,
                        uint32_t count = HMK_EDGE_SHARDS);
hipError_t launch_csr_degree_scan(const EdgeSegs &segs, uint32_t n, uint32_t row_limit, bool symmetric, uint32_t *deg,
                                  uint64_t *start, uint64_t *tile_scratch, int *score_range, hipStream_t s, uint32_t row_lo = 0, uint32_t scan_rows = 0);
// (only rows [row_lo, row_limit) are counted; scan_rows != 0: start[] gets that many rows + 1 -- a piece of a multi-device call)
hipError_t launch_csr_scan_only(const uint32_t *deg, const uint32_t *deg_lo, uint64_t *start, uint32_t n, uint64_t *tile_scratch,
                                int *score_range, hipStream_t s);
hipError_t launch_add_u32(uint32_t *dst, const uint32_t *src, uint32_t n, hipStream_t s);   // dst[k] += src[k]
// multi-device calls (rows [d * rows_per, (d + 1) * rows_per) belong to device d): a shard's edges dealt into one block per owning device
// -- cnt / off / cur: device uint64[HMK_MAX_DEVICES + 1] each; block d = out[off[d] .. off[d + 1]) -- and the owner's degree counters
// (its own + the g - 1 slices the others sent; rows it does not own cleared)
hipError_t launch_route_edges(const EdgeSegs &segs, uint32_t rows_per, uint32_t g, unsigned long long *cnt, unsigned long long *off, unsigned long long *cur,
                              uint64_t *out, uint64_t out_cap, hipStream_t s);
hipError_t launch_owned_degrees(uint32_t *deg, uint32_t n, uint32_t r0, uint32_t r1, const uint32_t *const *slices, uint32_t n_slices, hipStream_t s);
size_t scan_scratch_bytes(uint32_t n);       // bytes of tile_scratch for n counters
size_t pack_rows_scratch_bytes(uint32_t n);  // bytes of launch_pack_rows' scratch
// adj: Nbr[] or, if packed, NbrPacked[] = m << 8 | (score - base)
size_t csr_partition_scratch_bytes();
hipError_t launch_csr_scatter_partitioned(const EdgeSegs &segs, const uint64_t *start, uint32_t *cursor, void *adj, int base, uint32_t n,
                                          uint64_t *recs, void *scratch, const uint32_t *row_lower, int forced_shift, hipStream_t s,
                                          uint32_t row_lo = 0, uint32_t row_hi = 0xFFFFFFFFu);   // rows [row_lo, row_hi) only: what a device of a multi-device call owns
hipError_t launch_csr_scatter(const EdgeSegs &segs, bool symmetric, const uint64_t *start, uint32_t *cursor, void *adj,
                              bool packed, int base, uint32_t row_limit, hipStream_t s, uint32_t n = 0, uint32_t row_lo = 0);

// the band, prepared for phase 1 (BandPack, hmk_internal.h; k_band_* in k_edges.hip).  bstart / bup / badj: the band's CSR (packed
// entries, rows [upper | lower]); fdeg, fcur, owner_of: zeroed uint32[n]; totals: zeroed uint32[4] ([0] near entries, [1] entries of the
// travelling lists); h_*: the HOST's pinned block by its device-visible addresses -- the kernels store the pack there themselves (near
// and tr each hold at most `entries`); every other array is sized by the caller.
hipError_t launch_band_prepare(const uint64_t *bstart, const uint32_t *bup, const void *badj, uint32_t R, uint64_t entries, uint32_t n, uint32_t ft,
                               uint32_t tr_per_row, const int32_t *seq_size, uint32_t *near_cnt, uint32_t *near_up, uint32_t *near_start, uint32_t *far_top,
                               uint8_t *far_more, uint32_t *fdeg, uint32_t *fcur, uint32_t *totals, uint32_t *fstart, uint32_t *fadj,
                               uint32_t *tr_cnt, uint32_t *tr_start, uint64_t tr_cap, uint32_t *h_near_start, uint32_t *h_near_up, uint32_t *h_far_top,
                               uint8_t *h_far_more, uint32_t *h_near, uint32_t *h_tr_cnt, uint32_t *h_tr_start, uint32_t *h_tr, uint32_t nt,
                               uint32_t *near_top, uint32_t *h_near_top, hipStream_t s);

hipError_t launch_compact_edges(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts,
                                uint64_t *out, uint64_t out_capacity, unsigned long long *total, hipStream_t s);

// edge segments <-> "row blocks" (uint32 row_start[n + 2], uint32 adj[] = m << 8 | score - threshold);
// scratch: pack_rows_scratch_bytes(n), used in stream order
hipError_t launch_pack_rows(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t n,
                            int threshold, uint32_t *scratch, uint32_t *row_start, uint32_t *adj, uint64_t adj_capacity,
                            hipStream_t s);
hipError_t launch_unpack_rows(const uint32_t *row_start, const uint32_t *adj, uint32_t n, int threshold, uint64_t *out,
                              uint64_t out_capacity, hipStream_t s);

// pre-check of the greedy merge's second loop on the device-resident adjacency (k_edges.hip)
// in_cluster: one bit per sequence (launch_cluster_bitmap), bitmap = uint32[(n + 31) / 32]
hipError_t launch_cluster_bitmap(const int32_t *cluster_of, uint32_t n, uint32_t *bitmap, hipStream_t s);
// mode: 0 count (cand_cnt), 1 fill at the prefix sums cand_start, 2 single pass (blocks of entries taken from the
// HMK_PRE_REGIONS counters total[], which the caller zeroes, one per region of `capacity` entries of cand[]; cand_start[q] /
// cand_cnt[q] describe leftover q's block; a counter above capacity: nothing usable was written;
// retry != null: a first stage with small tables, retry[nl] / *retry_count (zeroed) take the rows that need the large ones)
constexpr uint32_t HMK_PRE_REGIONS = 256;
hipError_t launch_greedy_precheck(int mode, bool packed, const uint64_t *start, const void *adj, const int32_t *cluster_of,
                                  const uint32_t *in_cluster, const int32_t *usize, const uint32_t *leftover, uint32_t nl, uint32_t *cand_cnt,
                                  uint32_t *cand_start, GreedyCand *cand, uint32_t *overflow, unsigned long long *total,
                                  unsigned long long capacity, uint32_t *retry, uint32_t *retry_count, int first_stage_slots, hipStream_t s,
                                  uint32_t own_lo = 0, uint32_t own_hi = 0xFFFFFFFFu, uint32_t region_base = 0, uint32_t region_count = 0);
// (own_lo / own_hi: only the leftovers with an id in that range; region_base / region_count: the regions of cand[] this launch fills, 0 = all)
hipError_t launch_scan_u32(const uint32_t *counts, uint32_t *start, uint32_t n, uint64_t *tile_scratch, hipStream_t s);
// where launch_scan_u32 leaves the 64-bit grand total inside tile_scratch (the uint32 start[n] wraps beyond 2^32 - 1)
size_t scan_total_index(uint32_t n);
// device-side second loop (k_loop_*).  Subscriber lists: per cluster the (leftover, candidate entry) pairs listing it,
// subs = uint32[2 * entries]; pass 0 (fill = false) counts into the zeroed cursor[n_clusters], pass 1 fills.
hipError_t launch_loop_subscribers(bool fill, uint32_t nl, const uint32_t *cand_start, const uint32_t *cand_cnt, const GreedyCand *cand,
                                   uint32_t *cursor, const uint32_t *sub_start, uint64_t *subs, hipStream_t s);
// one round; counters: device uint32[4] ([3] = tentative joiners the round's eval saw: 0 means the loop is over),
// first / first_next: uint32[n_clusters] each; first must be all ones, first_next is reset for the next round
hipError_t launch_loop_round(bool packed, const RowPieces &rows, const uint32_t *leftover,
                             uint32_t nl, const uint32_t *cand_start, const uint32_t *cand_cnt, GreedyCand *cand, uint8_t *status,
                             uint32_t *choice, uint32_t *lists2, uint32_t *dirty, uint32_t round, uint32_t *first, uint32_t *taken,
                             uint32_t *cursor, uint32_t n_clusters, int passes, uint32_t *accepted, int32_t *join_slot,
                             const uint32_t *sub_start, const uint64_t *subs, void *clusters, const int32_t *seq_size,
                             uint32_t *counters, unsigned long long *host_word, int chain_mode, hipStream_t s);
hipError_t launch_loop_sort_subscribers(uint32_t n_clusters, const uint32_t *sub_start, uint64_t *subs, uint64_t *tmp, hipStream_t s);
// clusters: 16 bytes per cluster {joined = 0, id, size}, built on the device from the uploaded ids and sizes
hipError_t launch_loop_init(uint32_t n_clusters, const long long *csize, const int32_t *cid, void *cl, const uint32_t *sub_start,
                            uint32_t *cursor, uint32_t nl, uint32_t *list, uint32_t *dirty, uint32_t *counters, uint32_t *taken,
                            uint8_t *status, int32_t *join_slot, hipStream_t s);

// LocalAlignmentScorer dense block, register-resident DP (needs |M| <= 127, gap penalties <= 0, len <= lbmax)
// enc: the tagged-max DP (needs |M| <= 31 and -31 <= gap penalties <= 0)
hipError_t launch_local_block(int lbmax, bool enc, bool force_signed, bool force_unpacked, const uint8_t *res32, const uint8_t *len, const int32_t *d_matrix, uint32_t r0,
                              uint32_t r1, uint32_t c0, uint32_t c1, int gap_open, int gap_extend, int32_t *out,
                              hipStream_t s);

// force the deferred load of the code objects a clustering call launches from (hmk_create)
hipError_t warm_neighbors_module();
hipError_t warm_edges_module();

}  // namespace hmk
#endif
