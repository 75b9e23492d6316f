// hmk_device.h -- device-side helpers shared by the kernel files of libhammock_hip.so (gfx950 only).
#ifndef HMK_DEVICE_H
#define HMK_DEVICE_H

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "hmk_internal.h"
#include "hmk_kernels.h"

namespace hmk {


typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// LDS (address space 3) accesses through a 32-bit byte address, so that the
// per-lane table offset and the compile-time row offset meet in ONE ds_read
// (VGPR address + immediate) with no flat-pointer arithmetic in between.
#define HMK_LDS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const HMK_LDS uint8_t *)p;
}
template <typename T>
__device__ __forceinline__ T lds_read(uint32_t addr) {
    return *reinterpret_cast<const HMK_LDS T *>((uintptr_t)addr);
}

// -----------------------------------------------------------------------------
// literal per-pair scorers (device), residues read through LDS pointers
// -----------------------------------------------------------------------------

// ShiftedScorer.scoreWithShift, ShiftedScorer.java:48-95.  M: int32[576] in LDS.
__device__ __forceinline__ int shifted_score_literal(const int *M, const uint8_t *seq1, int len1,
                                                     const uint8_t *seq2, int len2, int max_shift,
                                                     int shift_penalty, int *shift_out = nullptr) {
    const uint8_t *shorter, *longer;
    int slen, llen;
    if (len1 >= len2) { shorter = seq2; slen = len2; longer = seq1; llen = len1; }   // :51-57
    else              { shorter = seq1; slen = len1; longer = seq2; llen = len2; }
    int best = INT32_MIN;
    int best_shift = 0;                                                                // :65
    const int diff = llen - slen;                                                      // :66
    for (int s = -max_shift; s <= max_shift + diff; s++) {                             // :67
        int actual = 0;
        if (s <= 0) {                                                                  // :69-72
            for (int i = 0; i < slen + s; i++) actual += M[shorter[i - s] * 24 + longer[i]];
        } else {                                                                       // :73-77
            const int lim = min(slen, llen - s);
            for (int i = 0; i < lim; i++) actual += M[shorter[i] * 24 + longer[i + s]];
        }
        actual += diff * shift_penalty;                                                // :79
        if (s < 0) actual += -s * 2 * shift_penalty;                                   // :80-82
        if (s > diff) actual += (s - diff) * 2 * shift_penalty;                        // :83-85
        if (actual > best) { best = actual; best_shift = s; }                          // :86-89
    }
    if (shift_out) *shift_out = (len1 >= len2) ? best_shift : -best_shift;             // :91-93
    return best;
}

enum { DIR_LEFT = 0, DIR_UP = 1, DIR_DIAGONAL = 2, DIR_NOWHERE = 3 };

// LocalAlignmentScorer.fillDynamicMatrices, LocalAlignmentScorer.java:31-86.
// One DP line lives in LDS as row[col * stride], each cell packed (H << 2) | Dir.
__device__ __forceinline__ int local_score_literal(const int *M, const uint8_t *seq1, int len1,
                                                   const uint8_t *seq2, int len2, int gap_open,
                                                   int gap_extend, uint32_t *row, int stride) {
    // line 0: H = 0, Dir = LEFT (:97-100); column 0 of every line: H = 0, Dir = UP (:93-96)
    for (int col = 1; col <= len2; col++) row[col * stride] = DIR_LEFT;
    int global_max = 0;                                                                // :32
    for (int line = 1; line <= len1; line++) {                                         // :40
        const int a = seq1[line - 1] * 24;
        int left_h = 0, left_d = DIR_UP;   // cell [line][0]
        int diag_h = 0;                    // cell [line-1][0]
        for (int col = 1; col <= len2; col++) {                                        // :41
            const uint32_t upc = row[col * stride];
            const int up_h = (int)(upc >> 2), up_d = (int)(upc & 3u);
            const int up = up_h + (up_d == DIR_UP ? gap_extend : gap_open);            // :43-48,:57
            const int left = left_h + (left_d == DIR_LEFT ? gap_extend : gap_open);    // :50-55,:58
            const int diag = diag_h + M[a + seq2[col - 1]];                            // :59
            const int mx = max(diag, max(up, left));                                   // :61
            int h, d;
            if (mx < 0) { h = 0; d = DIR_NOWHERE; }                                    // :63-65
            else {
                h = mx;                                                                // :67
                global_max = max(global_max, mx);                                      // :68-72
                d = DIR_LEFT;                       // one of the three always matches
                if (mx == up) d = DIR_UP;                                              // :76-78
                if (mx == diag) d = DIR_DIAGONAL;                                      // :79-81
            }
            row[col * stride] = ((uint32_t)h << 2) | (uint32_t)d;
            diag_h = up_h;
            left_h = h;
            left_d = d;
        }
    }
    return global_max;                                                                 // :85
}

// per-lane LDS staging of one sequence: 9-dword stride keeps the byte reads of
// the 64 lanes on different banks
constexpr int SEQ_STRIDE_DW = 9;

__device__ __forceinline__ void stage_sequence(uint32_t *dst, const uint8_t *res32, uint32_t idx) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(res32 + (size_t)idx * 32);
    const u32x4 lo = src[0], hi = src[1];
    dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[3] = lo.w;
    dst[4] = hi.x; dst[5] = hi.y; dst[6] = hi.z; dst[7] = hi.w;
}

// -----------------------------------------------------------------------------
// wave-level edge staging shared by the neighbour kernels
// -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Wave priority (s_setprio) in the neighbour kernels.  A wave that is issuing table reads goes ahead of the waves of its SIMD
// that are testing accumulators or staging hits, so the LDS queue -- the unit that bounds these kernels -- is fed first
// (length-12 shift-packed kernel: 3.329 -> 3.297 ms with priority 2; 1 and 3 gave 3.304 / 3.309 ms).  A wave that drains its
// stage goes ahead of everything: the drain is a chain of round trips, and at the lowest priority its few instructions waited
// behind every other wave's VALU work.  A tile's table build runs at priority 3 as well: until its tables stand a workgroup
// contributes no table reads at all (mixed lengths, a build every 28 batches: 5.29 -> 5.21 ms).
constexpr int PRIO_READ = 2, PRIO_BUILD = 3, PRIO_DRAIN = 3;
__device__ __forceinline__ void read_phase_begin(bool on) { if (on) __builtin_amdgcn_s_setprio(PRIO_READ); }
__device__ __forceinline__ void read_phase_end(bool on) { if (on) __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void build_begin() { __builtin_amdgcn_s_setprio(PRIO_BUILD); }   // a tile's table build
__device__ __forceinline__ void build_end() { __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void drain_begin() { __builtin_amdgcn_s_setprio(PRIO_DRAIN); }
__device__ __forceinline__ void drain_end() { __builtin_amdgcn_s_setprio(0); }

// What a flush does beside storing the edge: nothing, or counting the rows' degrees (NeighborParams::deg: the CSR build's first
// pass, fused into the scoring of a clustering call; fire-and-forget atomics).  EDGES_RUNTIME: decided by P.deg at run time.
// Only STORED edges are counted: an edge dropped by a segment overflow must not be, or the CSR sized from the counters (its
// scatter is enqueued before the host notices the overflow) would expect entries that are not there.
enum { EDGES_PLAIN = 0, EDGES_RUNTIME = 1, EDGES_COUNT = 2 };
template <int MODE = EDGES_RUNTIME>
__device__ __forceinline__ void place_edge(const NeighborParams &P, uint32_t x, uint32_t m) {
    if (MODE == EDGES_COUNT || (MODE == EDGES_RUNTIME && P.deg)) {   // (run time: wave-uniform)
        atomicAdd(&P.deg[x], 1u);
        if (P.symmetric) atomicAdd(&P.deg[P.deg_m_offset + m], 1u);
    }
}

// Which output segment a tile writes: each of the HMK_EDGE_SHARDS segments has its own cursor (returning atomics on ONE address
// from 256 CUs are served one after the other, ~100 ns each).
__device__ __forceinline__ uint32_t tile_shard(uint32_t tile) { return tile % HMK_EDGE_SHARDS; }

// Drains one wave's staged records to its output segment.  REC_DW dwords per
// record: [0] column (sorted position), [1] row within the tile, [2..] the NW
// accumulator dwords (SWAR) or the score itself (direct, NW == 0).
template <int NW, bool DEG = true>
__device__ __forceinline__ void flush_stage(const HMK_LDS uint32_t *stage, uint32_t cnt, const NeighborParams &P,
                                            const Tile &T, int g, bool lane16, uint32_t shard) {
    constexpr int REC_DW = (NW == 0) ? 3 : NW + 2;
    if (cnt == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // staged ds_writes land before the reads below
    // threadIdx.x lives in v0 for the whole kernel; an mbcnt-derived lane id is hoisted out of the tile loop by the
    // compiler and then SPILLED to scratch there (8 bytes per lane and tile: +45 MB of HBM writes on the 10^5 pass)
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&P.counts[shard], (unsigned long long)cnt);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const HMK_LDS uint32_t *rec = stage + k * REC_DW;
        const uint32_t col = rec[0], r = rec[1];
        int score;
        if (NW == 0) {
            score = (int)rec[2];
        } else {
            uint32_t mx = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t dw = rec[2 + w];
                if (lane16) {
                    mx = max(mx, max(dw & 0xFFFFu, dw >> 16));
                } else {
                    mx = max(mx, max(max(dw & 0xFFu, (dw >> 8) & 0xFFu), max((dw >> 16) & 0xFFu, dw >> 24)));
                }
            }
            score = (int)mx - g;
        }
        uint32_t x = T.row0 + r, m = col;
        if (!P.perm_identity) { x = P.perm[x]; m = P.perm[m]; }   // wave-uniform branch
        if (P.row_is_m || (P.symmetric && x > m)) { const uint32_t t = x; x = m; m = t; }
        const unsigned long long pos = base + k;
        if (pos < P.cap_per_shard) {
            P.edges[(unsigned long long)shard * P.cap_per_shard + pos] =
                ((unsigned long long)x << 40) | ((unsigned long long)m << 16) | (unsigned long long)((uint32_t)score & 0xFFFFu);
            // degrees of STORED edges only: an edge dropped by a segment overflow must not be counted, or the CSR built
            // from the counters (its scatter is enqueued before the host notices the overflow) would be sized for edges that
            // are not there and overrun the adjacency buffer
            if (DEG) place_edge<EDGES_RUNTIME>(P, x, m);   // stored edges only
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // reads done before the stage is reused
}

// Compact staging (the exact 8-bit lane kernel): one dword per hit, (column - tile's first column) | row << 16 |
// (score - threshold) << 20.  The lane proof bounds score - threshold to 0..127 (lane = 128 - threshold + score <= 255) and a
// tile has at most 65,536 columns and 16 rows, so the record is exact; staging it is a ds_write_b32 (2 LDS-array cycles per
// wave-instruction) where the 16-byte {column, row, accumulators} record was a ds_write_b128 (8): with a hit in ~15 % of the
// wave iterations that was 5 % of the kernel's LDS cycles (SQ_LDS_IDX_ACTIVE against 2 x the lookups).
template <int MODE>
__device__ __forceinline__ void flush_stage_compact(const HMK_LDS uint32_t *stage, uint32_t cnt, const NeighborParams &P,
                                                    const Tile &T, int threshold, uint32_t shard) {
    if (cnt == 0) return;
    drain_begin();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // staged ds_writes land before the reads below
    const uint32_t lane = threadIdx.x & 63u;                // (not mbcnt: see flush_stage)
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&P.counts[shard], (unsigned long long)cnt);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t rec = stage[k];
        const int score = (int)(rec >> 20) + threshold;
        uint32_t x = T.row0 + ((rec >> 16) & 0xFu), m = T.col0 + (rec & 0xFFFFu);
        if (!P.perm_identity) { x = P.perm[x]; m = P.perm[m]; }   // wave-uniform branch
        if (P.symmetric && x > m) { const uint32_t t = x; x = m; m = t; }
        const unsigned long long pos = base + k;
        if (pos < P.cap_per_shard) {
            P.edges[(unsigned long long)shard * P.cap_per_shard + pos] =
                ((unsigned long long)x << 40) | ((unsigned long long)m << 16) | (unsigned long long)((uint32_t)score & 0xFFFFu);
            if (MODE != EDGES_PLAIN) place_edge<MODE>(P, x, m);   // stored edges only
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // reads done before the stage is reused
    drain_end();
}

// The same for the plane kernels, whose lanes may be 16 bits wide: `rec_dw` (wave-uniform) is 1 for the record above with
// score = (record >> 20) + base_score, or 2 for {(column - first column) | row << 16, score} when score - threshold may
// not fit 12 bits (16-bit lanes).  The LONGER length bucket supplies a tile's rows, so P.row_is_m may swap the roles.
template <bool DEG>
__device__ __forceinline__ void flush_stage_packed(const HMK_LDS uint32_t *stage, uint32_t cnt, const NeighborParams &P,
                                                   const Tile &T, int base_score, uint32_t rec_dw, uint32_t shard) {
    if (cnt == 0) return;
    drain_begin();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // staged ds_writes land before the reads below
    const uint32_t lane = threadIdx.x & 63u;                // (not mbcnt: see flush_stage)
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&P.counts[shard], (unsigned long long)cnt);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t rec = stage[k * rec_dw];
        const int score = rec_dw == 1 ? (int)(rec >> 20) + base_score : (int)stage[k * 2 + 1];
        uint32_t x = T.row0 + ((rec >> 16) & 0xFu), m = T.col0 + (rec & 0xFFFFu);
        if (!P.perm_identity) { x = P.perm[x]; m = P.perm[m]; }   // wave-uniform branch
        if (P.row_is_m || (P.symmetric && x > m)) { const uint32_t t = x; x = m; m = t; }
        const unsigned long long pos = base + k;
        if (pos < P.cap_per_shard) {
            P.edges[(unsigned long long)shard * P.cap_per_shard + pos] =
                ((unsigned long long)x << 40) | ((unsigned long long)m << 16) | (unsigned long long)((uint32_t)score & 0xFFFFu);
            if (DEG) place_edge<EDGES_RUNTIME>(P, x, m);   // stored edges only
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // reads done before the stage is reused
    drain_end();
}

// largest of the eight byte lanes of two accumulator dwords
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t max_byte_lane(uint32_t w0, uint32_t w1) {
    const uint32_t M = 0x00FF00FFu;
    u16x2 a = __builtin_bit_cast(u16x2, w0 & M), b = __builtin_bit_cast(u16x2, (w0 >> 8) & M);
    u16x2 c = __builtin_bit_cast(u16x2, w1 & M), d = __builtin_bit_cast(u16x2, (w1 >> 8) & M);
    const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
    return max((uint32_t)m.x, (uint32_t)m.y);
}

// one table entry of NW dwords from LDS byte address `addr`
template <int NW>
__device__ __forceinline__ void lds_read_entry(uint32_t addr, uint32_t (&e)[NW]) {
    if constexpr (NW == 1) {
        e[0] = lds_read<uint32_t>(addr);
    } else if constexpr (NW == 2) {
        const u32x2 v = lds_read<u32x2>(addr);
        e[0] = v.x; e[1] = v.y;
    } else {
#pragma unroll
        for (int q = 0; q < NW / 4; q++) {
            const u32x4 v = lds_read<u32x4>(addr + 16 * q);
            e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
        }
    }
}

// One atomicAdd per distinct key of a wave instead of one per lane (a wave's 64 consecutive edges come
// from a handful of tile rows).  wave_groups() finds, without touching memory, each lane's group
// (lanes holding the same key): the group's first lane, this lane's rank in it and the group size.
// Must be called by all 64 lanes (wave-uniform control flow); lanes with valid == false take no part.
struct WaveGroup { uint32_t leader, rank, size; };
__device__ __forceinline__ WaveGroup wave_groups(uint32_t key, bool valid) {
    const uint32_t lane = threadIdx.x & 63;
    WaveGroup g{lane, 0, 0};
    uint64_t todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const bool mine = valid && key == k0;
        const uint64_t same = __ballot(mine);
        if (mine) {
            g.leader = (uint32_t)leader;
            g.rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            g.size = (uint32_t)__popcll(same);
        }
        todo &= ~same;
    }
    return g;
}

}  // namespace hmk
#endif
