// k_neighbors_rows.h -- all-vs-all ShiftedScorer (ShiftedScorer.java:48-95) with threshold -> edge list, "row-packed" form.
//
// The reference sums, for every shift s, the cells M[S[i - s]][L[i]] (s <= 0) or M[S[i]][L[i + s]] (s > 0) over the overlap
// (:67-77).  With the tile's ROW peptides as the longer-or-equal sequence and the lane's COLUMN peptide as the other one this
// reads: shift plane u (s = u - X) adds, for every column position j, the cell of row position i = j + u - X.  A cell depends
// on (row, i, column residue) only -- not on the shift -- so the workgroup keeps ONE table per group of 8 rows,
//
//     E[i][c] = 8 bytes: byte r = cell(row r, position i, residue c) + bias          (24 x 8 B = 192 B per position i)
//
// and a pair of (8 rows, 1 column) is scored by one ds_read_b64 per (u, j) inside the overlap + one v_add3 per two reads and
// dword: 8 rows advance together, one accumulator pair per shift plane.  Nothing is read that the reference does not add:
// 72 reads per 8 pairs at length 12 / max shift 3 = 72 B of LDS per pair.  The binding unit is the LDS byte rate (ds_read_b64:
// 256 B/clk/CU), so bytes are time.  A position's 24 entries are 48 consecutive dwords: any set of residues in a wave reads
// without a bank conflict (ds_read_b64 has 64 banks).
//
// Table reads are VOLATILE LDS loads: the compiler's load/store optimiser fuses two plain ds_read_b64 off one address register
// into a ds_read2_b64, which moves 128 B/clk/CU instead of 256 (MI355X_MICROARCH.md, LDS table); volatile loads are left alone
// and still take the immediate offset, so the tables are packed (192 bytes per row position).
//
// Lanes are 8 bits wide; hmk_plan.cpp (classify) proves per length class -- or per row bound -- that every lane stays in
// [0, 255]: lane = g + penalty(s) - bias * cells(s) + sum of biased cells, g = 128 - threshold, so "score >= threshold" is the
// lane's top bit.  Classes that do not fit, columns longer than the rows, and (X, D) pairs without an instantiation below run
// the kernels of k_neighbors.hip.
//
// Hits.  Two forms, chosen per shape (rows_inloop):
//   * more than ROWS_INLOOP_MAXCELLS cells per pair (10-mers and longer): the hit's score is cut out of the planes right where
//     they are in registers (one v_perm_b32 per plane) and staged with the record; the flush decodes and stores.
//   * short shapes (6- to 9-mers, the X = 1 capacity forms) are VALU-bound by themselves: a step only NOTES its hits in a
//     history word, the wave looks once per four steps, stages bare records, and the flush -- where every lane has a record of
//     its own -- fetches the record's column again and rescores it against its row group's table.
// What was measured on the way to this (fat records, prefetching the next column, a placing flush that ranked every edge inside
// its CSR row, spread-out tables, ...) is in DESIGN.md 5.7 with the commit that held the code.
//
// Integer scoring only: no MFMA, no dense contraction.
#ifndef HMK_NEIGHBORS_ROWS_H
#define HMK_NEIGHBORS_ROWS_H
#include "hmk_device.h"

namespace hmk {

// Records a wave stages before it drains them.  Every drain takes its place in the output segment with ONE returning atomic
// on the segment's cursor and pays a fixed cost (scalar loads, first gather): the short shapes, whose hits are dense at the
// reference's default thresholds (7-mers: 2.1 % of the pairs), take the largest stage; the capacity forms' tables are 4-5 x the
// one-length ones and LDS bounds their occupancy.
constexpr int ROWS_STAGE_CAPFORM = 320, ROWS_STAGE_EXACT = 640, ROWS_STAGE_SHORT = 1024;
constexpr int ROWS_INLOOP_MAXCELLS = 45;   // shapes of at most this many cells per pair keep the rescoring flush
constexpr int rows_cells(int x, int cap) { return cap * (2 * x + 1) - x * (x + 1); }   // cells per pair of equal lengths
constexpr bool rows_inloop(int x, int cap) { return rows_cells(x, cap) > ROWS_INLOOP_MAXCELLS; }
// ... of a shape: a one-length form whose rows are LONGER than its columns (d > 0: a class of a mixed-length set, k_neighbors_rows_lens)
// always extracts in the loop (its pairs have at least 44 cells)
constexpr bool rows_inloop_shape(int x, int d, int cap, bool exact) { return (exact && d > 0) || rows_inloop(x, cap); }
constexpr int rows_stage(int x, int d, int cap, bool exact) {
    return !exact ? ROWS_STAGE_CAPFORM : rows_inloop_shape(x, d, cap, exact) ? ROWS_STAGE_EXACT : ROWS_STAGE_SHORT;
}
// groups of 8 rows per tile.  Capacity forms (a length bucket's short column runs): 1 / 2 / 3 / 4 groups measured 4.87 / 4.48 /
// 4.58 / 4.74 ms on BASELINE config 4a.  One-length shapes: long column runs, one group (2.65-2.67 ms with 1 or 2 at length
// 12) -- but the short ones, where what a step does once per COLUMN (fetch, offsets) is a quarter of its VALU work and VALU is as
// busy as the LDS pipe, take two.
constexpr int rows_groups(int x, int /*d*/, int cap, bool exact) {
    return !exact ? 2 : rows_inloop(x, cap) ? 1 : 2;
}
// the rescoring flush of a two-group one-length shape drains ONE group per wave-instruction (two-ended stage, see the kernel)
constexpr bool rows_two_ended(int x, int d, int cap, bool exact, int g) { return exact && !rows_inloop_shape(x, d, cap, exact) && g == 2; }

template <typename T>
__device__ __forceinline__ T rows_table_read(uint32_t addr) {
    return *reinterpret_cast<const volatile HMK_LDS T *>((uintptr_t)addr);
}
// waves per SIMD a shape is compiled for: what its LDS footprint lets a CU hold, and no more than its registers (column
// offsets + 2 accumulators per plane + reads in flight and the rest) allow without spilling
// (the capacity forms keep a second offset set and run-time bounds alive: at 8 waves the smallest of them spilled 40 bytes
// per lane inside the batch loop; 7 waves: BASELINE config 4a 4.30 -> 4.21 ms, 6 waves 4.27)
constexpr int rows_waves(int nd, int cap, int lds_bytes, bool exact) {
    const int v = cap + 2 * nd + 36 + (exact ? 0 : 6);
    const int by_regs = v <= 64 ? 8 : v <= 72 ? 7 : v <= 80 ? 6 : v <= 96 ? 5 : 4;
    const int by_lds = 163840 / ((lds_bytes + 511) / 512 * 512);
    return by_lds < by_regs ? (by_lds < 1 ? 1 : by_lds) : by_regs;
}
constexpr int rows_tab_bytes(int x, int d, int cap, bool exact, int g) {
    return g * (cap + d + (exact ? 0 : 2 * x + d)) * 192;   // positions + (capacity form) the end table's ND - 1 positions
}
constexpr int rows_lds_bytes(int x, int d, int cap, bool exact, int g) {   // must match the kernel's LDS map
    return rows_tab_bytes(x, d, cap, exact, g) + 576 + 8 * g * 32 + 4 * rows_stage(x, d, cap, exact) * 4;
}

template <int N, class F, int... Is>
__device__ __forceinline__ void rows_static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void rows_static_for(F &&f) { rows_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// Everything that depends on the shape only: X max shift, D = row length - column length (>= 0), CAP column-length capacity
// (EXACT_LB: THE column length), G groups of 8 rows per tile.
template <int X, int D, int CAP, bool EXACT_LB, int G>
struct RowsShape {
    static constexpr int ND = 2 * X + D + 1;          // shift planes
    static constexpr int NI = CAP + D;                // row positions held per group
    static constexpr int ENT = 192;                   // bytes per row position: 24 residues x 8 rows
    static constexpr int NEND = EXACT_LB ? 0 : ND - 1;   // end table: the row's last ND - 1 positions again, indexed from the row's end
    static constexpr int TAB_BYTES = rows_tab_bytes(X, D, CAP, EXACT_LB, G);
    static constexpr int LPADW = (CAP <= 8) ? 2 : (CAP <= 16) ? 4 : 8; // residue dwords a lane loads (rows are P.lpad bytes apart)
    static constexpr int TW = (X + 3) / 4;            // dwords holding the last X residues of a column
    static constexpr int TWN = TW > 0 ? TW : 1;
    static constexpr int NT = X > 0 ? X : 1;
    static_assert(X >= 0 && D >= 0 && CAP >= 2 * X && CAP >= 1 && CAP <= 32 && G >= 1 && G <= 8, "shape");
    static_assert(TAB_BYTES <= 65536, "table offsets must fit the DS immediate");
    // byte address of row position i of group g / of the position e places before the row's end; position tables are linear
    // in g (GROUP_STEP bytes per group), which lets the flush add a per-record group to the offsets instead
    static constexpr int GROUP_STEP = NI * ENT;
    static constexpr int pos_addr(int g, int i) { return (g * NI + i) * ENT; }
    static constexpr int end_addr(int g, int e) { return (G * NI + g * NEND + e) * ENT; }

    // the column's residue words as they lie in memory (zero for a lane without a column); tw: the last X residues, wherever
    // the column ends (capacity form only: unaligned dword loads; the array is padded)
    static __device__ __forceinline__ void load_words(const uint8_t *rowp, bool live, int lbs, uint32_t (&words)[LPADW], uint32_t (&tw)[TWN]) {
#pragma unroll
        for (int q = 0; q < LPADW; q++) words[q] = 0;
#pragma unroll
        for (int q = 0; q < TWN; q++) tw[q] = 0;
        if (live) {
            if constexpr (LPADW == 2) {   // columns of at most 8 residues: half the bytes through the L2
                const u32x2 v0 = reinterpret_cast<const u32x2 *>(rowp)[0];
                words[0] = v0.x; words[1] = v0.y;
            } else {
                const u32x4 v0 = reinterpret_cast<const u32x4 *>(rowp)[0];
                words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
            }
            if constexpr (LPADW == 8) {
                const u32x4 v1 = reinterpret_cast<const u32x4 *>(rowp)[1];
                words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
            }
            if (!EXACT_LB && TW > 0) {
#pragma unroll
                for (int q = 0; q < TW; q++) __builtin_memcpy(&tw[q], rowp + (lbs - X) + 4 * q, 4);
            }
        }
    }
    // base + byte k of w in ONE instruction (sub-dword operand select); the plain C form costs a v_bfe and a v_add when the base
    // is a per-lane value (the flush: table + group) -- with a compile-time base (the main loop) the v_bfe alone
    template <int K>
    static __device__ __forceinline__ uint32_t add_byte(uint32_t base, uint32_t w) {
        uint32_t r;
        if constexpr (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(w));
        else if constexpr (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(w));
        else if constexpr (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(base), "v"(w));
        else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(base), "v"(w));
        return r;
    }
    // A column's residues -> table offsets (residue * 8): off[j] for position j, toff[q] for position lbs - X + q (the last X).
    // `base` is added to every offset (the table's LDS address; LANE_BASE: + a per-lane displacement), `tbase` to the tail ones.
    template <bool LANE_BASE = false>
    static __device__ __forceinline__ void offsets_of(const uint32_t (&words_in)[LPADW], const uint32_t (&tw_in)[TWN], uint32_t base, uint32_t tbase,
                                                      uint32_t (&off)[CAP], uint32_t (&toff)[NT]) {
        if constexpr (LANE_BASE) {
            // residues are < 32, so byte k of (word << 3) is residue * 8 exactly (the three bits that move in are zero)
            uint32_t words[LPADW];
#pragma unroll
            for (int q = 0; q < LPADW; q++) words[q] = words_in[q] << 3;
            rows_static_for<CAP>([&](auto jt) {
                constexpr int J = decltype(jt)::value;
                off[J] = add_byte<(J & 3)>(base, words[J >> 2]);
            });
        } else {
            // (byte select, then the shift: one v_lshlrev_b32_sdwa per residue -- no shifted copy of the word, no mask for byte 0;
            // the compiler finds the form itself for bytes 1-3 and turns byte 0 back into shift + mask, hence the asm for that one)
            const uint32_t three = 3u;   // (a VGPR operand: the same register the compiler keeps for its own SDWA shifts)
#pragma unroll
            for (int j = 0; j < CAP; j++) {
                if ((j & 3) == 0) {
                    uint32_t r;
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(words_in[j >> 2]));
                    off[j] = base + r;
                } else {
                    off[j] = base + (((words_in[j >> 2] >> ((j & 3) * 8)) & 0xFFu) << 3);
                }
            }
        }
        if (EXACT_LB) {
#pragma unroll
            for (int q = 0; q < X; q++) toff[q] = off[CAP - X + q];
        } else {
            uint32_t tw[TWN];
#pragma unroll
            for (int q = 0; q < TWN; q++) tw[q] = tw_in[q] << 3;
#pragma unroll
            for (int q = 0; q < X; q++) toff[q] = tbase + ((tw[q >> 2] >> ((q & 3) * 8)) & 0xFFu);
        }
    }
    static __device__ __forceinline__ void offsets(const uint8_t *rowp, bool live, int lbs, uint32_t base, uint32_t tbase,
                                                   uint32_t (&off)[CAP], uint32_t (&toff)[NT]) {
        uint32_t words[LPADW], tw[TWN];
        load_words(rowp, live, lbs, words, tw);
        offsets_of(words, tw, base, tbase, off, toff);
    }

    // All shift sums of (8 rows of group GI) x (this lane's column): W0 / W1[u] = the 8 byte lanes of plane u.
    template <int GI>
    static __device__ __forceinline__ void accumulate(const uint32_t (&off)[CAP], const uint32_t (&toff)[NT], int lbs,
                                                      const uint32_t (&ci)[ND], uint32_t (&W0)[ND], uint32_t (&W1)[ND]) {
        if constexpr (EXACT_LB) {
            // everything is known at compile time: plane by plane, the plane's reads summed two at a time (one v_add3 per
            // dword and pair; an odd count starts with a plain add): ceil(reads / 2) VALU instructions per dword, the minimum
#pragma unroll
            for (int u = 0; u < ND; u++) {
                constexpr int NMAIN = CAP - X;
                const int j0 = X - u > 0 ? X - u : 0;                 // main positions j0 .. NMAIN - 1
                const int nt = ND - 2 - u >= X ? X : (ND - 2 - u < 0 ? 0 : ND - 1 - u);   // tail positions q = 0 .. nt - 1 (u <= ND - 2 - q)
                const int n = NMAIN - j0 + nt;
                uint32_t a0 = ci[u], a1 = ci[u];
                // read k of the plane: k < NMAIN - j0: main position j0 + k; else tail position k - (NMAIN - j0)
                auto rd = [&](int k) {
                    const int j = k < NMAIN - j0 ? j0 + k : NMAIN + (k - (NMAIN - j0));   // column position (tail q = j - NMAIN)
                    return rows_table_read<u32x2>(off[j] + (uint32_t)pos_addr(GI, j + u - X));
                };
                int k = 0;
                if (n & 1) { const u32x2 e = rd(0); a0 += e.x; a1 += e.y; k = 1; }
#pragma unroll
                for (; k + 1 < n; k += 2) {
                    const u32x2 e0 = rd(k), e1 = rd(k + 1);
                    a0 = a0 + e0.x + e1.x; a1 = a1 + e0.y + e1.y;
                }
                W0[u] = a0; W1[u] = a1;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < ND; u++) { W0[u] = ci[u]; W1[u] = ci[u]; }
        // column position J, all planes that pair it with a row position >= 0: i = J + u - X
        auto add_pos = [&](auto jt) {
            constexpr int J = decltype(jt)::value;
#pragma unroll
            for (int u = (X - J > 0 ? X - J : 0); u < ND; u++) {
                const u32x2 e = rows_table_read<u32x2>(off[J] + (uint32_t)pos_addr(GI, J + u - X));
                W0[u] += e.x; W1[u] += e.y;
            }
        };
        // the main positions j < lbs - X (every plane's row position stays below the row's end), two at a time so that
        // the adds pair up into v_add3; an odd count takes position 0 on its own first (cf. k_neighbors_planes)
        auto add_pairs = [&](auto start_tag) {
            constexpr int J0 = decltype(start_tag)::value;
            int nmain = lbs - X;
            asm volatile("" : "+s"(nmain));   // keep the tests scalar (s_cmp + s_cbranch), see k_neighbors_planes
#pragma unroll
            for (int j = J0; j + 1 < CAP - X; j += 2) {
                if (j + 1 >= nmain) break;
                // both positions of a plane next to each other: one v_add3 per dword
#pragma unroll
                for (int u = 0; u < ND; u++) {
                    const bool v0 = u >= X - j, v1 = u >= X - (j + 1);
                    if (v0 && v1) {
                        const u32x2 e0 = rows_table_read<u32x2>(off[j] + (uint32_t)pos_addr(GI, j + u - X));
                        const u32x2 e1 = rows_table_read<u32x2>(off[j + 1] + (uint32_t)pos_addr(GI, j + 1 + u - X));
                        W0[u] = W0[u] + e0.x + e1.x; W1[u] = W1[u] + e0.y + e1.y;
                    } else if (v1) {
                        const u32x2 e1 = rows_table_read<u32x2>(off[j + 1] + (uint32_t)pos_addr(GI, j + 1 + u - X));
                        W0[u] += e1.x; W1[u] += e1.y;
                    }
                }
            }
        };
        if ((lbs - X) & 1) {
            add_pos(std::integral_constant<int, 0>{});
            add_pairs(std::integral_constant<int, 1>{});
        } else {
            add_pairs(std::integral_constant<int, 0>{});
        }
        // the last X column positions: position lbs - X + q pairs with row position lbs - 2X + q + u, which is inside the
        // row (length lbs + D) for planes u <= ND - 2 - q only; that row position is ND - 2 - q - u places before the row's
        // end, whatever the column length (end table)
#pragma unroll
        for (int u = 0; u < ND; u++) {
#pragma unroll
            for (int q = 0; q < X; q++) {
                if (u <= ND - 2 - q) {
                    const u32x2 e = rows_table_read<u32x2>(toff[q] + (uint32_t)end_addr(GI, ND - 2 - q - u));
                    W0[u] += e.x; W1[u] += e.y;
                }
            }
        }
    }
};

// The flush is a function that is NOT inlined: inlined, it (the rescoring kind scores whole columns again and wants registers of
// its own) made the compiler keep the append loop's state in scratch memory on every pass.  What it needs beside the wave's
// stage -- the kernel's arguments, the tile and its class -- it reads where the kernel reads them: the kernel passes the
// address of its kernarg segment (the intrinsic is null inside a callee), the workgroup id reaches a callee as an implicit
// scalar input, and everything behind them is uniform read-only memory, so all of it arrives by scalar loads in SGPRs and the
// function's vector registers are free for the rescoring.
typedef const __attribute__((address_space(4))) uint8_t *RowsConstBytes;
template <typename T>
__device__ __forceinline__ T rows_const_load(const void *p) {   // a uniform address in read-only memory: s_load
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>((uintptr_t)p);
}
struct RowsKernArgs { NeighborParams P; uint32_t tile_base; };   // k_neighbors_rows' explicit arguments as they lie in the kernarg segment

// Drains one wave's staged records.  MODE: what a flush does beside storing the edge (EDGES_PLAIN / EDGES_COUNT: the rows'
// degree counters of a clustering call, fire-and-forget atomics).
//   in-loop shapes:  record = column - tile's first column | row within the tile << 16 | (score - threshold) << 24.
//   rescoring shapes: record = column | the hit's bit in its lane's history word << 16 | group << 21; its score is worked out
//   here.  Two-ended stages (rows_two_ended): group 0's records are stage[0 .. cnt), group 1's stage[CAP - cnt_hi .. CAP), and a
//   wave-instruction of the rescoring touches ONE group's table: its ds_read_b64 (64 banks, a position's 24 entries are 48
//   dwords) are conflict-free for any set of residues.  (Round 4 read the 4-byte half that holds the record's row, ds_read_b32:
//   32 banks for the same 48 dwords, and with two groups' tables 336 dwords apart -- SQ_LDS_BANK_CONFLICT counted 1.9 extra
//   cycles per read at the 7-mers' default threshold, a tenth of the pass's CU cycles.)
template <int X, int D, int CAP, bool EXACT_LB, int G, int MODE>
__device__ __attribute__((noinline)) void flush_stage_rows(const HMK_LDS uint32_t *stage_v, uint32_t cnt_v, uint32_t cnt_hi_v, uint32_t tab_addr,
                                                           uint32_t ka_lo, uint32_t ka_hi) {
    using S = RowsShape<X, D, CAP, EXACT_LB, G>;
    constexpr bool TWO = rows_two_ended(X, D, CAP, EXACT_LB, G);
    constexpr int STAGE_CAP = rows_stage(X, D, CAP, EXACT_LB);
    // (arguments arrive in vector registers; all of them are wave-uniform)
    const uint32_t cnt_lo = __builtin_amdgcn_readfirstlane(cnt_v);
    const uint32_t cnt_hi = TWO ? __builtin_amdgcn_readfirstlane(cnt_hi_v) : 0u;
    const uint32_t cnt = cnt_lo + cnt_hi;
    if (cnt == 0) return;
    const HMK_LDS uint32_t *stage = (const HMK_LDS uint32_t *)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)stage_v);
    struct {
        const uint8_t *res_sorted; const uint32_t *perm; uint64_t *edges; unsigned long long *counts; uint64_t cap_per_shard;
        uint32_t *deg;
        uint32_t lpad, symmetric, perm_identity, row0, col0, shard, deg_m_offset;
        int threshold;
        uint32_t cinit[8];
    } A;
    int lbs;
    {
        const RowsConstBytes ka = (RowsConstBytes)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(ka_hi) << 32) |
                                                   (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(ka_lo));   // (the builtin returns int: no sign extension)
        const auto *K = reinterpret_cast<const __attribute__((address_space(4))) RowsKernArgs *>(ka);
        const uint32_t tile = K->tile_base + blockIdx.x;
        A.res_sorted = K->P.res_sorted; A.perm = K->P.perm; A.edges = K->P.edges; A.counts = K->P.counts;
        A.cap_per_shard = K->P.cap_per_shard;
        A.deg = K->P.deg;
        A.lpad = K->P.lpad; A.symmetric = K->P.symmetric; A.perm_identity = K->P.perm_identity; A.deg_m_offset = K->P.deg_m_offset;
        const Tile *Tp = K->P.tiles + tile;
        A.row0 = rows_const_load<uint32_t>(&Tp->row0);
        A.col0 = rows_const_load<uint32_t>(&Tp->col0);
        const TileClass *Cp = K->P.classes + rows_const_load<uint32_t>(&Tp->cls);
        A.shard = tile % HMK_EDGE_SHARDS;   // tile_shard(), hmk_device.h
        A.threshold = 128 - rows_const_load<int32_t>(&Cp->g);
#pragma unroll
        for (int q = 0; q < 8; q++) A.cinit[q] = (q * 4 < S::ND) ? rows_const_load<uint32_t>(&Cp->cinit[q]) : 0u;
        lbs = EXACT_LB ? CAP : (int)rows_const_load<uint8_t>(&Cp->lb);
    }
    drain_begin();
    asm volatile("" ::: "memory");   // (the wave's own staged ds_writes and the ds_reads below execute in order: a compiler barrier is all this takes)
    const uint32_t lane = threadIdx.x & 63u;                // (not mbcnt: see flush_stage, hmk_device.h)
    const int la = lbs + D;
    const uint32_t tab = __builtin_amdgcn_readfirstlane(tab_addr);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&A.counts[A.shard], (unsigned long long)cnt);
    // the rows' degree counters of a clustering call: fire-and-forget, nothing of these is waited for.  With the sorted order =
    // the caller's (one length bucket) the 64 records' smaller ends are the tile's 8 or 16 rows: one atomic per distinct row
    // with the group's size instead of one per record (10^5 clustering call: scoring 3.43 -> 3.27 ms).  Stored edges only.
    auto count_degrees = [&](uint32_t x, uint32_t m, bool ok) {
        if (MODE != EDGES_COUNT) return;
        if (A.perm_identity) {
            const WaveGroup g = wave_groups(x, ok);
            if (ok && g.rank == 0) atomicAdd(&A.deg[x], g.size);
        } else if (ok) {
            atomicAdd(&A.deg[x], 1u);
        }
        if (ok && A.symmetric) atomicAdd(&A.deg[A.deg_m_offset + m], 1u);
    };
    if constexpr (rows_inloop_shape(X, D, CAP, EXACT_LB)) {
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
        const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        base = ((unsigned long long)bhi << 32) | blo;
        for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
            const uint32_t k = k0 + lane;
            const bool live = k < cnt;
            const uint32_t rec = live ? stage[k] : 0u;
            uint32_t x = A.row0 + ((rec >> 16) & 0x3Fu), m = A.col0 + (rec & 0xFFFFu);
            if (!A.perm_identity && live) { x = A.perm[x]; m = A.perm[m]; }
            if (A.symmetric && x > m) { const uint32_t t = x; x = m; m = t; }
            const int score = A.threshold + (int)(rec >> 24);
            const unsigned long long pos = base + k;
            const bool ok = live && pos < A.cap_per_shard;
            count_degrees(x, m, ok);
            if (ok)
                A.edges[(unsigned long long)A.shard * A.cap_per_shard + pos] =
                    ((unsigned long long)x << 40) | ((unsigned long long)m << 16) | (unsigned long long)((uint32_t)score & 0xFFFFu);
        }
        asm volatile("" ::: "memory");
        drain_end();
        return;
    } else {
    // ---- rescoring: vector memory by hand --------------------------------------------------------------------------------
    // Left to the compiler, an iteration of this loop (64 records) is a chain of round trips: the records' columns, (mixed
    // lengths: the permutation,) and -- because the compiler does not count loads and stores in flight across a loop's back
    // edge and waits with vmcnt(0) wherever it needs one of them -- the edge stores' completion as well.  Here ONE asm
    // statement per iteration issues the loads for the NEXT 64 records (columns, tails, permutation) and waits for them
    // (cdna_hip_programming.md, inline asm: "the loads and their s_waitcnt in ONE statement", early-clobber outputs -- the only
    // form in which no value the compiler can see is still in flight), and the stores, issued by asm after it, have the next
    // iteration's rescoring to complete.  Every lane issues everything at valid addresses (a lane without a record reads the
    // tile's first column): no branch inside the statement.
    // (A first version let the rescoring run between the issue and the wait, each load an asm statement of its own with a
    // "=v" output: the compiler copied such a register -- v_mov of a loaded word, hoisted above the wait -- before the load
    // had written it, and 7-mers at a dense threshold got stale columns in one wave of four, now and then.)
    constexpr bool DEFER = EXACT_LB;   // records of a history word (see the kernel)
    struct Next {
        uint32_t rt, mcol;                  // row within the tile, column (sorted position)
        uint32_t w[S::LPADW], tw[S::TWN];   // the column's residue words
        uint32_t px, pm;                    // perm[row], perm[column], or the indices themselves
    } nx;
    // record index -> where it lies in the stage (two-ended: indices >= cnt_lo count down from the stage's top)
    auto stage_at = [&](uint32_t k) -> uint32_t { return (TWO && k >= cnt_lo) ? stage[(uint32_t)STAGE_CAP - 1u - (k - cnt_lo)] : stage[k]; };
    // records [f, e) of the wave's stage, one per lane, decoded into nx (see the append loop for the record's fields)
    auto decode = [&](uint32_t f, uint32_t e) {
        const bool live = f + lane < e;
        const uint32_t rec = live ? stage_at(f + lane) : 0u;
        const uint32_t q = (rec >> 16) & 31u, grp = rec >> 21;   // bit q of the lane's hit word at the step that looked
        const uint32_t b = DEFER ? q | 3u : q;                // where the bit was when its step noted it
        const uint32_t back = DEFER ? 3u - (q & 3u) : 0u;     // ... that many steps ago
        const uint32_t r = (b >> 3) + 4u - (b & 4u);          // bit 8r + 7: row r; bit 8r + 3: row 4 + r
        nx.rt = live ? grp * 8u + r : 0u;
        nx.mcol = A.col0 + (live ? (rec & 0xFFFFu) - back * 256u : 0u);
    };
    // the statement: loads for the records decoded last + the wait.  One asm per combination, each with exactly the operands
    // it uses (outputs are early-clobber: a superset of operands cost 16 registers and a spill).
#define HMK_FLUSH_ST_00(W_S, ...) asm volatile(W_S "s_waitcnt vmcnt(0)" : __VA_ARGS__ : [colp] "v"(colp) : "memory")
#define HMK_FLUSH_ST_01(W_S, ...) asm volatile(W_S "global_load_dword %[p0], %[pxp], off\n\tglobal_load_dword %[p1], %[pmp], off\n\t" "s_waitcnt vmcnt(0)" : __VA_ARGS__, [p0] "=&v"(p0), [p1] "=&v"(p1) : [colp] "v"(colp), [pxp] "v"(pxp), [pmp] "v"(pmp) : "memory")
#define HMK_FLUSH_ST_10(W_S, ...) asm volatile(W_S "global_load_dword %[t0], %[tp], off\n\tglobal_load_dword %[t1], %[tp], off offset:4\n\t" "s_waitcnt vmcnt(0)" : __VA_ARGS__, [t0] "=&v"(t0), [t1] "=&v"(t1) : [colp] "v"(colp), [tp] "v"(tp) : "memory")
#define HMK_FLUSH_ST_11(W_S, ...) asm volatile(W_S "global_load_dword %[t0], %[tp], off\n\tglobal_load_dword %[t1], %[tp], off offset:4\n\t" "global_load_dword %[p0], %[pxp], off\n\tglobal_load_dword %[p1], %[pmp], off\n\t" "s_waitcnt vmcnt(0)" : __VA_ARGS__, [t0] "=&v"(t0), [t1] "=&v"(t1), [p0] "=&v"(p0), [p1] "=&v"(p1) : [colp] "v"(colp), [tp] "v"(tp), [pxp] "v"(pxp), [pmp] "v"(pmp) : "memory")
#define HMK_FLUSH_ST(W_S, ...)                                                                                              \
    do {                                                                                                                  \
        if constexpr (EXACT_LB || S::TW == 0) { if (A.perm_identity) HMK_FLUSH_ST_00(W_S, __VA_ARGS__); else HMK_FLUSH_ST_01(W_S, __VA_ARGS__); } \
        else { if (A.perm_identity) HMK_FLUSH_ST_10(W_S, __VA_ARGS__); else HMK_FLUSH_ST_11(W_S, __VA_ARGS__); }             \
    } while (0)
    auto statement = [&]() {
        const uint8_t *colp = A.res_sorted + (size_t)nx.mcol * A.lpad;
        const uint8_t *tp = colp + (lbs - X);
        const uint32_t *pxp = A.perm + (A.row0 + nx.rt), *pmp = A.perm + nx.mcol;
        u32x2 w2 = {0, 0};
        u32x4 wa = {0, 0, 0, 0}, wb = {0, 0, 0, 0};
        uint32_t t0 = 0, t1 = 0, p0 = 0, p1 = 0;
        if constexpr (S::LPADW == 2) HMK_FLUSH_ST("global_load_dwordx2 %[w2], %[colp], off\n\t", [w2] "=&v"(w2));
        else if constexpr (S::LPADW == 4) HMK_FLUSH_ST("global_load_dwordx4 %[wa], %[colp], off\n\t", [wa] "=&v"(wa));
        else HMK_FLUSH_ST("global_load_dwordx4 %[wa], %[colp], off\n\tglobal_load_dwordx4 %[wb], %[colp], off offset:16\n\t", [wa] "=&v"(wa), [wb] "=&v"(wb));
        if constexpr (S::LPADW == 2) { nx.w[0] = w2.x; nx.w[1] = w2.y; }
        else {
            nx.w[0] = wa.x; nx.w[1] = wa.y; nx.w[2] = wa.z; nx.w[3] = wa.w;
            if constexpr (S::LPADW == 8) { nx.w[4] = wb.x; nx.w[5] = wb.y; nx.w[6] = wb.z; nx.w[7] = wb.w; }
        }
#pragma unroll
        for (int t = 0; t < S::TWN; t++) nx.tw[t] = 0;
        if constexpr (!EXACT_LB && S::TW > 0) { nx.tw[0] = t0; if constexpr (S::TW > 1) nx.tw[1] = t1; }
        nx.px = A.perm_identity ? A.row0 + nx.rt : p0;
        nx.pm = A.perm_identity ? nx.mcol : p1;
    };
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    // Two-ended stages: the iterations [0, cnt_lo) hold group 0's records, those from the next multiple of 64 on group 1's -- an
    // iteration never mixes groups.  k indexes RECORDS (what decode / stage_at take); `it` walks iterations.
    const uint32_t lo_iters = TWO ? (cnt_lo + 63u) / 64u : (cnt + 63u) / 64u;
    const uint32_t n_iters = TWO ? lo_iters + (cnt_hi + 63u) / 64u : lo_iters;
    // first record and the end of the records of iteration `it`
    auto iter_first = [&](uint32_t it) -> uint32_t { return (TWO && it >= lo_iters) ? cnt_lo + (it - lo_iters) * 64u : it * 64u; };
    auto iter_end = [&](uint32_t it) -> uint32_t { return (TWO && it < lo_iters) ? cnt_lo : cnt; };
    decode(iter_first(0), iter_end(0));
    statement();
    for (uint32_t it = 0; it < n_iters; it++) {   // wave-uniform trip count: the table reads below run for whole waves
        const uint32_t k = iter_first(it) + lane;
        const bool live = k < iter_end(it);
        uint32_t words[S::LPADW], tw[S::TWN];
#pragma unroll
        for (int t = 0; t < S::LPADW; t++) words[t] = nx.w[t];
#pragma unroll
        for (int t = 0; t < S::TWN; t++) tw[t] = nx.tw[t];
        const uint32_t rt = nx.rt, r = rt & 7u;
        uint32_t x = nx.px, m = nx.pm;
        if (A.symmetric && x > m) { const uint32_t t = x; x = m; m = t; }
        const unsigned long long pos = base + k;
        const bool ok = live && pos < A.cap_per_shard;   // stored edges only
        const unsigned long long slot = (unsigned long long)A.shard * A.cap_per_shard + pos;
        count_degrees(x, m, ok);
        uint32_t off[CAP], toff[S::NT];
        uint32_t mx = 0;   // best shift = largest plane sum
        if constexpr (TWO) {
            // ONE group per iteration (wave-uniform): whole 8-byte entries, conflict-free, both halves summed, the record's row
            // cut out of the pair with one v_perm_b32 per plane (selector byte 0 = r: 0-3 pick from the low dword, 4-7 from the
            // high one; the other selector bytes give zero)
            const uint32_t gbase = tab + (it >= lo_iters ? (uint32_t)S::GROUP_STEP : 0u);
            S::template offsets_of<true>(words, tw, gbase, 0u, off, toff);
            const uint32_t sel = 0x0c0c0c00u | r;
#pragma unroll
            for (int u = 0; u < S::ND; u++) {
                constexpr int LA = CAP + D;
                const int jlo = X - u > 0 ? X - u : 0, jhi = LA + X - u < CAP ? LA + X - u : CAP;
                uint32_t a0 = ((A.cinit[u >> 2] >> ((u & 3) * 8)) & 0xFFu) * 0x01010101u, a1 = a0;
                int j = jlo;
                if ((jhi - jlo) & 1) { const u32x2 e = rows_table_read<u32x2>(off[j] + (uint32_t)S::pos_addr(0, j + u - X)); a0 += e.x; a1 += e.y; j++; }
#pragma unroll
                for (; j + 1 < jhi; j += 2) {
                    const u32x2 e0 = rows_table_read<u32x2>(off[j] + (uint32_t)S::pos_addr(0, j + u - X));
                    const u32x2 e1 = rows_table_read<u32x2>(off[j + 1] + (uint32_t)S::pos_addr(0, j + 1 + u - X));
                    a0 = a0 + e0.x + e1.x; a1 = a1 + e0.y + e1.y;
                }
                // (an empty volatile asm that "modifies" the sums: the plane's adds must stand before the next plane's volatile
                // reads are issued.  A scheduling barrier alone does not do it -- the adds are sunk below all the reads before the
                // scheduler runs, and the reads' results spill.)
                asm volatile("" : "+v"(a0), "+v"(a1));
                mx = max(mx, __builtin_amdgcn_perm(a1, a0, sel));
            }
        } else {
            // The record's ROW is byte r of every 8-byte table entry: the lane reads the 4-byte half that holds it (ds_read_b32),
            // sums the halves as they are -- four byte lanes, proven not to carry -- and cuts its byte out of each plane's sum.
            const uint32_t grp = rt >> 3;
            S::template offsets_of<true>(words, tw, tab + grp * (uint32_t)S::GROUP_STEP + (r & 4u), 0u, off, toff);
            const uint32_t sh = (r & 3u) * 8u;
            static_assert(!EXACT_LB, "one-length rescoring shapes are two-ended");
            // One plane at a time, in a loop that is NOT unrolled (this path is cold and must stay small in registers): the
            // plane's cells straight from the position table, row position i = j + u - X wherever it lies inside the row -- the
            // literal form of ShiftedScorer.java:67-77 on the packed cells, independent of the main loop's unrolled schedule.
#pragma unroll 1
            for (int u = 0; u < S::ND; u++) {
                uint32_t cw = A.cinit[0];
#pragma unroll
                for (int q = 1; q < 8; q++) cw = (u >> 2) == q ? A.cinit[q] : cw;
                uint32_t a = ((cw >> ((u & 3) * 8)) & 0xFFu) * 0x01010101u;
#pragma unroll
                for (int j = 0; j < CAP; j++) {
                    const int i = j + u - X;
                    if (j < lbs && i >= 0 && i < la)   // wave-uniform
                        a += lds_read<uint32_t>(off[j] + (uint32_t)S::pos_addr(0, i));
                }
                mx = max(mx, (a >> sh) & 0xFFu);
            }
        }
        const int score = (int)mx - 128 + A.threshold;   // lane = 128 - threshold + score
        // the next iteration's records (past the last one: every lane reads the tile's first column -- harmless, and no branch)
        if (it + 1 < n_iters) decode(iter_first(it + 1), iter_end(it + 1)); else decode(cnt, cnt);
        statement();
        if (ok) {
            const unsigned long long e = ((unsigned long long)x << 40) | ((unsigned long long)m << 16) | (unsigned long long)((uint32_t)score & 0xFFFFu);
            const uint64_t *ep = A.edges + slot;
            asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(ep), "v"(e) : "memory");
        }
    }
#undef HMK_FLUSH_ST
#undef HMK_FLUSH_ST_00
#undef HMK_FLUSH_ST_01
#undef HMK_FLUSH_ST_10
#undef HMK_FLUSH_ST_11
    asm volatile("" ::: "memory");   // (the stage's reads are done: their values were used)
    drain_end();
    }
}

template <class F, int... Is>
__device__ __forceinline__ void rows_for_each_group(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// One tile (the workgroup's).  MODE: what a flush does beside storing the edge (hmk_device.h).  smem: the kernel's ONE static LDS object
// (rows_lds_bytes of it are used): its base address is a compile-time constant, so table offsets fold into the ds_read immediate.
// Inlined into its kernel (the flush reads the kernel's arguments through the kernarg pointer: P and tile_base must be the kernel's first two).
template <int X, int D, int CAP, bool EXACT_LB, int G, int MODE>
__device__ __forceinline__ void rows_tile(const NeighborParams &P, const uint32_t tile_base, uint8_t *smem) {
    using S = RowsShape<X, D, CAP, EXACT_LB, G>;
    constexpr int ND = S::ND, NI = S::NI, NEND = S::NEND, TAB_BYTES = S::TAB_BYTES;
    constexpr int R = 8 * G;
    constexpr int STAGE_CAP = rows_stage(X, D, CAP, EXACT_LB);  // records per wave; flushed when fewer than 64 slots are free
    constexpr int LDS_BYTES = TAB_BYTES + 576 + R * 32 + 4 * STAGE_CAP * 4;
    static_assert(LDS_BYTES == rows_lds_bytes(X, D, CAP, EXACT_LB, G), "rows_lds_bytes must match the LDS map");
    uint8_t *tab = smem;
    uint8_t *mb = smem + TAB_BYTES;
    uint8_t *rowres = mb + 576;
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(rowres + R * 32);

    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la;
    int lbs = EXACT_LB ? CAP : (int)Cp->lb;    // column length (wave-uniform)
    const bool case_b = Cp->case_b != 0;       // the column is the SHORTER sequence: cell = M[c][row[i]], else M[row[i]][c]
    const int tid = threadIdx.x;
    // 32-bit LDS pointer, wave-uniform (kept in a scalar register: nothing to spill around the flush call)
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + __builtin_amdgcn_readfirstlane(tid >> 6) * STAGE_CAP;

    build_begin();
    const uint32_t tab_addr = lds_addr(tab);
    for (int e = tid; e < 576; e += 256) mb[e] = P.mb[e];
    for (int e = tid; e < R * 32; e += 256) {
        const int r = e >> 5, k = e & 31;
        uint8_t v = 0;
        if ((uint32_t)r < T.nrows && (uint32_t)k < P.lpad) v = P.res_sorted[(size_t)(T.row0 + r) * P.lpad + k];
        rowres[e] = v;
    }
    __syncthreads();
    // ---- the cell tables: work item e = ((g * (NI + NEND) + k) * 24 + c) * 2 + h gathers rows 8g + 4h .. + 3 of row position
    // k (k < NI) or of the position k - NI places before the row's end (end table) ----
    for (int e = tid; e < G * (NI + NEND) * 24 * 2; e += 256) {
        const int h = e & 1, ic = e >> 1;
        const int gi = ic / 24, c = ic - gi * 24;
        const int g = gi / (NI + NEND), k = gi - g * (NI + NEND);
        const int i = k < NI ? k : la - 1 - (k - NI);
        const int dst = k < NI ? S::pos_addr(g, k) : S::end_addr(g, k - NI);
        uint32_t v = 0;
        if (i >= 0 && i < la) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = 8 * g + 4 * h + q;
                if ((uint32_t)r < T.nrows) {
                    const int a = rowres[r * 32 + i];
                    v |= (uint32_t)(case_b ? mb[c * 24 + a] : mb[a * 24 + c]) << (q * 8);
                }
            }
        }
        *reinterpret_cast<uint32_t *>(tab + dst + c * 8 + h * 4) = v;
    }
    __syncthreads();
    build_end();

    // initial lanes: every byte of plane u starts at g + penalty(s) - bias * cells(s) (TileClass::cinit, one byte per shift)
    uint32_t ci[ND];
#pragma unroll
    for (int u = 0; u < ND; u++) ci[u] = ((Cp->cinit[u >> 2] >> ((u & 3) * 8)) & 0xFFu) * 0x01010101u;

    constexpr bool TWO = rows_two_ended(X, D, CAP, EXACT_LB, G);
    uint32_t cnt = 0;     // staged records of this wave (wave-uniform); two-ended: group 0's, from the stage's bottom
    uint32_t cnt_hi = 0;  // two-ended: group 1's, from the stage's top
    const uint32_t col_end = T.col0 + T.ncols;
    const uint32_t n_batches = (T.ncols + 255) / 256;
    const bool interior = T.diag == 0 && T.ncols % 256 == 0;  // every lane's column is a real pair
    const bool prio = true;

    // (The batch loop needs two of the kernel's arguments.  As fields of P they keep the WHOLE argument block -- 16 SGPRs -- alive
    // through the loop, and around the flush call the allocator parks that block in VGPR lanes and reads all 16 back at every step:
    // 18 v_readlane of a step's 150 VALU instructions.  Copying the two out removes them and makes the pass SLOWER, 2.560 against
    // 2.525 ms: they sit where the step's global load is in flight.  DESIGN.md 5.7.)
    const uint8_t *const res_sorted = P.res_sorted;
    const uint32_t lpad_s = P.lpad;

    constexpr bool INLOOP = rows_inloop_shape(X, D, CAP, EXACT_LB);
    constexpr bool DEFER = EXACT_LB && !INLOOP;   // (mixed lengths: short column runs, two groups -- 1.3 % slower with it)
    // Hits are rare per pair (0.26 % at the default threshold of 12-mers) but not per step: a wave tests 512 pairs at a time.
    // DEFER: the test's result is only NOTED at every step -- the top bits of the eight rows' bytes, merged into one word per
    // lane and shifted into a 4-step history (hm uses every 4th bit: step j of a quad lands on the bits = j mod 4) -- and the
    // wave looks at the history once per quad: one ballot, one append loop whose number of turns is the largest number of hits
    // any LANE has in the quad instead of four of them.
    const uint64_t ka64 = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t ka_lo = (uint32_t)ka64, ka_hi = (uint32_t)(ka64 >> 32);
    uint32_t acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = 0;
    for (uint32_t bt = 0; bt < n_batches; bt++) {
        const uint32_t colrel = bt * 256 + tid;
        const uint32_t col = T.col0 + colrel;
        uint32_t off[CAP], toff[S::NT];
        S::offsets(res_sorted + (size_t)col * lpad_s, col < col_end, lbs, tab_addr, tab_addr, off, toff);
        const bool look = !DEFER || (bt & 3u) == 3u || bt + 1 == n_batches;   // wave-uniform

        auto one_group = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            if ((uint32_t)(8 * g) >= T.nrows) return;   // wave-uniform
            uint32_t W0[ND], W1[ND];
            read_phase_begin(prio);
            S::template accumulate<g>(off, toff, lbs, ci, W0, W1);
            read_phase_end(prio);

            // ---- threshold test: some (row, shift) lane has its top bit set <=> score >= threshold ----
            uint32_t o0 = W0[0], o1 = W1[0];
#pragma unroll
            for (int u = 1; u < ND; u++) { o0 |= W0[u]; o1 |= W1[u]; }
            if constexpr (!DEFER)
                if (__ballot(((o0 | o1) & 0x80808080u) != 0) == 0) return;
            // ---- rare path (a hit somewhere in the wave): every lane appends ITS hits, one per turn ----
            // hm: bit 8r + 7 set <=> row r of the group (r < 4) reached the threshold for this lane's column, bit 8r + 3 <=> row
            // 4 + r.  The rows beyond the tile's last start at the initial lane value, which may itself have the top bit set:
            // rm0 / rm1 (wave-uniform) leave them out.
            const uint32_t rows_here = T.nrows - 8 * g;
            const uint32_t rm0 = rows_here >= 4 ? 0x80808080u : 0x80808080u & ((1u << (8 * rows_here)) - 1u);
            const uint32_t rm1 = rows_here >= 8 ? 0x80808080u : rows_here <= 4 ? 0u : 0x80808080u & ((1u << (8 * (rows_here - 4))) - 1u);
            uint32_t hm = (o0 & rm0) | ((o1 & rm1) >> 4);
            if (!interior) {  // wave-uniform: only edge tiles filter
                if (col >= col_end) hm = 0;
                if (T.diag != 0) {
                    const int k = (int)col - (int)(T.row0 + 8 * g);   // the row this column IS, relative to the group
                    if (T.diag == 1) {   // triangle: keep rows r with column > row, i.e. r < k
                        const int k0 = k < 0 ? 0 : k > 4 ? 4 : k, k1 = k < 4 ? 0 : k > 8 ? 4 : k - 4;
                        hm &= (k0 >= 4 ? 0x80808080u : 0x80808080u & ((1u << (8 * k0)) - 1u)) |
                              (k1 >= 4 ? 0x08080808u : 0x08080808u & ((1u << (8 * k1)) - 1u));
                    } else if (k >= 0 && k < 8) {   // full square minus the diagonal
                        hm &= ~(k < 4 ? 0x80u << (8 * k) : 0x08u << (8 * (k - 4)));
                    }
                }
            }
            if constexpr (DEFER) {
                // history: this step's bits stay at 3 mod 4, the earlier steps' move down by one per step
                acc[g] = (acc[g] >> 1) | hm;
                if (!look) return;
                hm = acc[g];
                acc[g] = 0;
                if (__ballot(hm != 0) == 0) return;
            }
            // (the flush sits OUTSIDE the append loop: the rescoring kind needs most of the register file; inside the loop the
            // compiler kept the loop's state in scratch memory for every turn of it)
            for (;;) {
                bool full = false;
                for (;;) {
                    const bool any = hm != 0;
                    const uint64_t mask = __ballot(any);
                    if (mask == 0) break;
                    if (cnt + cnt_hi > (uint32_t)(STAGE_CAP - 64)) { full = true; break; }   // keep room for one wave of hits
                    if (any) {
                        const uint32_t q = (uint32_t)__builtin_ctz(hm);
                        if constexpr (INLOOP) {
                            // the hit's row r (bit 8r + 7: row r; bit 8r + 3: row 4 + r) is byte r of every plane's register pair:
                            // one v_perm_b32 per plane (selector byte 0 = r: 0-3 pick from the low dword, 4-7 from the high one;
                            // the other selector bytes give zero), the largest of them is the pair's lane = 128 - threshold + score
                            const uint32_t r = (q >> 3) + 4u - (q & 4u);
                            const uint32_t sel = 0x0c0c0c00u | r;
                            uint32_t mx = __builtin_amdgcn_perm(W1[0], W0[0], sel);
#pragma unroll
                            for (int u = 1; u < ND; u++) mx = max(mx, __builtin_amdgcn_perm(W1[u], W0[u], sel));
                            stage[cnt + mbcnt64(mask)] = colrel | ((uint32_t)(8 * g) + r) << 16 | (mx - 128u) << 24;
                        } else {
                            // the record: this step's column | the hit's bit << 16 | group << 21.  Which row and which of the quad's
                            // steps the bit stands for is worked out by the flush, where all 64 lanes have a record: here a turn
                            // runs for the few lanes that still have a hit, and every instruction of it costs the wave a VALU slot
                            const uint32_t rec = (colrel | (uint32_t)g << 21) | q << 16;
                            if constexpr (TWO && g == 1) stage[(uint32_t)(STAGE_CAP - 1) - (cnt_hi + mbcnt64(mask))] = rec;
                            else stage[cnt + mbcnt64(mask)] = rec;
                        }
                        hm &= hm - 1u;   // clear the lowest set bit
                    }
                    if constexpr (TWO && g == 1) cnt_hi += (uint32_t)__popcll(mask);
                    else cnt += (uint32_t)__popcll(mask);
                }
                if (!full) break;
                flush_stage_rows<X, D, CAP, EXACT_LB, G, MODE>(stage, cnt, cnt_hi, tab_addr, ka_lo, ka_hi);
                cnt = 0;
                cnt_hi = 0;
            }
        };
        rows_for_each_group(std::make_integer_sequence<int, G>{}, one_group);
    }
    flush_stage_rows<X, D, CAP, EXACT_LB, G, MODE>(stage, cnt, cnt_hi, tab_addr, ka_lo, ka_hi);
}

template <int X, int D, int CAP, bool EXACT_LB, int G, int MODE>
__global__ void __launch_bounds__(256, rows_waves(2 * X + D + 1, CAP, rows_lds_bytes(X, D, CAP, EXACT_LB, G), EXACT_LB))
k_neighbors_rows(const NeighborParams P, const uint32_t tile_base) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[rows_lds_bytes(X, D, CAP, EXACT_LB, G)];
    rows_tile<X, D, CAP, EXACT_LB, G, MODE>(P, tile_base, smem);
}

// Mixed lengths, max shift 2 or 3 (BASELINE config 4a: 3): the launch group of (row length - column length = D, column capacity CAPB) holds
// tiles of several column lengths lb.  The capacity form takes lb at run time -- scalar-tested position pairs, the overlap's tail through
// an end table, a second set of offsets: 2.2 VALU instructions per table read where the one-length form needs 1.85, and with three
// launches resident the pass ran at 0.84-0.86 of its LDS bytes whatever the streams did (tools/trace_config4a.py).  Here the tile's class
// picks the ONE-LENGTH form of its own lb (rows up to length ROWS_LENS_MAXLA; two row groups per tile: column runs are short), everything
// compile-time; a length outside that range takes the capacity form as before.  One kernel per (D, CAPB) as before: same groups, same tiles.
// BASELINE config 4a: 4.28-4.38 -> 4.12-4.17 ms (0.86-0.87 of its LDS bytes).  (The bodies are compiled for the capacity form's register
// estimate, 5-7 waves per SIMD; asked for one, two or three waves more they fit without a spill -- and the pass takes the same time.)
constexpr int ROWS_LENS_MAXLA = 20;
constexpr int rows_lens_first(int capb) { return capb <= 12 ? 6 : capb <= 16 ? 13 : 17; }
constexpr bool rows_lens_has(int x, int d, int capb, int lb) { return lb >= rows_lens_first(capb) && lb <= capb && lb >= 2 * x && lb + d <= ROWS_LENS_MAXLA; }
constexpr int rows_lens_lds(int x, int d, int capb) {
    int m = rows_lds_bytes(x, d, capb, false, 2);
    for (int lb = rows_lens_first(capb); lb <= capb; lb++)
        if (rows_lens_has(x, d, capb, lb) && rows_lds_bytes(x, d, lb, true, 2) > m) m = rows_lds_bytes(x, d, lb, true, 2);
    return m;
}
constexpr int rows_lens_waves(int x, int d, int capb) {
    const int lds = rows_lens_lds(x, d, capb);
    int w = rows_waves(2 * x + d + 1, capb, lds, false);
    for (int lb = rows_lens_first(capb); lb <= capb; lb++)
        if (rows_lens_has(x, d, capb, lb) && rows_waves(2 * x + d + 1, lb, lds, true) < w) w = rows_waves(2 * x + d + 1, lb, lds, true);
    return w;
}
template <int X, int D, int CAPB, int MODE, int LB>
__device__ __forceinline__ void rows_lens_pick(const NeighborParams &P, const uint32_t tile_base, uint8_t *smem, int lb) {
    if constexpr (LB > CAPB) {
        rows_tile<X, D, CAPB, false, 2, MODE>(P, tile_base, smem);   // (a length without a form of its own)
    } else {
        if constexpr (rows_lens_has(X, D, CAPB, LB)) {
            if (lb == LB) { rows_tile<X, D, LB, true, 2, MODE>(P, tile_base, smem); return; }   // wave-uniform
        }
        rows_lens_pick<X, D, CAPB, MODE, LB + 1>(P, tile_base, smem, lb);
    }
}
template <int X, int D, int CAPB, int MODE>
__global__ void __launch_bounds__(256, rows_lens_waves(X, D, CAPB))
k_neighbors_rows_lens(const NeighborParams P, const uint32_t tile_base) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[rows_lens_lds(X, D, CAPB)];
    const int lb = (int)P.classes[P.tiles[tile_base + blockIdx.x].cls].lb;   // (scalar loads)
    rows_lens_pick<X, D, CAPB, MODE, rows_lens_first(CAPB)>(P, tile_base, smem, lb);
}


}  // namespace hmk
#endif
