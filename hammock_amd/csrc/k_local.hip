// k_local.hip -- LocalAlignmentScorer (LocalAlignmentScorer.java:27-86, direction-matrix Smith-Waterman):
// register-resident striped DP, tagged-max and packed-int16 forms; dense blocks and thresholded ordered pairs.
#include "hmk_device.h"

namespace hmk {

// -----------------------------------------------------------------------------
// LocalAlignmentScorer DP core shared by k_local_block and k_neighbors_local
// -----------------------------------------------------------------------------
// One row sequence (wave-uniform, as a query profile in LDS at row_q_addr: Q[c][iq] = four int8 per
// dword for lines 4 iq .. 4 iq + 3) against this lane's column sequence (boff[j] = residue * 32).
//
// ENC = false: the plain form -- H, U = H + (Dir == UP ? ext : open), lcand = H + (Dir == LEFT ? ext : open),
//   direction flags from compares (LocalAlignmentScorer.java:43-81).
// ENC = true ("tagged max"): every candidate is carried as 4 * value + tag, tag 3 = DIAGONAL, 2 = UP,
//   1 = LEFT, so ONE max3 yields the cell value AND the reference's direction priority on ties
//   (DIAGONAL > UP > LEFT, :73-81); max(.., 0) gives NOWHERE (tag 0) for mx < 0 (:63-65).  The gap
//   penalty of the next cell is a byte-table lookup by tag (v_perm_b32), not compares:
//       U' = he + PU[tag],  PU[tag] = 4 * (tag == UP   ? ext : open) + 2 - tag
//       L' = he + PL[tag],  PL[tag] = 4 * (tag == LEFT ? ext : open) + 1 - tag
//   Profile bytes hold 4 * score.  Needs |M| <= 31 and -31 <= penalties <= 0.
template <int LBMAX, bool ENC>
__device__ __forceinline__ int sw_row(uint32_t row_q_addr, int strips, int ncols, const uint32_t (&boff)[LBMAX],
                                      int gap_open, int gap_extend) {
    if (ENC) {
        auto b = [](int v) { return (uint32_t)v & 0xFFu; };
        const uint32_t PU = b(4 * gap_open + 2) | (b(4 * gap_open + 1) << 8) | (b(4 * gap_extend) << 16) | (b(4 * gap_open - 1) << 24);
        const uint32_t PL = b(4 * gap_open + 1) | (b(4 * gap_extend) << 8) | (b(4 * gap_open - 1) << 16) | (b(4 * gap_open - 2) << 24);
        int H[LBMAX], U[LBMAX];
#pragma unroll
        for (int j = 0; j < LBMAX; j++) { H[j] = 3; U[j] = 4 * gap_open + 2; }   // line 0: H = 0, Dir = LEFT (:97-100)
        int gm = 0;
        for (int st = 0; st < strips; st++) {
            const uint32_t strip_addr = row_q_addr + (uint32_t)st * 4u;
            int hd[4], lc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { hd[k] = 3; lc[k] = 4 * gap_open + 1; }  // column 0: H = 0, Dir = UP (:93-96)
            uint32_t qnext = lds_read<uint32_t>(strip_addr + boff[0]);
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                if (j < ncols) {
                    const uint32_t q = qnext;   // profile dword of column j, fetched one column ahead
                    if (j + 1 < LBMAX) qnext = lds_read<uint32_t>(strip_addr + boff[j + 1]);
                    int up = U[j];
                    int habove = H[j];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int diag = hd[k] + (int)(int8_t)(q >> (8 * k));        // 4 * (H[l-1][c-1] + M) + 3   (:59)
                        const int he = max(max(diag, max(up, lc[k])), 0);            // :61-67 with the direction in the low bits
                        gm = max(gm, he);                                            // :68-72
                        const uint32_t tag = (uint32_t)he & 3u;
                        const int pu = (int)(int8_t)__builtin_amdgcn_perm(0u, PU, tag);
                        const int pl = (int)(int8_t)__builtin_amdgcn_perm(0u, PL, tag);
                        hd[k] = habove;
                        habove = he | 3;
                        up = he + pu;                                                // :43-48,:57 for the cell below
                        lc[k] = he + pl;                                             // :50-55,:58 for the cell to the right
                    }
                    H[j] = habove;
                    U[j] = up;
                }
            }
        }
        return gm >> 2;
    } else {
        int H[LBMAX], U[LBMAX];
#pragma unroll
        for (int j = 0; j < LBMAX; j++) { H[j] = 0; U[j] = gap_open; }   // line 0: H = 0, Dir = LEFT (:97-100)
        int gmax = 0;
        for (int st = 0; st < strips; st++) {
            const uint32_t strip_addr = row_q_addr + (uint32_t)st * 4u;
            int hd[4], lc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { hd[k] = 0; lc[k] = gap_open; }  // column 0: H = 0, Dir = UP (:93-96)
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                if (j < ncols) {
                    const uint32_t q = lds_read<uint32_t>(strip_addr + boff[j]);
                    int up = U[j];
                    int habove = H[j];          // H[line-1][j]: the next column's diagonal for line k = 0
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int sc = (int)(int8_t)(q >> (8 * k));
                        const int diag = hd[k] + sc;                      // :59
                        const int left = lc[k];
                        const int mx = max(diag, max(up, left));         // :61
                        const bool neg = mx < 0;                          // :63
                        const bool d_eq = mx == diag, u_eq = mx == up;
                        const int h = max(mx, 0);                         // :64,:67
                        const bool is_up = u_eq && !d_eq && !neg;         // :76-81
                        const bool is_left = !(u_eq || d_eq || neg);      // :73-81
                        gmax = max(gmax, h);                              // :68-72
                        hd[k] = habove;
                        habove = h;
                        up = h + (is_up ? gap_extend : gap_open);
                        lc[k] = h + (is_left ? gap_extend : gap_open);
                    }
                    H[j] = habove;
                    U[j] = up;
                }
            }
        }
        return gmax;
    }
}

// query profiles of R rows into LDS: entry (r, c, iq) = four int8 (score * scale) for lines 4 iq .. 4 iq + 3;
// -128 for the pad residue (c == 24) and for lines beyond the row's length
template <int R>
__device__ __forceinline__ void build_profiles(uint32_t *q, const int8_t *m8, const uint8_t *rowres, const int *row_len,
                                               uint32_t nrows, int scale, int tid) {
    for (int e = tid; e < R * 25 * 8; e += 256) {
        const int r = e / 200, rem = e - r * 200, c = rem >> 3, iq = rem & 7;
        const int l1 = (uint32_t)r < nrows ? row_len[r] : 0;
        uint32_t dw = 0;
        for (int k = 0; k < 4; k++) {
            const int i = iq * 4 + k;
            int v = -128;
            if (i < l1 && c < 24) v = m8[rowres[r * 32 + i] * 24 + c] * scale;
            dw |= ((uint32_t)v & 0xFFu) << (8 * k);
        }
        q[e] = dw;
    }
}

// -----------------------------------------------------------------------------
// k_local_block: LocalAlignmentScorer for a dense block, register-resident DP
// -----------------------------------------------------------------------------
// score(seq1 = row, seq2 = column) for rows [r0, r1) x columns [c0, c1)
// (LocalAlignmentScorer.java:27-86).  One COLUMN sequence per lane (its DP state lives in
// VGPRs), the ROW sequence is wave-uniform.  Per row the workgroup keeps a "query profile"
// in LDS: Q[c][iq] = the substitution scores M[row[4 iq + k]][c], k = 0..3, as four int8 in one
// dword, so one ds_read_b32 feeds four cells.  The DP runs in strips of four lines, column by
// column; per column only H and the "up candidate" of the strip's last line survive:
//     H[j]   score of cell (line, j)
//     U[j]   H[j] + (Dir[j] == UP   ? gapExtend : gapOpen)   -- what the cell below adds (:43-48,:57)
//     lcand  H    + (Dir    == LEFT ? gapExtend : gapOpen)   -- what the cell to the right adds (:50-58)
// Dir follows the reference's assignment order (:73-81): DIAGONAL if mx == diag, else UP if
// mx == up, else LEFT; NOWHERE (neither flag) when mx < 0.
// Padding (lines >= len1, columns >= the lane's len2) uses score -128: with gap penalties <= 0 a
// padded cell can never exceed the largest real cell, so the running maximum is unaffected.
// Preconditions checked by the host: |M| <= 127, gapOpen <= 0, gapExtend <= 0 (else k_pairs<1>).
template <int LBMAX, bool ENC>
__global__ void __launch_bounds__(256)
k_local_block(const uint8_t *__restrict__ res32, const uint8_t *__restrict__ len, const int32_t *__restrict__ Mg,
              uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int gap_open, int gap_extend,
              int32_t *__restrict__ out) {
    constexpr int R = 16;                 // rows per workgroup pass
    constexpr int QROW = 25 * 8 * 4;      // profile bytes per row: 25 residues (24 + pad) x 8 dwords
    __shared__ __attribute__((aligned(16))) uint8_t smem[R * QROW + 576 + R * 32];
    int8_t *m8 = reinterpret_cast<int8_t *>(smem + R * QROW);
    uint8_t *rowres = smem + R * QROW + 576;
    const int tid = threadIdx.x;
    const uint32_t width = c1 - c0;
    const uint32_t col = c0 + blockIdx.x * 256 + tid;
    const bool col_ok = col < c1;
    const uint32_t row_base = r0 + blockIdx.y * R;
    const uint32_t nrows = min((uint32_t)R, r1 - row_base);

    for (int e = tid; e < 576; e += 256) m8[e] = (int8_t)Mg[e];
    for (int e = tid; e < R * 32; e += 256) {
        const uint32_t r = (uint32_t)e >> 5;
        rowres[e] = r < nrows ? res32[(size_t)(row_base + r) * 32 + (e & 31)] : 0;
    }
    __syncthreads();
    __shared__ int row_len[R];
    if (tid < R) row_len[tid] = (uint32_t)tid < nrows ? len[row_base + tid] : 0;
    __syncthreads();
    build_profiles<R>(reinterpret_cast<uint32_t *>(smem), m8, rowres, row_len, nrows, ENC ? 4 : 1, tid);
    __syncthreads();

    // this lane's column sequence -> profile byte offsets (pad residue 24 beyond its length)
    uint32_t boff[LBMAX];
    int len2 = 0;
    {
        uint32_t words[8];
#pragma unroll
        for (int q = 0; q < 8; q++) words[q] = 0;
        if (col_ok) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(res32 + (size_t)col * 32);
            const u32x4 v0 = src[0], v1 = src[1];
            words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
            words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
            len2 = len[col];
        }
#pragma unroll
        for (int j = 0; j < LBMAX; j++) {
            const uint32_t c = j < len2 ? ((words[j >> 2] >> ((j & 3) * 8)) & 0xFFu) : 24u;
            boff[j] = c * 32u;
        }
    }
    // widest column of the wave: columns beyond it are skipped wave-uniformly
    int wmax = len2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    wmax = __builtin_amdgcn_readfirstlane(wmax);

    const uint32_t q_addr = lds_addr(smem);
    for (uint32_t r = 0; r < nrows; r++) {
        const int len1 = row_len[r];
        const int gmax = sw_row<LBMAX, ENC>(q_addr + r * QROW, (len1 + 3) >> 2, wmax, boff, gap_open, gap_extend);
        if (col_ok) out[(size_t)(row_base + r - r0) * width + (col - c0)] = gmax;
    }
}

// -----------------------------------------------------------------------------
// k_neighbors_local: all ORDERED pairs with LocalAlignmentScorer, thresholded -> edge list
// -----------------------------------------------------------------------------
// Same striped register DP as k_local_block, on the length-bucketed tiles of the neighbour plan:
// one (row length, column length) class per tile, so no padding columns and a wave-uniform column
// bound.  Rows are seq1 (lines), columns seq2; the edge (x = column, m = row) carries
// sequenceScore(seq1 = m, seq2 = x) like every other edge (row_is_m = 1 swaps them at flush).
template <int LBMAX, bool ENC>
__global__ void __launch_bounds__(256)
k_neighbors_local(const NeighborParams P, const uint32_t tile_base, const int32_t *__restrict__ Mg, int gap_open,
                  int gap_extend, int threshold) {
    constexpr int R = 16;
    constexpr int QROW = 25 * 8 * 4;
    constexpr int STAGE_CAP = 128;
    constexpr int REC_DW = 3;
    __shared__ __attribute__((aligned(16))) uint8_t smem[R * QROW + 576 + R * 32 + 4 * STAGE_CAP * REC_DW * 4];
    int8_t *m8 = reinterpret_cast<int8_t *>(smem + R * QROW);
    uint8_t *rowres = smem + R * QROW + 576;
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(smem + R * QROW + 576 + R * 32);
    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la, lb = Cp->lb;   // la: row (seq1) length, lb: column (seq2) length
    const uint32_t shard = tile_shard(tile_base + blockIdx.x);
    const int tid = threadIdx.x;
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);   // 32-bit LDS pointer: no 64-bit flat pointer held (and spilled) across the tile

    for (int e = tid; e < 576; e += 256) m8[e] = (int8_t)Mg[e];
    for (int e = tid; e < R * 32; e += 256) {
        const uint32_t r = (uint32_t)e >> 5, k = e & 31;
        rowres[e] = (r < T.nrows && k < P.lpad) ? P.res_sorted[(size_t)(T.row0 + r) * P.lpad + k] : 0;
    }
    __syncthreads();
    __shared__ int row_len[R];
    if (tid < R) row_len[tid] = (uint32_t)tid < T.nrows ? la : 0;
    __syncthreads();
    build_profiles<R>(reinterpret_cast<uint32_t *>(smem), m8, rowres, row_len, T.nrows, ENC ? 4 : 1, tid);
    __syncthreads();

    const uint32_t q_addr = lds_addr(smem);
    const uint32_t col_end = T.col0 + T.ncols;
    const int strips = (la + 3) >> 2;
    uint32_t cnt = 0;
    for (uint32_t c0 = T.col0; c0 < col_end; c0 += 256) {
        const uint32_t col = c0 + tid;
        const bool col_ok = col < col_end;
        uint32_t boff[LBMAX];
        {
            uint32_t words[8];
#pragma unroll
            for (int q = 0; q < 8; q++) words[q] = 0;
            if (col_ok) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(P.res_sorted + (size_t)col * P.lpad);
                const u32x4 v0 = src[0];
                words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
                if (LBMAX > 16) {
                    const u32x4 v1 = src[1];
                    words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
                }
            }
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                const uint32_t c = (col_ok && j < lb) ? ((words[j >> 2] >> ((j & 3) * 8)) & 0xFFu) : 24u;
                boff[j] = c * 32u;
            }
        }
        for (uint32_t r = 0; r < T.nrows; r++) {
            const int gmax = sw_row<LBMAX, ENC>(q_addr + r * QROW, strips, lb, boff, gap_open, gap_extend);
            bool keep = col_ok && gmax >= threshold;
            if (T.diag) keep = keep && col != T.row0 + r;   // a sequence is never paired with itself
            const uint64_t mask = __ballot(keep);
            if (mask != 0) {
                if (cnt > (uint32_t)(STAGE_CAP - 64)) {
                    flush_stage<0>(stage, cnt, P, T, 0, false, shard);
                    cnt = 0;
                }
                if (keep) {
                    HMK_LDS uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
                    rec[0] = col;
                    rec[1] = r;
                    rec[2] = (uint32_t)gmax;
                }
                cnt += (uint32_t)__popcll(mask);
            }
        }
    }
    flush_stage<0>(stage, cnt, P, T, 0, false, shard);
}

// -----------------------------------------------------------------------------
// k_neighbors_local_pk: the tagged-max DP on TWO column sequences per lane (packed int16 halves)
// -----------------------------------------------------------------------------
// Every value of the tagged-max DP fits 16 bits (4 * 32 * 31 + 3 < 2^15), so one VGPR carries the
// cells of two column sequences and v_pk_add_i16 / v_pk_max_i16 advance both: 13 VALU instructions
// per PAIR of cells instead of 10-11 per cell (the kernel sits on the integer VALU-issue roofline,
// DESIGN.md 5.3b).  Profiles hold 4 * score as int16, four lines per ds_read_b64; one v_perm_b32 per
// line zips the two sequences' scores.  The gap-penalty table lookup returns both halves at once:
// selector byte 2h = tag_h (low-byte table in S1), byte 2h + 1 = tag_h + 4 (high-byte table in S0).
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 pk(uint32_t u) { return __builtin_bit_cast(s16x2, u); }
__device__ __forceinline__ uint32_t un(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int N> struct LocalIndex { static constexpr int value = N; };
__device__ __forceinline__ u16x2 pku(uint32_t u) { return __builtin_bit_cast(u16x2, u); }
__device__ __forceinline__ uint32_t unu(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// SAT = true (gap_open <= -1: every table entry below is a magnitude >= 0 and fits one byte): the up / left candidates are
// he - magnitude with UNSIGNED SATURATION (v_pk_sub_u16 clamp).  A candidate below zero becomes 0 = "value 0, NOWHERE", which is
// what the clamp of :63-67 makes of it when it wins -- so max(up, left) >= 0 always, the cell is max(diagonal, that) with no third
// maximum against zero, the byte selector is ONE v_and_or_b32 (the table's high bytes are the constant 0, selector 0x0c) instead of
// v_and + v_mad, and the columns leave the unrolled loop with a break, so no value has to be copied to keep two paths in the same
// registers: 11 VALU instructions per pair of cells instead of 14.25.  SAT = false is the form of rounds 1-3
// (signed candidates, two-table lookup), kept for gap_open == 0.
template <int LBMAX, bool SAT>
__device__ __forceinline__ uint32_t sw_row_pk(uint32_t row_q_addr, int strips, int nlines, int ncols, const uint32_t (&boff_lo)[LBMAX],
                                              const uint32_t (&boff_hi)[LBMAX], int gap_open, int gap_extend) {
    // 16-bit table entries by tag (0 NOWHERE, 1 LEFT, 2 UP, 3 DIAGONAL), split into low / high bytes
    const int pu[4] = {4 * gap_open + 2, 4 * gap_open + 1, 4 * gap_extend, 4 * gap_open - 1};
    const int pl[4] = {4 * gap_open + 1, 4 * gap_extend, 4 * gap_open - 1, 4 * gap_open - 2};
    uint32_t PUlo = 0, PUhi = 0, PLlo = 0, PLhi = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t u = (uint32_t)(SAT ? -pu[t] : pu[t]), l = (uint32_t)(SAT ? -pl[t] : pl[t]);
        PUlo |= (u & 0xFFu) << (8 * t); PUhi |= ((u >> 8) & 0xFFu) << (8 * t);
        PLlo |= (l & 0xFFu) << (8 * t); PLhi |= ((l >> 8) & 0xFFu) << (8 * t);
    }
    // v_perm_b32 reads at most one SGPR: pin one table of each pair in a VGPR once instead of a
    // v_mov per use
    if (!SAT) {
        asm volatile("" : "+v"(PUlo));
        asm volatile("" : "+v"(PLlo));
    }
    const uint32_t both = 0x00010001u;
    const s16x2 zero = pk(0u);
    const uint32_t u_first = SAT ? 0u : (uint32_t)((4 * gap_open + 2) & 0xFFFF) * both;   // the candidates of line 0 / column 0: below zero
    const uint32_t l_first = SAT ? 0u : (uint32_t)((4 * gap_open + 1) & 0xFFFF) * both;
    uint32_t H[LBMAX], U[LBMAX];
#pragma unroll
    for (int j = 0; j < LBMAX; j++) { H[j] = 3u * both; U[j] = u_first; }
    s16x2 gm = zero;
    // (v_and_or_b32 is a VOP3: no literals, one scalar operand -- the mask in an SGPR, the constant in a VGPR)
    uint32_t tag_mask = 0x00030003u, sel_high = 0x0c000c00u;
    if (SAT) {
        asm volatile("" : "+s"(tag_mask));
        asm volatile("" : "+v"(sel_high));
    }
    // one strip of KL lines (4, or what is left of the row sequence in its last strip: the lines beyond it are padding that can
    // never hold the maximum -- computing them anyway was 12 % of the cells at lengths 7..20)
    auto strip = [&](auto KLc, int st) __attribute__((always_inline)) -> void {
        constexpr int KL = decltype(KLc)::value;
        uint32_t strip_addr = row_q_addr + (uint32_t)st * 8u;
        if (SAT) asm volatile("" : "+s"(strip_addr));   // (or the 2 x LBMAX profile addresses of the last strip are computed above the choice of its form and held in VGPRs)
        uint32_t hd[4], lc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { hd[k] = 3u * both; lc[k] = l_first; }
        // one column; SAT: the next one is reached from inside it, so "no more columns" LEAVES the chain (an exit per column,
        // no join between columns: hd / lc / habove rotate by renaming, not by v_mov)
        auto column = [&](auto self, auto J) __attribute__((always_inline)) -> void {
            constexpr int j = decltype(J)::value;
            if constexpr (j < LBMAX) {
                if (SAT) { if (j >= ncols) return; }
                if (SAT || j < ncols) {
                    const u32x2 qa = lds_read<u32x2>(strip_addr + boff_lo[j]);   // lines 0,1 | 2,3 of sequence "lo"
                    const u32x2 qb = lds_read<u32x2>(strip_addr + boff_hi[j]);   // ... of sequence "hi"
                    uint32_t sc[4];
                    sc[0] = __builtin_amdgcn_perm(qb.x, qa.x, 0x05040100u);
                    sc[1] = __builtin_amdgcn_perm(qb.x, qa.x, 0x07060302u);
                    sc[2] = __builtin_amdgcn_perm(qb.y, qa.y, 0x05040100u);
                    sc[3] = __builtin_amdgcn_perm(qb.y, qa.y, 0x07060302u);
                    uint32_t up = U[j];
                    uint32_t habove = H[j];
#pragma unroll
                    for (int k = 0; k < KL; k++) {
                        const s16x2 diag = pk(hd[k]) + pk(sc[k]);                                        // :59
                        const s16x2 m1 = __builtin_elementwise_max(pk(up), pk(lc[k]));
                        s16x2 he = __builtin_elementwise_max(diag, m1);                                  // :61
                        if (!SAT) he = __builtin_elementwise_max(he, zero);                              // :63-67 (SAT: m1 >= 0 already)
                        gm = __builtin_elementwise_max(gm, he);                                          // :68-72
                        const uint32_t heu = un(he);
                        hd[k] = habove;
                        habove = heu | 0x00030003u;
                        if (SAT) {
                            const uint32_t sel = (heu & tag_mask) | sel_high;                            // byte 2h = tag_h, byte 2h + 1 = the constant 0
                            up = unu(__builtin_elementwise_sub_sat(pku(heu), pku(__builtin_amdgcn_perm(0u, PUlo, sel))));      // :43-48,:57
                            lc[k] = unu(__builtin_elementwise_sub_sat(pku(heu), pku(__builtin_amdgcn_perm(0u, PLlo, sel))));   // :50-55,:58
                        } else {
                            const uint32_t sel = (heu & 0x00030003u) * 0x0101u + 0x04000400u;
                            up = un(he + pk(__builtin_amdgcn_perm(PUhi, PUlo, sel)));
                            lc[k] = un(he + pk(__builtin_amdgcn_perm(PLhi, PLlo, sel)));
                        }
                    }
                    H[j] = habove;
                    U[j] = up;
                }
                self(self, LocalIndex<j + 1>{});
            }
        };
        column(column, LocalIndex<0>{});
    };
    if (SAT) {   // the last strip with the lines it has
        const int last = strips - 1, kl = nlines - 4 * last;
        for (int st = 0; st < last; st++) strip(LocalIndex<4>{}, st);
        if (kl >= 4) strip(LocalIndex<4>{}, last);
        else if (kl == 3) strip(LocalIndex<3>{}, last);
        else if (kl == 2) strip(LocalIndex<2>{}, last);
        else strip(LocalIndex<1>{}, last);
    } else {
        for (int st = 0; st < strips; st++) strip(LocalIndex<4>{}, st);
    }
    const uint32_t g = un(gm);
    return ((g & 0xFFFFu) >> 2) | (((g >> 16) >> 2) << 16);   // both maxima are >= 0
}

template <int LBMAX, bool SAT>
__global__ void __launch_bounds__(256)
k_neighbors_local_pk(const NeighborParams P, const uint32_t tile_base, const int32_t *__restrict__ Mg, int gap_open,
                     int gap_extend, int threshold) {
    constexpr int R = 16;
    constexpr int QROW = 25 * 8 * 8;      // 25 residues x 8 strips x (4 lines x int16)
    constexpr int STAGE_CAP = 192;        // room for two flush-free appends (lo and hi halves)
    constexpr int REC_DW = 3;
    __shared__ __attribute__((aligned(16))) uint8_t smem[R * QROW + 576 + R * 32 + 4 * STAGE_CAP * REC_DW * 4];
    int8_t *m8 = reinterpret_cast<int8_t *>(smem + R * QROW);
    uint8_t *rowres = smem + R * QROW + 576;
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(smem + R * QROW + 576 + R * 32);
    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la, lb = Cp->lb;
    const uint32_t shard = tile_shard(tile_base + blockIdx.x);
    const int tid = threadIdx.x;
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);   // 32-bit LDS pointer: no 64-bit flat pointer held (and spilled) across the tile

    for (int e = tid; e < 576; e += 256) m8[e] = (int8_t)Mg[e];
    for (int e = tid; e < R * 32; e += 256) {
        const uint32_t r = (uint32_t)e >> 5, k = e & 31;
        rowres[e] = (r < T.nrows && k < P.lpad) ? P.res_sorted[(size_t)(T.row0 + r) * P.lpad + k] : 0;
    }
    __syncthreads();
    // profiles: entry (r, c, strip) = four int16 (4 * score; -128 for padding lines / the pad residue)
    for (int e = tid; e < R * 25 * 8; e += 256) {
        const int r = e / 200, rem = e - r * 200, c = rem >> 3, iq = rem & 7;
        uint32_t w[2] = {0, 0};
        for (int k = 0; k < 4; k++) {
            const int i = iq * 4 + k;
            int v = -128;
            if ((uint32_t)r < T.nrows && i < la && c < 24) v = m8[rowres[r * 32 + i] * 24 + c] * 4;
            w[k >> 1] |= ((uint32_t)v & 0xFFFFu) << (16 * (k & 1));
        }
        reinterpret_cast<u32x2 *>(smem)[e] = u32x2{w[0], w[1]};
    }
    __syncthreads();

    const uint32_t q_addr = lds_addr(smem);
    const uint32_t col_end = T.col0 + T.ncols;
    const int strips = (la + 3) >> 2;
    uint32_t cnt = 0;
    for (uint32_t c0 = T.col0; c0 < col_end; c0 += 512) {
        uint32_t colv[2];
        bool okv[2];
        uint32_t boff[2][LBMAX];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            colv[h] = c0 + h * 256 + tid;
            okv[h] = colv[h] < col_end;
            uint32_t words[8];
#pragma unroll
            for (int q = 0; q < 8; q++) words[q] = 0;
            if (okv[h]) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(P.res_sorted + (size_t)colv[h] * P.lpad);
                const u32x4 v0 = src[0];
                words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
                if (LBMAX > 16) {
                    const u32x4 v1 = src[1];
                    words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
                }
            }
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                const uint32_t c = (okv[h] && j < lb) ? ((words[j >> 2] >> ((j & 3) * 8)) & 0xFFu) : 24u;
                boff[h][j] = c * 64u;   // 8 strips x 8 bytes per residue
            }
        }
        for (uint32_t r = 0; r < T.nrows; r++) {
            const uint32_t g2 = sw_row_pk<LBMAX, SAT>(q_addr + r * QROW, strips, la, lb, boff[0], boff[1], gap_open, gap_extend);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int gmax = (int)((g2 >> (16 * h)) & 0xFFFFu);
                bool keep = okv[h] && gmax >= threshold;
                if (T.diag) keep = keep && colv[h] != T.row0 + r;
                const uint64_t mask = __ballot(keep);
                if (mask != 0) {
                    if (cnt > (uint32_t)(STAGE_CAP - 64)) {
                        flush_stage<0>(stage, cnt, P, T, 0, false, shard);
                        cnt = 0;
                    }
                    if (keep) {
                        HMK_LDS uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
                        rec[0] = colv[h];
                        rec[1] = r;
                        rec[2] = (uint32_t)gmax;
                    }
                    cnt += (uint32_t)__popcll(mask);
                }
            }
        }
    }
    flush_stage<0>(stage, cnt, P, T, 0, false, shard);
}

// -----------------------------------------------------------------------------
// k_local_block_pk: the dense block on the packed DP (two column sequences per lane, sw_row_pk)
// -----------------------------------------------------------------------------
// Same contract as k_local_block; a workgroup takes 16 rows x 512 columns (lane t: columns t and 256 + t of its slice).  Columns
// are not length-sorted here, so a lane's sequences are padded with the pad residue up to the wave's longest one, and rows carry
// their own length (strips and the last strip's lines per row).  Taken when the tagged range holds (|M| <= 31, -31 <= penalties
// <= 0); SAT as in the neighbour pass (gap_open <= -1).
template <int LBMAX, bool SAT>
__global__ void __launch_bounds__(256)
k_local_block_pk(const uint8_t *__restrict__ res32, const uint8_t *__restrict__ len, const int32_t *__restrict__ Mg,
                 uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int gap_open, int gap_extend, int32_t *__restrict__ out) {
    constexpr int R = 16;
    constexpr int QROW = 25 * 8 * 8;      // 25 residues x 8 strips x (4 lines x int16)
    __shared__ __attribute__((aligned(16))) uint8_t smem[R * QROW + 576 + R * 32];
    __shared__ int row_len[R];
    int8_t *m8 = reinterpret_cast<int8_t *>(smem + R * QROW);
    uint8_t *rowres = smem + R * QROW + 576;
    const int tid = threadIdx.x;
    const uint32_t width = c1 - c0;
    const uint32_t row_base = r0 + blockIdx.y * R;
    const uint32_t nrows = min((uint32_t)R, r1 - row_base);

    for (int e = tid; e < 576; e += 256) m8[e] = (int8_t)Mg[e];
    for (int e = tid; e < R * 32; e += 256) {
        const uint32_t r = (uint32_t)e >> 5;
        rowres[e] = r < nrows ? res32[(size_t)(row_base + r) * 32 + (e & 31)] : 0;
    }
    if (tid < R) row_len[tid] = (uint32_t)tid < nrows ? len[row_base + tid] : 0;
    __syncthreads();
    for (int e = tid; e < R * 25 * 8; e += 256) {
        const int r = e / 200, rem = e - r * 200, c = rem >> 3, iq = rem & 7;
        uint32_t w[2] = {0, 0};
        for (int k = 0; k < 4; k++) {
            const int i = iq * 4 + k;
            int v = -128;
            if (i < row_len[r] && c < 24) v = m8[rowres[r * 32 + i] * 24 + c] * 4;
            w[k >> 1] |= ((uint32_t)v & 0xFFFFu) << (16 * (k & 1));
        }
        reinterpret_cast<u32x2 *>(smem)[e] = u32x2{w[0], w[1]};
    }
    __syncthreads();

    uint32_t colv[2];
    bool okv[2];
    uint32_t boff[2][LBMAX];
    int wmax = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        colv[h] = c0 + blockIdx.x * 512 + h * 256 + tid;
        okv[h] = colv[h] < c1;
        uint32_t words[8];
#pragma unroll
        for (int q = 0; q < 8; q++) words[q] = 0;
        int len2 = 0;
        if (okv[h]) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(res32 + (size_t)colv[h] * 32);
            const u32x4 v0 = src[0], v1 = src[1];
            words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
            words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
            len2 = len[colv[h]];
        }
#pragma unroll
        for (int j = 0; j < LBMAX; j++) {
            const uint32_t c = j < len2 ? ((words[j >> 2] >> ((j & 3) * 8)) & 0xFFu) : 24u;
            boff[h][j] = c * 64u;
        }
        wmax = max(wmax, len2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    wmax = __builtin_amdgcn_readfirstlane(wmax);

    const uint32_t q_addr = lds_addr(smem);
    for (uint32_t r = 0; r < nrows; r++) {
        const int len1 = row_len[r];
        uint32_t g2 = 0;
        if (len1 > 0 && wmax > 0) g2 = sw_row_pk<LBMAX, SAT>(q_addr + r * QROW, (len1 + 3) >> 2, len1, wmax, boff[0], boff[1], gap_open, gap_extend);
#pragma unroll
        for (int h = 0; h < 2; h++)
            if (okv[h]) out[(size_t)(row_base + r - r0) * width + (colv[h] - c0)] = (int32_t)((g2 >> (16 * h)) & 0xFFFFu);
    }
}

// -----------------------------------------------------------------------------
// k_neighbors_local_literal: the same pass with the literal DP (LocalAlignmentScorer.java:31-86 line by line)
// -----------------------------------------------------------------------------
// For what the striped kernels do not take: positive gap penalties (the reference imposes no sign, :43-55) and matrix
// entries beyond int8.  Same tiles, same edge orientation (row = seq1 = m, column = seq2 = x), one column per lane.
__global__ void __launch_bounds__(256)
k_neighbors_local_literal(const NeighborParams P, const uint32_t tile_base, const int32_t *__restrict__ Mg, int gap_open,
                          int gap_extend, int threshold) {
    constexpr int R = 16, STAGE_CAP = 128, REC_DW = 3;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *M = reinterpret_cast<int *>(smem);                                   // 576 dwords
    uint32_t *colseq = reinterpret_cast<uint32_t *>(smem + 2304);             // 256 x 9 dwords
    uint32_t *rowseq = colseq + 256 * SEQ_STRIDE_DW;                          // R x 8 dwords
    uint32_t *dp = rowseq + R * 8;                                            // 33 x 256 dwords: one DP line per lane
    uint32_t *stage_all = dp + 33 * 256;
    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la, lb = Cp->lb;
    const uint32_t shard = tile_shard(tile_base + blockIdx.x);
    const int tid = threadIdx.x;
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);
    for (int e = tid; e < 576; e += 256) M[e] = Mg[e];
    for (int e = tid; e < R * 8; e += 256) {
        const int r = e >> 3, q = e & 7;
        uint32_t v = 0;
        if ((uint32_t)r < T.nrows && (uint32_t)(q * 4) < P.lpad)
            v = reinterpret_cast<const uint32_t *>(P.res_sorted + (size_t)(T.row0 + r) * P.lpad)[q];
        rowseq[e] = v;
    }
    __syncthreads();
    uint32_t cnt = 0;
    const uint32_t col_end = T.col0 + T.ncols;
    uint32_t *mine = colseq + tid * SEQ_STRIDE_DW;
    for (uint32_t c0 = T.col0; c0 < col_end; c0 += 256) {
        const uint32_t col = c0 + tid;
        if (col < col_end) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(P.res_sorted + (size_t)col * P.lpad);
            for (uint32_t q = 0; q < P.lpad / 4; q++) mine[q] = src[q];
        }
        for (uint32_t r = 0; r < T.nrows; r++) {
            if (cnt > (uint32_t)(STAGE_CAP - 64)) {
                flush_stage<0>(stage, cnt, P, T, 0, false, shard);
                cnt = 0;
            }
            bool keep = col < col_end;
            if (T.diag) keep = keep && col != T.row0 + r;
            int score = 0;
            if (keep) {
                score = local_score_literal(M, reinterpret_cast<const uint8_t *>(rowseq + r * 8), la,
                                            reinterpret_cast<const uint8_t *>(mine), lb, gap_open, gap_extend, dp + tid, 256);
                keep = score >= threshold;
            }
            const uint64_t mask = __ballot(keep);
            if (mask != 0) {
                if (keep) {
                    HMK_LDS uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
                    rec[0] = col;
                    rec[1] = r;
                    rec[2] = (uint32_t)score;
                }
                cnt += (uint32_t)__popcll(mask);
            }
        }
    }
    flush_stage<0>(stage, cnt, P, T, 0, false, shard);
}

hipError_t launch_neighbors_local_literal(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, const int32_t *d_matrix,
                                          int gap_open, int gap_extend, int threshold, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    const size_t lds = 2304 + 256 * SEQ_STRIDE_DW * 4 + 16 * 8 * 4 + 33 * 256 * 4 + 4 * 128 * 3 * 4;
    hipLaunchKernelGGL(k_neighbors_local_literal, dim3(n_tiles), dim3(256), lds, s, P, tile_base, d_matrix, gap_open, gap_extend,
                       threshold);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------
// force_signed / force_unpacked (the test switches HMK_LOCAL_SIGNED / HMK_LOCAL_NO_PK): the signed tagged-max form where the
// saturating one would run (it is what gap_open = 0 runs) / the one-column-per-lane form of the tagged-max DP.
hipError_t launch_neighbors_local(int lbmax, bool enc, bool force_signed, bool force_unpacked, const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles,
                                  const int32_t *d_matrix, int gap_open, int gap_extend, int threshold, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
#define HMK_NL(LB, E) hipLaunchKernelGGL((k_neighbors_local<LB, E>), dim3(n_tiles), dim3(256), 0, s, P, tile_base, d_matrix, \
                                         gap_open, gap_extend, threshold)
#define HMK_NLP1(LB, SAT) hipLaunchKernelGGL((k_neighbors_local_pk<LB, SAT>), dim3(n_tiles), dim3(256), 0, s, P, tile_base, d_matrix, \
                                             gap_open, gap_extend, threshold)
#define HMK_NLP(LB) do { if (sat) HMK_NLP1(LB, true); else HMK_NLP1(LB, false); } while (0)
    const bool sat = gap_open <= -1 && !force_signed;   // (enc already says -31 <= penalties <= 0)
    const bool packed = enc && !force_unpacked;   // two column sequences per lane
    if (lbmax <= 12) { if (packed) HMK_NLP(12); else if (enc) HMK_NL(12, true); else HMK_NL(12, false); }
    else if (lbmax <= 20) { if (packed) HMK_NLP(20); else if (enc) HMK_NL(20, true); else HMK_NL(20, false); }
    else { if (packed) HMK_NLP(32); else if (enc) HMK_NL(32, true); else HMK_NL(32, false); }
#undef HMK_NLP
#undef HMK_NLP1
#undef HMK_NL
    return hipGetLastError();
}

hipError_t launch_local_block(int lbmax, bool enc, bool force_signed, bool force_unpacked, const uint8_t *res32, const uint8_t *len, const int32_t *d_matrix,
                              uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int gap_open, int gap_extend, int32_t *out,
                              hipStream_t s) {
    if (r1 <= r0 || c1 <= c0) return hipSuccess;
    if (enc && !force_unpacked) {   // two column sequences per lane
        const dim3 grid2((c1 - c0 + 511) / 512, (r1 - r0 + 15) / 16);
        const bool sat = gap_open <= -1 && !force_signed;
#define HMK_LBP(LB) do { if (sat) hipLaunchKernelGGL((k_local_block_pk<LB, true>), grid2, dim3(256), 0, s, res32, len, d_matrix, r0, r1, c0, c1, gap_open, gap_extend, out); \
                         else hipLaunchKernelGGL((k_local_block_pk<LB, false>), grid2, dim3(256), 0, s, res32, len, d_matrix, r0, r1, c0, c1, gap_open, gap_extend, out); } while (0)
        if (lbmax <= 12) HMK_LBP(12); else if (lbmax <= 20) HMK_LBP(20); else HMK_LBP(32);
#undef HMK_LBP
        return hipGetLastError();
    }
    const dim3 grid((c1 - c0 + 255) / 256, (r1 - r0 + 15) / 16);
#define HMK_LB(LB, E) hipLaunchKernelGGL((k_local_block<LB, E>), grid, dim3(256), 0, s, res32, len, d_matrix, r0, r1, c0, c1, \
                                         gap_open, gap_extend, out)
    if (lbmax <= 20) { if (enc) HMK_LB(20, true); else HMK_LB(20, false); }
    else { if (enc) HMK_LB(32, true); else HMK_LB(32, false); }
#undef HMK_LB
    return hipGetLastError();
}

}  // namespace hmk
