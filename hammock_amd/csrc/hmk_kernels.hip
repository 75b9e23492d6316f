// hmk_kernels.hip -- hand-written gfx950 (CDNA4) kernels of libhammock_hip.so.
//
// Integer scoring only: no MFMA, no dense contraction.  The bound that matters
// is the LDS lookup rate (ds_read_b64: 32 lanes/clk/CU) and VALU issue, see
// DESIGN.md "Kernels".
//
//   k_neighbors_swar   all-vs-all ShiftedScorer (ShiftedScorer.java:48-95) with
//                      threshold -> edge list.  One (row, column) pair per lane
//                      per step; the row peptide is wave-uniform and expanded
//                      once per tile into per-position lookup tables in LDS
//                      whose entries hold ALL shifts of one column position as
//                      packed 8/16-bit lanes, so one ds_read_b64 + two v_add_u32
//                      advance all 7 shift sums of a pair by one residue.
//   k_neighbors_direct generic tier for length classes whose sums do not fit
//                      the packed lanes.
//   k_pairs_*          one pair per lane, literal loops: the parity probes
//                      behind hmk_score_pairs_* / hmk_score_block_*.
#include <hip/hip_runtime.h>
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

template <int SCORER>  // 0 shifted, 1 local
__global__ void __launch_bounds__(256)
k_pairs(const uint8_t *__restrict__ res32, const uint8_t *__restrict__ len, const int32_t *__restrict__ Mg,
        const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj, uint64_t n_pairs,
        uint32_t block_r0, uint32_t block_c0, uint32_t block_w,  // block mode when pi == nullptr
        int a, int b, int32_t *__restrict__ out, int32_t *__restrict__ out_shift) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *M = reinterpret_cast<int *>(smem);                                   // 576 dwords
    uint32_t *seqs = reinterpret_cast<uint32_t *>(smem + 2304);               // 256 * 2 * 9 dwords
    uint32_t *dp = seqs + 256 * 2 * SEQ_STRIDE_DW;                            // local: 33 * 256 dwords
    const int tid = threadIdx.x;
    for (int e = tid; e < 576; e += 256) M[e] = Mg[e];
    __syncthreads();
    uint32_t *s1 = seqs + tid * 2 * SEQ_STRIDE_DW;
    uint32_t *s2 = s1 + SEQ_STRIDE_DW;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + tid; k < n_pairs; k += (uint64_t)gridDim.x * 256) {
        uint32_t i, j;
        if (pi) { i = pi[k]; j = pj[k]; }
        else { i = block_r0 + (uint32_t)(k / block_w); j = block_c0 + (uint32_t)(k % block_w); }
        stage_sequence(s1, res32, i);
        stage_sequence(s2, res32, j);
        const int l1 = len[i], l2 = len[j];
        int score;
        if (SCORER == 0) {
            int shift = 0;
            score = shifted_score_literal(M, reinterpret_cast<const uint8_t *>(s1), l1,
                                          reinterpret_cast<const uint8_t *>(s2), l2, a, b, &shift);
            if (out_shift) out_shift[k] = shift;
        } else
            score = local_score_literal(M, reinterpret_cast<const uint8_t *>(s1), l1,
                                        reinterpret_cast<const uint8_t *>(s2), l2, a, b, dp + tid, 256);
        out[k] = score;
    }
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

// Drains one wave's staged records to its output segment.  REC_DW dwords per
// record: [0] column (sorted position), [1] row within the tile, [2..] the NW
// accumulator dwords (SWAR) or the score itself (direct, NW == 0).
template <int NW>
__device__ __forceinline__ void flush_stage(const uint32_t *stage, uint32_t cnt, const NeighborParams &P,
                                            const Tile &T, int g, bool lane16, uint32_t shard) {
    constexpr int REC_DW = (NW == 0) ? 3 : NW + 2;
    if (cnt == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // staged ds_writes land before the reads below
    const uint32_t lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&P.counts[shard], (unsigned long long)cnt);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t *rec = stage + k * REC_DW;
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
        uint32_t x = P.perm[T.row0 + r], m = P.perm[col];
        if (P.row_is_m || (P.symmetric && x > m)) { const uint32_t t = x; x = m; m = t; }
        const unsigned long long pos = base + k;
        if (pos < P.cap_per_shard)
            P.edges[(unsigned long long)shard * P.cap_per_shard + pos] =
                ((unsigned long long)x << 40) | ((unsigned long long)m << 16) | (unsigned long long)((uint32_t)score & 0xFFFFu);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // reads done before the stage is reused
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

// -----------------------------------------------------------------------------
// k_neighbors_swar: the hot kernel -- every sequence has length 12, max shift 3, 8-bit lanes
// -----------------------------------------------------------------------------
// (EXACT is always true; other lengths run k_neighbors_planes below.)
// LDS map (one __shared__ array):
//   tab     R * LBMAX * 24 * NW dwords   per row r, column position j, residue c:
//                                        NW dwords of packed lanes, lane t = shift index
//   mb      576 B                        biased matrix bytes
//   rowres  R * 32 B                     residues of the tile's rows
//   stage   4 waves * STAGE_CAP records  hits waiting to be written out
template <int NW, int R, int CPL, int LBMAX, bool EXACT>
// The production tiling (R = 6, CPL = 2) is held to 72 VGPRs = 7 waves/SIMD, which is also what its 22.6 KB of
// LDS allow per CU (80 VGPRs / 6 waves otherwise): +2.4 % measured.
__global__ void __launch_bounds__(256, (R == 6 && CPL == 2) ? 7 : 1)
k_neighbors_swar(const NeighborParams P, const uint32_t tile_base) {
    constexpr int ES = NW * 4;                 // table entry bytes
    constexpr int ROWBYTES = LBMAX * 24 * ES;  // one row's tables
    constexpr int TAB_BYTES = R * ROWBYTES;
    constexpr int STAGE_CAP = 128;             // records per wave; flushed when fewer than 64 slots are free
    constexpr int REC_DW = NW + 2;
    constexpr int LPADW = (LBMAX <= 16) ? 4 : 8;  // residue dwords a lane needs (rows are P.lpad bytes apart)
    static_assert(TAB_BYTES <= 65536, "row tables must stay addressable by the DS immediate offset");
    static_assert(EXACT && NW == 2 && LBMAX == 12, "this kernel is the length 12, max shift 3, 8-bit lane case");
    // one STATIC LDS object: its base address is a compile-time constant, so table
    // offsets fold into the ds_read immediate instead of costing a v_add per lookup
    constexpr int LDS_BYTES = TAB_BYTES + 576 + R * 32 + 4 * STAGE_CAP * REC_DW * 4;
    static_assert(LDS_BYTES <= 65536, "LDS budget");
    __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];
    uint8_t *tab = smem;
    uint8_t *mb = smem + TAB_BYTES;
    uint8_t *rowres = mb + 576;
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(rowres + R * 32);

    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const bool lane16 = Cp->path == PATH_U16;
    const int g = Cp->g;
    const uint32_t himask = lane16 ? 0x80008000u : 0x80808080u;
    const uint32_t shard = (tile_base + blockIdx.x) % HMK_EDGE_SHARDS;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    uint32_t *stage = stage_all + wave * (STAGE_CAP * REC_DW);

    // ---- stage the matrix and the row residues ---------------------------------
    for (int e = tid; e < 576; e += 256) mb[e] = P.mb[e];
    for (int e = tid; e < R * 32; e += 256) {
        const int r = e >> 5, k = e & 31;
        uint8_t v = 0;
        if ((uint32_t)r < T.nrows && (uint32_t)k < P.lpad) v = P.res_sorted[(size_t)(T.row0 + r) * P.lpad + k];
        rowres[e] = (uint8_t)(v >> 3);   // residues are stored pre-multiplied by the entry size (8)
    }
    __syncthreads();

    // ---- expand the R row peptides into lookup tables --------------------------
    // entry (r, j, c), lane t (shift s = t - X):
    //   column is the longer/equal one (L): cell = M[row[i]][c], i = j - s   (ShiftedScorer.java:71,75)
    //   column is the shorter one (S):      cell = M[c][row[i]], i = j + s
    if (EXACT) {
        // length 12, max shift 3, 8-bit lanes: one thread per (row, residue) gathers the 12 cells
        // P[i] = M[row[i]][c] once, reversed and zero padded, and cuts the 12 per-position entries out of
        // it as byte windows (lane t of position j is P[j + 3 - t]; lane 7 is unused)
        for (int e = tid; e < R * 24; e += 256) {
            const int r = e / 24, c = e - r * 24;
            uint32_t pp[6] = {0, 0, 0, 0, 0, 0};
            if ((uint32_t)r < T.nrows) {
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const uint32_t v = mb[rowres[r * 32 + i] * 24 + c];
                    pp[(14 - i) >> 2] |= v << (((14 - i) & 3) * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < 12; j++) {
                constexpr int dummy = 0; (void)dummy;
                const int v0 = 11 - j;
                const uint32_t lo = __builtin_amdgcn_alignbyte(pp[(v0 >> 2) + 1], pp[v0 >> 2], v0 & 3);
                const uint32_t hi = __builtin_amdgcn_alignbyte(pp[(v0 >> 2) + 2], pp[(v0 >> 2) + 1], v0 & 3) & 0x00FFFFFFu;
                *reinterpret_cast<u32x2 *>(tab + r * ROWBYTES + (j * 24 + c) * ES) = u32x2{lo, hi};
            }
        }
    }
    __syncthreads();

    uint32_t cinit[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) cinit[w] = Cp->cinit[w];

    uint32_t cnt = 0;  // staged records of this wave (wave-uniform)
    const uint32_t tab_addr = lds_addr(tab);
    const uint32_t col_end = T.col0 + T.ncols;
    const uint32_t n_batches = (T.ncols + 256 * CPL - 1) / (256 * CPL);
    const bool interior = T.diag == 0 && T.ncols % (256 * CPL) == 0;  // every lane's column is a real pair

    for (uint32_t bt = 0; bt < n_batches; bt++) {
        // ---- this lane's CPL column peptides -> per-position table offsets ------
        uint32_t off[CPL][LBMAX];
        uint32_t colpos[CPL];
#pragma unroll
        for (int p = 0; p < CPL; p++) {
            const uint32_t col = T.col0 + (bt * CPL + p) * 256 + tid;
            colpos[p] = col;
            uint32_t words[LPADW];
#pragma unroll
            for (int q = 0; q < LPADW; q++) words[q] = 0;
            if (col < col_end) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(P.res_sorted + (size_t)col * P.lpad);
                const u32x4 v0 = src[0];
                words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
                if (LPADW == 8) {
                    const u32x4 v1 = src[1];
                    words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
                }
            }
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                const uint32_t c = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                off[p][j] = tab_addr + (uint32_t)(j * 24 * ES) + c;   // c is stored as residue * 8
            }
        }

#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((uint32_t)r < T.nrows) {
                // all CPL accumulations first (CPL * LB independent LDS reads in flight), tests after
                uint32_t W[CPL][NW];
#pragma unroll
                for (int p = 0; p < CPL; p++) {
#pragma unroll
                    for (int w = 0; w < NW; w++) W[p][w] = cinit[w];
#pragma unroll
                    for (int j = 0; j < LBMAX; j++) {
                        {
                            const uint32_t ea = off[p][j] + (uint32_t)(r * ROWBYTES);
                            if (NW == 1) {
                                W[p][0] += lds_read<uint32_t>(ea);
                            } else if (NW == 2) {
                                const u32x2 e = lds_read<u32x2>(ea);
                                W[p][0] += e.x; W[p][1] += e.y;
                            } else {
#pragma unroll
                                for (int q = 0; q < NW / 4; q++) {
                                    const u32x4 e = lds_read<u32x4>(ea + 16 * q);
                                    W[p][4 * q + 0] += e.x; W[p][4 * q + 1] += e.y;
                                    W[p][4 * q + 2] += e.z; W[p][4 * q + 3] += e.w;
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < CPL; p++) {
                    uint32_t any = W[p][0];
#pragma unroll
                    for (int w = 1; w < NW; w++) any |= W[p][w];
                    const bool hit = (any & himask) != 0;  // some shift reached score >= threshold
                    if (__ballot(hit) != 0) {              // wave-uniform, rare
                        if (cnt > (uint32_t)(STAGE_CAP - 64)) {  // keep room for one wave of hits
                            flush_stage<NW>(stage, cnt, P, T, g, lane16, shard);
                            cnt = 0;
                        }
                        const uint32_t col = colpos[p];
                        bool keep = hit;
                        if (!interior) {  // wave-uniform: only edge tiles filter
                            keep = keep && col < col_end;
                            if (T.diag == 1) keep = keep && col > T.row0 + r;   // triangle: column after row
                            if (T.diag == 2) keep = keep && col != T.row0 + r;  // full square minus the diagonal
                        }
                        const uint64_t mask = __ballot(keep);
                        if (keep) {
                            uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
                            rec[0] = col;
                            rec[1] = (uint32_t)r;
#pragma unroll
                            for (int w = 0; w < NW; w++) rec[2 + w] = W[p][w];
                        }
                        cnt += (uint32_t)__popcll(mask);
                    }
                }
            }
        }
    }
    flush_stage<NW>(stage, cnt, P, T, g, lane16, shard);
}

// -----------------------------------------------------------------------------
// k_neighbors_planes: the SWAR neighbour kernel for arbitrary lengths
// -----------------------------------------------------------------------------
// Same tables as k_neighbors_swar (entry (r, j, c) = NW dwords of packed shift lanes), stored as
// PLANES so that every LDS read is bank-conflict free: NW / 2 planes of 8-byte entries plus, for odd NW,
// one plane of 4-byte entries.  Inside a plane the 24 residues of one position are 24 consecutive
// entries = 48 (24) consecutive banks, so lanes reading different residues never share a bank; with
// 16- or 32-byte entries read as ds_read_b128 residues c and c + 16 (c + 8) collide and 43-59 % of the
// LDS cycles were conflict cycles (PMC, profiles/round1_pmc_config4a.json).  Any NW from 1 to 8 is its
// own instantiation (no rounding of the shift count up to a power of two).
//
// LDS map:  tab  R rows x [NP planes x LBMAX x 24 x 8 B | H x LBMAX x 24 x 4 B]
//           mb 576 B, rowres R x 32 B, stage 4 waves x 128 records x 3 dwords (col, row, score)
// Plane strides.  The compiler fuses two ds_reads off the same address register into ds_read2[st64]
// when their immediates differ by < 2048 B or by a multiple of 512 B (256 B for 4-byte reads); a fused
// read whose halves are a multiple of 256 B apart hits the same banks with both halves (measured: 43 %
// conflict cycles with strides of 1536 / 3072 B).  Strides of 8 x odd bytes, >= 2048, rule the fusion out;
// planes_layout_ok() checks every pair of (row, plane) offsets at compile time.
constexpr int plane64_bytes(int lbmax) { return lbmax * 24 * 8 + 8; }
constexpr int rows_for(int rowbytes, int nw) {  // rows per tile: table bytes and R x NW accumulator registers, tuned on config 4a
#ifndef HMK_TAB_BUDGET
#define HMK_TAB_BUDGET 20480
#endif
#ifndef HMK_ACC_CAP
#define HMK_ACC_CAP 16
#endif
    int r = HMK_TAB_BUDGET / rowbytes;
    if (r > HMK_ACC_CAP / nw) r = HMK_ACC_CAP / nw;
    return r > 16 ? 16 : (r < 1 ? 1 : r);
}
constexpr bool planes_layout_ok(int lbmax, int nw, int pad32) {
    const int np = nw / 2, s64 = plane64_bytes(lbmax);
    const int rb = np * s64 + (nw & 1) * (lbmax * 24 * 4 + pad32);
    const int r_rows = rows_for(rb, nw);
    for (int a = 0; a < r_rows * np; a++)
        for (int b = a + 1; b < r_rows * np; b++) {
            const int d = ((b / np) * rb + (b % np) * s64) - ((a / np) * rb + (a % np) * s64);
            if (d < 2048 || d % 512 == 0) return false;
        }
    if (nw & 1)
        for (int k = 1; k < r_rows; k++)
            if (k * rb < 1024 || (k * rb) % 256 == 0) return false;
    return true;
}
constexpr int plane32_bytes(int lbmax, int nw) {  // smallest padding of the 4-byte plane that passes the check
    for (int pad = 8; pad <= 256; pad += 8)
        if (planes_layout_ok(lbmax, nw, pad)) return lbmax * 24 * 4 + pad;
    return -1;
}
constexpr int planes_rowbytes(int lbmax, int nw) { return (nw / 2) * plane64_bytes(lbmax) + (nw & 1) * plane32_bytes(lbmax, nw); }

template <int NW, int R, int LBMAX>
__global__ void __launch_bounds__(256) k_neighbors_planes(const NeighborParams P, const uint32_t tile_base) {
    constexpr int NP = NW / 2, H = NW & 1;
    constexpr int CPL = NW <= 2 ? 2 : 1;
    constexpr int PLANE64 = plane64_bytes(LBMAX);
    constexpr int ROWBYTES = planes_rowbytes(LBMAX, NW);
    static_assert(plane32_bytes(LBMAX, NW) > 0 && R == rows_for(ROWBYTES, NW),
                  "no padding found that keeps table reads from being fused into a same-bank ds_read2");
    constexpr int TAB_BYTES = R * ROWBYTES;
    constexpr int STAGE_CAP = 128, REC_DW = 3;
    constexpr int LPADW = (LBMAX <= 16) ? 4 : 8;
    constexpr int LDS_BYTES = TAB_BYTES + 576 + R * 32 + 4 * STAGE_CAP * REC_DW * 4;
    static_assert(TAB_BYTES <= 65536 && LDS_BYTES <= 65536, "LDS budget / DS immediate range");
    __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];
    uint8_t *tab = smem;
    uint8_t *mb = smem + TAB_BYTES;
    uint8_t *rowres = mb + 576;
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(rowres + R * 32);

    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la, lb = Cp->lb, nd = Cp->nd, X = Cp->x;
    const bool case_b = Cp->case_b != 0;
    const bool lane16 = Cp->path == PATH_U16;
    const int g = Cp->g;
    const uint32_t himask = lane16 ? 0x80008000u : 0x80808080u;
    const uint32_t shard = (tile_base + blockIdx.x) % HMK_EDGE_SHARDS;
    const int tid = threadIdx.x;
    uint32_t *stage = stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);

    for (int e = tid; e < 576; e += 256) mb[e] = P.mb[e];
    for (int e = tid; e < R * 32; e += 256) {
        const int r = e >> 5, k = e & 31;
        uint8_t v = 0;
        if ((uint32_t)r < T.nrows && (uint32_t)k < P.lpad) v = P.res_sorted[(size_t)(T.row0 + r) * P.lpad + k];
        rowres[e] = v;
    }
    __syncthreads();

    // ---- expand the R row peptides into lookup tables (same cells as k_neighbors_swar) ----
    //   column is the longer/equal one (L): cell = M[row[i]][c], i = j - s   (ShiftedScorer.java:71,75)
    //   column is the shorter one (S):      cell = M[c][row[i]], i = j + s
    {
        const int per_row = lb * 24;
        const int lanes_per_dw = lane16 ? 2 : 4;
        const int lane_bits = lane16 ? 16 : 8;
        for (int e = tid; e < R * per_row; e += 256) {
            const int r = e / per_row;
            const int rem = e - r * per_row;
            const int j = rem / 24;
            const int c = rem - j * 24;
            uint32_t dw[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                uint32_t acc = 0;
                for (int k = 0; k < lanes_per_dw; k++) {
                    const int t = w * lanes_per_dw + k;
                    const int i = case_b ? (j + t - X) : (j - t + X);
                    if (t < nd && i >= 0 && i < la && (uint32_t)r < T.nrows) {
                        const int a = rowres[r * 32 + i];
                        const uint32_t v = case_b ? mb[c * 24 + a] : mb[a * 24 + c];
                        acc |= v << (k * lane_bits);
                    }
                }
                dw[w] = acc;
            }
            uint8_t *row = tab + r * ROWBYTES;
#pragma unroll
            for (int q = 0; q < NP; q++)
                *reinterpret_cast<u32x2 *>(row + q * PLANE64 + (j * 24 + c) * 8) = u32x2{dw[2 * q], dw[2 * q + 1]};
            if (H) *reinterpret_cast<uint32_t *>(row + NP * PLANE64 + (j * 24 + c) * 4) = dw[NW - 1];
        }
    }
    __syncthreads();

    uint32_t cinit[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) cinit[w] = Cp->cinit[w];

    uint32_t cnt = 0;  // staged records of this wave (wave-uniform)
    const uint32_t tab_addr = lds_addr(tab);
    const uint32_t col_end = T.col0 + T.ncols;
    const uint32_t n_batches = (T.ncols + 256 * CPL - 1) / (256 * CPL);
    const bool interior = T.diag == 0 && T.ncols % (256 * CPL) == 0;  // every lane's column is a real pair

    for (uint32_t bt = 0; bt < n_batches; bt++) {
#pragma unroll
        for (int p = 0; p < CPL; p++) {
            // ---- this lane's column peptide -> per-position entry index ----------------
            const uint32_t col = T.col0 + (bt * CPL + p) * 256 + tid;
            uint32_t words[LPADW];
#pragma unroll
            for (int q = 0; q < LPADW; q++) words[q] = 0;
            if (col < col_end) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(P.res_sorted + (size_t)col * P.lpad);
                const u32x4 v0 = src[0];
                words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
                if (LPADW == 8) {
                    const u32x4 v1 = src[1];
                    words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
                }
            }
            uint32_t off64[NP ? LBMAX : 1], off32[H ? LBMAX : 1];
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                const uint32_t c = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                if (NP) off64[j] = tab_addr + (uint32_t)(j * 24 * 8) + c * 8;
                if (H) off32[j] = tab_addr + (uint32_t)(NP * PLANE64 + j * 24 * 4) + c * 4;
            }

            // ---- position-major accumulation: the `j < lb` tests are wave-uniform branches; inside one
            // branch the reads of all R rows for two positions are in flight together, and a pair of
            // positions costs one v_add3 per accumulator dword.  Rows past T.nrows read zero tables.
            uint32_t W[R][NW];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int w = 0; w < NW; w++) W[r][w] = cinit[w];
            auto read_entry = [&](int j, int r, uint32_t (&e)[NW]) {
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    const u32x2 v = lds_read<u32x2>(off64[NP ? j : 0] + (uint32_t)(r * ROWBYTES + q * PLANE64));
                    e[2 * q] = v.x; e[2 * q + 1] = v.y;
                }
                if (H) e[NW - 1] = lds_read<uint32_t>(off32[H ? j : 0] + (uint32_t)(r * ROWBYTES));
            };
#pragma unroll
            for (int j = 0; j < LBMAX; j += 2) {
                if (j + 1 < lb) {
                    uint32_t e0[R][NW], e1[R][NW];
#pragma unroll
                    for (int r = 0; r < R; r++) { read_entry(j, r, e0[r]); read_entry(j + 1, r, e1[r]); }
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int w = 0; w < NW; w++) W[r][w] = W[r][w] + e0[r][w] + e1[r][w];
                } else if (j < lb) {
                    uint32_t e0[R][NW];
#pragma unroll
                    for (int r = 0; r < R; r++) read_entry(j, r, e0[r]);
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int w = 0; w < NW; w++) W[r][w] += e0[r][w];
                }
            }

            // ---- threshold test: some shift lane has its top bit set <=> score >= threshold ----
            // one combined test for the R rows first: most batches hold no hit at all
            uint32_t all = 0;
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int w = 0; w < NW; w++) all |= W[r][w];
            if (__ballot((all & himask) != 0) == 0) continue;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t any = W[r][0];
#pragma unroll
                for (int w = 1; w < NW; w++) any |= W[r][w];
                const bool hit = (any & himask) != 0 && (uint32_t)r < T.nrows;
                if (__ballot(hit) != 0) {  // wave-uniform, rare
                    if (cnt > (uint32_t)(STAGE_CAP - 64)) {
                        flush_stage<0>(stage, cnt, P, T, 0, false, shard);
                        cnt = 0;
                    }
                    bool keep = hit;
                    if (!interior) {
                        keep = keep && col < col_end;
                        if (T.diag == 1) keep = keep && col > T.row0 + r;   // triangle: column after row
                        if (T.diag == 2) keep = keep && col != T.row0 + r;  // full square minus the diagonal
                    }
                    const uint64_t mask = __ballot(keep);
                    if (keep) {
                        uint32_t mx = 0;  // best shift = largest lane
#pragma unroll
                        for (int w = 0; w < NW; w++) {
                            const uint32_t dw = W[r][w];
                            if (lane16) mx = max(mx, max(dw & 0xFFFFu, dw >> 16));
                            else mx = max(mx, max(max(dw & 0xFFu, (dw >> 8) & 0xFFu), max((dw >> 16) & 0xFFu, dw >> 24)));
                        }
                        uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
                        rec[0] = col;
                        rec[1] = (uint32_t)r;
                        rec[2] = (uint32_t)((int)mx - g);
                    }
                    cnt += (uint32_t)__popcll(mask);
                }
            }
        }
    }
    flush_stage<0>(stage, cnt, P, T, 0, false, shard);
}

// -----------------------------------------------------------------------------
// k_neighbors_direct: generic tier (same tiles, literal scorer, one column per lane)
// -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_neighbors_direct(const NeighborParams P, const uint32_t tile_base, const int32_t *__restrict__ Mg,
                   int max_shift, int shift_penalty, int threshold) {
    constexpr int R = 16;
    constexpr int STAGE_CAP = 128;
    constexpr int REC_DW = 3;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *M = reinterpret_cast<int *>(smem);                                   // 2304 B
    uint32_t *colseq = reinterpret_cast<uint32_t *>(smem + 2304);             // 256 * 9 dwords
    uint32_t *rowseq = colseq + 256 * SEQ_STRIDE_DW;                          // R * 8 dwords
    uint32_t *stage_all = rowseq + R * 8;

    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la, lb = Cp->lb;
    const uint32_t shard = (tile_base + blockIdx.x) % HMK_EDGE_SHARDS;
    const int tid = threadIdx.x;
    uint32_t *stage = stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);

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
            if (T.diag == 1) keep = keep && col > T.row0 + r;
            if (T.diag == 2) keep = keep && col != T.row0 + r;
            int score = 0;
            if (keep) {
                // edge (x = row, m = column) carries sequenceScore(seq1 = m, seq2 = x)
                score = shifted_score_literal(M, reinterpret_cast<const uint8_t *>(mine), lb,
                                              reinterpret_cast<const uint8_t *>(rowseq + r * 8), la,
                                              max_shift, shift_penalty);
                keep = score >= threshold;
            }
            const uint64_t mask = __ballot(keep);
            if (mask != 0) {
                if (keep) {
                    uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
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
    const uint32_t shard = (tile_base + blockIdx.x) % HMK_EDGE_SHARDS;
    const int tid = threadIdx.x;
    uint32_t *stage = stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);

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
                    uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
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

template <int LBMAX>
__device__ __forceinline__ uint32_t sw_row_pk(uint32_t row_q_addr, int strips, int ncols, const uint32_t (&boff_lo)[LBMAX],
                                              const uint32_t (&boff_hi)[LBMAX], int gap_open, int gap_extend) {
    // 16-bit table entries by tag (0 NOWHERE, 1 LEFT, 2 UP, 3 DIAGONAL), split into low / high bytes
    const int pu[4] = {4 * gap_open + 2, 4 * gap_open + 1, 4 * gap_extend, 4 * gap_open - 1};
    const int pl[4] = {4 * gap_open + 1, 4 * gap_extend, 4 * gap_open - 1, 4 * gap_open - 2};
    uint32_t PUlo = 0, PUhi = 0, PLlo = 0, PLhi = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        PUlo |= ((uint32_t)pu[t] & 0xFFu) << (8 * t); PUhi |= (((uint32_t)pu[t] >> 8) & 0xFFu) << (8 * t);
        PLlo |= ((uint32_t)pl[t] & 0xFFu) << (8 * t); PLhi |= (((uint32_t)pl[t] >> 8) & 0xFFu) << (8 * t);
    }
    // v_perm_b32 reads at most one SGPR: pin one table of each pair in a VGPR once instead of a
    // v_mov per use
    asm volatile("" : "+v"(PUlo));
    asm volatile("" : "+v"(PLlo));
    const uint32_t both = 0x00010001u;
    const s16x2 zero = pk(0u);
    uint32_t H[LBMAX], U[LBMAX];
#pragma unroll
    for (int j = 0; j < LBMAX; j++) { H[j] = 3u * both; U[j] = (uint32_t)((4 * gap_open + 2) & 0xFFFF) * both; }
    s16x2 gm = zero;
    for (int st = 0; st < strips; st++) {
        const uint32_t strip_addr = row_q_addr + (uint32_t)st * 8u;
        uint32_t hd[4], lc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { hd[k] = 3u * both; lc[k] = (uint32_t)((4 * gap_open + 1) & 0xFFFF) * both; }
#pragma unroll
        for (int j = 0; j < LBMAX; j++) {
            if (j < ncols) {
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
                for (int k = 0; k < 4; k++) {
                    const s16x2 diag = pk(hd[k]) + pk(sc[k]);                                        // :59
                    const s16x2 m1 = __builtin_elementwise_max(pk(up), pk(lc[k]));
                    const s16x2 he = __builtin_elementwise_max(__builtin_elementwise_max(diag, m1), zero);  // :61-67
                    gm = __builtin_elementwise_max(gm, he);                                          // :68-72
                    const uint32_t heu = un(he);
                    const uint32_t sel = (heu & 0x00030003u) * 0x0101u + 0x04000400u;
                    const uint32_t padd_u = __builtin_amdgcn_perm(PUhi, PUlo, sel);
                    const uint32_t padd_l = __builtin_amdgcn_perm(PLhi, PLlo, sel);
                    hd[k] = habove;
                    habove = heu | 0x00030003u;
                    up = un(he + pk(padd_u));                                                         // :43-48,:57
                    lc[k] = un(he + pk(padd_l));                                                      // :50-55,:58
                }
                H[j] = habove;
                U[j] = up;
            }
        }
    }
    const uint32_t g = un(gm);
    return ((g & 0xFFFFu) >> 2) | (((g >> 16) >> 2) << 16);   // both maxima are >= 0
}

template <int LBMAX>
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
    const uint32_t shard = (tile_base + blockIdx.x) % HMK_EDGE_SHARDS;
    const int tid = threadIdx.x;
    uint32_t *stage = stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);

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
            const uint32_t g2 = sw_row_pk<LBMAX>(q_addr + r * QROW, strips, lb, boff[0], boff[1], gap_open, gap_extend);
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
                        uint32_t *rec = stage + (cnt + mbcnt64(mask)) * REC_DW;
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
// edge list -> CSR adjacency on the device (feeds the host greedy merge)
// -----------------------------------------------------------------------------
// The neighbour kernel leaves HMK_EDGE_SHARDS segments of packed edges.  Three small
// passes turn them into start[n + 1] / adj[] (each undirected edge stored under both
// ends when the matrix is symmetric): degree count, exclusive scan, scatter.  Order
// inside a row is arbitrary (atomic cursors); the merge does not depend on it.
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

// score_range[0] / [1]: smallest / largest edge score (decides whether the 4-byte adjacency fits);
// score_range[2]: edges that name a sequence outside [0, n) or a self pair (not counted; the caller gives up)
__global__ void __launch_bounds__(256)
k_edge_degree(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
              uint32_t *__restrict__ deg, int symmetric, int *__restrict__ score_range, uint32_t n) {
    const uint32_t shard = blockIdx.y;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    int lo = INT_MAX, hi = INT_MIN;
    // wave-uniform loop: the x side of a wave's 64 consecutive edges has a handful of distinct values
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); k0 < cnt; k0 += (uint64_t)gridDim.x * 256) {
        const uint64_t k = k0 + (threadIdx.x & 63);
        bool valid = k < cnt;
        const uint64_t e = valid ? seg[k] : 0;
        if (valid && (HMK_EDGE_X(e) >= n || HMK_EDGE_M(e) >= n || HMK_EDGE_X(e) == HMK_EDGE_M(e))) {
            atomicAdd(&score_range[2], 1);
            valid = false;
        }
        const WaveGroup g = wave_groups(HMK_EDGE_X(e), valid);
        if (valid && g.rank == 0) atomicAdd(&deg[HMK_EDGE_X(e)], g.size);
        if (valid && symmetric) atomicAdd(&deg[HMK_EDGE_M(e)], 1u);
        if (valid) {
            const int sc = HMK_EDGE_SCORE(e);
            lo = min(lo, sc);
            hi = max(hi, sc);
        }
    }
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, __shfl_down(lo, o, 64));
        hi = max(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&score_range[0], lo);
        atomicMax(&score_range[1], hi);
    }
}

// Exclusive scan of deg[n] -> start[n + 1] in three coalesced passes over tiles of 2048 counters:
// tile sums, scan of the tile sums (one block), tile-local scan + offset.  T = uint32 or uint64.
constexpr uint32_t SCAN_TILE = 2048;

__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *__restrict__ deg, uint64_t *__restrict__ tile_sum, uint32_t n) {
    __shared__ uint64_t red[4];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint64_t v = 0;
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {
        const uint32_t k = base + j * 256 + threadIdx.x;
        if (k < n) v += deg[k];
    }
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one block: exclusive scan of the n_tiles tile sums in place; tile_sum[n_tiles] = grand total
__global__ void __launch_bounds__(1024) k_scan_tile_offsets(uint64_t *__restrict__ tile_sum, uint32_t n_tiles) {
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n_tiles + 1023) / 1024;
    const uint32_t lo = min(n_tiles, tid * chunk), hi = min(n_tiles, lo + chunk);
    uint64_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += tile_sum[k];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    for (uint32_t k = lo; k < hi; k++) { const uint64_t d = tile_sum[k]; tile_sum[k] = run; run += d; }
    if (tid == 1023) tile_sum[n_tiles] = part[1023];
}

template <typename T>
__global__ void __launch_bounds__(256)
k_scan_tiles(const uint32_t *__restrict__ deg, const uint64_t *__restrict__ tile_off, T *__restrict__ start, uint32_t n,
             uint32_t n_tiles, const uint32_t *__restrict__ tail_word) {
    __shared__ uint32_t v[SCAN_TILE];
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * SCAN_TILE, tid = threadIdx.x;
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {  // coalesced load
        const uint32_t k = base + j * 256 + tid;
        v[j * 256 + tid] = k < n ? deg[k] : 0;
    }
    __syncthreads();
    uint32_t loc[SCAN_TILE / 256], sum = 0;           // thread tid owns the 8 consecutive counters tid * 8 ..
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) { loc[j] = sum; sum += v[tid * (SCAN_TILE / 256) + j]; }
    uint32_t inc = sum;                               // inclusive scan of the thread sums inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if ((tid & 63) >= (uint32_t)o) inc += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) woff += wsum[w];
    const uint32_t excl = woff + inc - sum;
    __syncthreads();
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) v[tid * (SCAN_TILE / 256) + j] = excl + loc[j];
    __syncthreads();
    const uint64_t off = tile_off[blockIdx.x];
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {  // coalesced store
        const uint32_t k = base + j * 256 + tid;
        if (k < n) start[k] = (T)(off + v[j * 256 + tid]);
    }
    if (blockIdx.x == 0 && tid == 0) {
        start[n] = (T)tile_off[n_tiles];
        if (tail_word) start[n + 1] = (T)*tail_word;
    }
}

// tile_scratch: uint64[ceil(n / SCAN_TILE) + 1]
template <typename T>
static void launch_scan(const uint32_t *deg, T *start, uint32_t n, uint64_t *tile_scratch, const uint32_t *tail_word,
                        hipStream_t s) {
    const uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, s, deg, tile_scratch, n);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(1024), 0, s, tile_scratch, n_tiles);
    hipLaunchKernelGGL((k_scan_tiles<T>), dim3(n_tiles), dim3(256), 0, s, deg, tile_scratch, start, n, n_tiles, tail_word);
}

// NbrT = Nbr: {m, score}.  NbrT = NbrPacked: m << 8 | (score - base), the caller has checked the score range.
template <class NbrT>
__global__ void __launch_bounds__(256)
k_edge_scatter(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
               const uint64_t *__restrict__ start, uint32_t *__restrict__ cursor, NbrT *__restrict__ adj, int symmetric,
               int base) {
    const uint32_t shard = blockIdx.y;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); k0 < cnt; k0 += (uint64_t)gridDim.x * 256) {
        const uint64_t k = k0 + (threadIdx.x & 63);
        const bool valid = k < cnt;
        const uint64_t e = valid ? seg[k] : 0;
        const uint32_t x = HMK_EDGE_X(e), m = HMK_EDGE_M(e);
        const int32_t s = HMK_EDGE_SCORE(e);
        const WaveGroup g = wave_groups(x, valid);   // one atomic per distinct x of the wave
        uint32_t basex = 0;
        if (valid && g.rank == 0) basex = atomicAdd(&cursor[x], g.size);
        basex = (uint32_t)__shfl((int)basex, (int)g.leader, 64);
        if (!valid) continue;
        const uint64_t px = start[x] + basex + g.rank;
        if constexpr (sizeof(NbrT) == 4) {
            const uint32_t rel = (uint32_t)(s - base) & 0xFFu;
            adj[px] = NbrT{(m << 8) | rel};
            if (symmetric) adj[start[m] + atomicAdd(&cursor[m], 1u)] = NbrT{(x << 8) | rel};
        } else {
            adj[px] = NbrT{m, s};
            if (symmetric) adj[start[m] + atomicAdd(&cursor[m], 1u)] = NbrT{x, s};
        }
    }
}

// The HMK_EDGE_SHARDS output segments -> one contiguous block (device to device), so a
// fixed-size collective can ship a rank's edges without any host round trip.
__global__ void __launch_bounds__(256)
k_compact_edges(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
                uint64_t *__restrict__ out, uint64_t out_capacity, unsigned long long *__restrict__ total) {
    const uint32_t shard = blockIdx.y;
    uint64_t base = 0, all = 0;
    for (uint32_t q = 0; q < HMK_EDGE_SHARDS; q++) {
        const uint64_t c = min((uint64_t)counts[q], cap_per_shard);
        if (q < shard) base += c;
        all += c;
    }
    if (blockIdx.x == 0 && shard == 0 && threadIdx.x == 0) *total = all;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < cnt; k += (uint64_t)gridDim.x * 256)
        if (base + k < out_capacity) out[base + k] = seg[k];
}

// -----------------------------------------------------------------------------
// "row blocks": the 4-byte-per-edge exchange format of the multi-GPU path
// -----------------------------------------------------------------------------
// A rank's edge segments regrouped by x: row_start[n + 2] (uint32; [n] = total, [n + 1] =
// number of edges whose score - threshold did not fit 8 bits, must be 0) and one uint32 per
// edge, m << 8 | (score - threshold).  Halves the bytes the all-gather ships over xGMI.
constexpr int ROWS_UNROLL = 4;  // independent 64-edge groups per wave iteration (memory-level parallelism)

__global__ void __launch_bounds__(256)
k_rows_degree(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
              uint32_t *__restrict__ deg, uint32_t *__restrict__ misfit, int threshold) {
    const uint32_t shard = blockIdx.y, lane = threadIdx.x & 63;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t k0 = wave * (64 * ROWS_UNROLL); k0 < cnt; k0 += n_waves * (64 * ROWS_UNROLL)) {
        uint64_t e[ROWS_UNROLL];
        bool valid[ROWS_UNROLL];
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint64_t k = k0 + u * 64 + lane;
            valid[u] = k < cnt;
            e[u] = valid[u] ? seg[k] : 0;
        }
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint32_t x = HMK_EDGE_X(e[u]);
            const WaveGroup g = wave_groups(x, valid[u]);
            if (valid[u] && g.rank == 0) atomicAdd(&deg[x], g.size);
            const int32_t rel = HMK_EDGE_SCORE(e[u]) - threshold;
            if (valid[u] && (rel < 0 || rel > 255)) atomicAdd(misfit, 1u);
        }
    }
}

__global__ void __launch_bounds__(256)
k_rows_scatter(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
               const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t *__restrict__ adj,
               uint64_t adj_capacity, int threshold) {
    const uint32_t shard = blockIdx.y, lane = threadIdx.x & 63;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t k0 = wave * (64 * ROWS_UNROLL); k0 < cnt; k0 += n_waves * (64 * ROWS_UNROLL)) {
        uint64_t e[ROWS_UNROLL];
        bool valid[ROWS_UNROLL];
        WaveGroup g[ROWS_UNROLL];
        uint32_t base[ROWS_UNROLL];
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint64_t k = k0 + u * 64 + lane;
            valid[u] = k < cnt;
            e[u] = valid[u] ? seg[k] : 0;
        }
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {  // the group leaders' atomics of all groups are in flight together
            const uint32_t x = HMK_EDGE_X(e[u]);
            g[u] = wave_groups(x, valid[u]);
            base[u] = 0;
            if (valid[u] && g[u].rank == 0) base[u] = start[x] + atomicAdd(&cursor[x], g[u].size);
        }
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint32_t b = (uint32_t)__shfl((int)base[u], (int)g[u].leader, 64);
            if (valid[u]) {
                const uint64_t pos = (uint64_t)b + g[u].rank;
                if (pos < adj_capacity)
                    adj[pos] = (HMK_EDGE_M(e[u]) << 8) | (uint32_t)((HMK_EDGE_SCORE(e[u]) - threshold) & 0xFF);
            }
        }
    }
}

// row blocks -> packed 8-byte edges, out[start[x] + k]; one wave per row, rows strided over the grid
__global__ void __launch_bounds__(256)
k_rows_unpack(const uint32_t *__restrict__ start, const uint32_t *__restrict__ adj, uint32_t n, int threshold,
              uint64_t *__restrict__ out, uint64_t out_capacity) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (uint32_t x = wave; x < n; x += n_waves) {
        const uint32_t b = start[x], e = start[x + 1];
        for (uint32_t k = b + lane; k < e; k += 64) {
            const uint32_t a = adj[k];
            const int32_t sc = (int32_t)(a & 0xFF) + threshold;
            if (k < out_capacity) out[k] = ((uint64_t)x << 40) | ((uint64_t)(a >> 8) << 16) | (uint64_t)(uint16_t)(int16_t)sc;
        }
    }
}

// -----------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------
template <int NW, int R, int CPL, int LBMAX, bool EXACT>
static hipError_t launch_swar_t(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, hipStream_t s) {
    hipLaunchKernelGGL((k_neighbors_swar<NW, R, CPL, LBMAX, EXACT>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
    return hipGetLastError();
}

// Hot-path tilings of the exact length-12, NW = 2 kernel: {rows per tile, columns per lane}.
static const int kHotVariants[][2] = {{16, 2}, {8, 2}, {12, 2}, {8, 4}, {5, 2}, {7, 2}, {4, 2}, {6, 2}, {6, 3}, {6, 1}};
constexpr int kNumHotVariants = sizeof(kHotVariants) / sizeof(kHotVariants[0]);

// Generic instantiations: column-length capacity LBMAX x dwords per entry NW.  Rows per tile
// R = what fits a 40 KB table budget (<= 16); 2 columns per lane for the narrow entries.
constexpr int swar_r(int lbmax, int nw) {
    return rows_for(planes_rowbytes(lbmax, nw), nw);
}

int swar_lbmax_for(int lb) { return lb <= 12 ? 12 : lb <= 16 ? 16 : lb <= 20 ? 20 : 32; }  // plane strides need >= 11 positions

int swar_rows_per_tile(int lbmax, int nw, bool exact, int hot_variant) {
    if (exact && lbmax == 12 && nw == 2 && hot_variant >= 0 && hot_variant < kNumHotVariants)
        return kHotVariants[hot_variant][0];
    return swar_r(lbmax, nw);
}

hipError_t launch_neighbors_swar(int lbmax, int nw, bool exact, int hot_variant, const NeighborParams &P,
                                 uint32_t tile_base, uint32_t n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    if (exact && lbmax == 12 && nw == 2) {
        switch (hot_variant) {
            case 0: return launch_swar_t<2, 16, 2, 12, true>(P, tile_base, n_tiles, s);
            case 1: return launch_swar_t<2, 8, 2, 12, true>(P, tile_base, n_tiles, s);
            case 2: return launch_swar_t<2, 12, 2, 12, true>(P, tile_base, n_tiles, s);
            case 3: return launch_swar_t<2, 8, 4, 12, true>(P, tile_base, n_tiles, s);
            case 4: return launch_swar_t<2, 5, 2, 12, true>(P, tile_base, n_tiles, s);
            case 5: return launch_swar_t<2, 7, 2, 12, true>(P, tile_base, n_tiles, s);
            case 6: return launch_swar_t<2, 4, 2, 12, true>(P, tile_base, n_tiles, s);
            case 7: return launch_swar_t<2, 6, 2, 12, true>(P, tile_base, n_tiles, s);
            case 8: return launch_swar_t<2, 6, 3, 12, true>(P, tile_base, n_tiles, s);
            case 9: return launch_swar_t<2, 6, 1, 12, true>(P, tile_base, n_tiles, s);
            default: return hipErrorInvalidValue;
        }
    }
#define HMK_CASE(LB, NWV)                                                                                             \
    if (lbmax == LB && nw == NWV) {                                                                                   \
        hipLaunchKernelGGL((k_neighbors_planes<NWV, swar_r(LB, NWV), LB>), dim3(n_tiles), dim3(256), 0, s, P, tile_base); \
        return hipGetLastError();                                                                                     \
    }
#define HMK_CASES(LB) HMK_CASE(LB, 1) HMK_CASE(LB, 2) HMK_CASE(LB, 3) HMK_CASE(LB, 4) HMK_CASE(LB, 5) HMK_CASE(LB, 6) \
                      HMK_CASE(LB, 7) HMK_CASE(LB, 8)
    HMK_CASES(12) HMK_CASES(16) HMK_CASES(20) HMK_CASES(32)
#undef HMK_CASES
#undef HMK_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_neighbors_direct(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles,
                                   const int32_t *d_matrix, int max_shift, int shift_penalty, int threshold,
                                   hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    const size_t lds = 2304 + 256 * SEQ_STRIDE_DW * 4 + 16 * 8 * 4 + 4 * 128 * 3 * 4;
    hipLaunchKernelGGL(k_neighbors_direct, dim3(n_tiles), dim3(256), lds, s, P, tile_base, d_matrix, max_shift,
                       shift_penalty, threshold);
    return hipGetLastError();
}

hipError_t launch_neighbors_local(int lbmax, bool enc, const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles,
                                  const int32_t *d_matrix, int gap_open, int gap_extend, int threshold, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
#define HMK_NL(LB, E) hipLaunchKernelGGL((k_neighbors_local<LB, E>), dim3(n_tiles), dim3(256), 0, s, P, tile_base, d_matrix, \
                                         gap_open, gap_extend, threshold)
#define HMK_NLP(LB) hipLaunchKernelGGL((k_neighbors_local_pk<LB>), dim3(n_tiles), dim3(256), 0, s, P, tile_base, d_matrix, \
                                       gap_open, gap_extend, threshold)
    const bool packed = enc && getenv("HMK_LOCAL_NO_PK") == nullptr;   // two column sequences per lane
    if (lbmax <= 12) { if (packed) HMK_NLP(12); else if (enc) HMK_NL(12, true); else HMK_NL(12, false); }
    else if (lbmax <= 20) { if (packed) HMK_NLP(20); else if (enc) HMK_NL(20, true); else HMK_NL(20, false); }
    else { if (packed) HMK_NLP(32); else if (enc) HMK_NL(32, true); else HMK_NL(32, false); }
#undef HMK_NLP
#undef HMK_NL
    return hipGetLastError();
}

hipError_t launch_csr_degree_scan(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t n,
                                  bool symmetric, uint32_t *deg, uint64_t *start, uint64_t *tile_scratch, int *score_range,
                                  hipStream_t s) {
    const int init[3] = {INT_MAX, INT_MIN, 0};
    hipError_t e = hipMemcpyAsync(score_range, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_edge_degree, dim3(512, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, deg,
                       symmetric ? 1 : 0, score_range, n);
    launch_scan<uint64_t>(deg, start, n, tile_scratch, nullptr, s);
    return hipGetLastError();
}

hipError_t launch_csr_scatter(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts,
                              bool symmetric, const uint64_t *start, uint32_t *cursor, void *adj, bool packed, int base,
                              hipStream_t s) {
    if (packed)
        hipLaunchKernelGGL((k_edge_scatter<NbrPacked>), dim3(512, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard,
                           counts, start, cursor, (NbrPacked *)adj, symmetric ? 1 : 0, base);
    else
        hipLaunchKernelGGL((k_edge_scatter<Nbr>), dim3(512, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts,
                           start, cursor, (Nbr *)adj, symmetric ? 1 : 0, base);
    return hipGetLastError();
}

hipError_t launch_compact_edges(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts,
                                uint64_t *out, uint64_t out_capacity, unsigned long long *total, hipStream_t s) {
    hipLaunchKernelGGL(k_compact_edges, dim3(128, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, out,
                       out_capacity, total);
    return hipGetLastError();
}

hipError_t launch_pack_rows(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t n,
                            int threshold, uint32_t *scratch, uint32_t *row_start, uint32_t *adj, uint64_t adj_capacity,
                            hipStream_t s) {
    // scratch: uint32 deg[n], cursor[n], misfit, pad to 8 bytes, then the scan's uint64 tile sums
    hipError_t e = hipMemsetAsync(scratch, 0, ((size_t)2 * n + 1) * 4, s);
    if (e != hipSuccess) return e;
    uint32_t *deg = scratch, *cursor = scratch + n, *misfit = scratch + 2 * (size_t)n;
    uint64_t *tile_scratch = (uint64_t *)(scratch + 2 * (size_t)n + 2);
    hipLaunchKernelGGL(k_rows_degree, dim3(128, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, deg, misfit,
                       threshold);
    launch_scan<uint32_t>(deg, row_start, n, tile_scratch, misfit, s);
    hipLaunchKernelGGL(k_rows_scatter, dim3(128, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, row_start,
                       cursor, adj, adj_capacity, threshold);
    return hipGetLastError();
}

size_t pack_rows_scratch_bytes(uint32_t n) { return ((size_t)2 * n + 2) * 4 + scan_scratch_bytes(n); }
size_t scan_scratch_bytes(uint32_t n) { return ((size_t)(n + SCAN_TILE - 1) / SCAN_TILE + 1) * 8; }

hipError_t launch_unpack_rows(const uint32_t *row_start, const uint32_t *adj, uint32_t n, int threshold, uint64_t *out,
                              uint64_t out_capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_rows_unpack, dim3(1024), dim3(256), 0, s, row_start, adj, n, threshold, out, out_capacity);
    return hipGetLastError();
}

hipError_t launch_local_block(int lbmax, bool enc, const uint8_t *res32, const uint8_t *len, const int32_t *d_matrix,
                              uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int gap_open, int gap_extend, int32_t *out,
                              hipStream_t s) {
    if (r1 <= r0 || c1 <= c0) return hipSuccess;
    const dim3 grid((c1 - c0 + 255) / 256, (r1 - r0 + 15) / 16);
#define HMK_LB(LB, E) hipLaunchKernelGGL((k_local_block<LB, E>), grid, dim3(256), 0, s, res32, len, d_matrix, r0, r1, c0, c1, \
                                         gap_open, gap_extend, out)
    if (lbmax <= 20) { if (enc) HMK_LB(20, true); else HMK_LB(20, false); }
    else { if (enc) HMK_LB(32, true); else HMK_LB(32, false); }
#undef HMK_LB
    return hipGetLastError();
}

hipError_t launch_pairs(int scorer, const uint8_t *res32, const uint8_t *len, const int32_t *d_matrix,
                        const uint32_t *pi, const uint32_t *pj, uint64_t n_pairs, uint32_t r0, uint32_t c0,
                        uint32_t width, int a, int b, int32_t *out, int32_t *out_shift, hipStream_t s) {
    if (n_pairs == 0) return hipSuccess;
    uint64_t blocks = (n_pairs + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const size_t lds_shift = 2304 + 256 * 2 * SEQ_STRIDE_DW * 4;
    const size_t lds_local = lds_shift + 33 * 256 * 4;
    if (scorer == 0)
        hipLaunchKernelGGL(k_pairs<0>, dim3((uint32_t)blocks), dim3(256), lds_shift, s, res32, len, d_matrix, pi, pj,
                           n_pairs, r0, c0, width, a, b, out, out_shift);
    else
        hipLaunchKernelGGL(k_pairs<1>, dim3((uint32_t)blocks), dim3(256), lds_local, s, res32, len, d_matrix, pi, pj,
                           n_pairs, r0, c0, width, a, b, out, out_shift);
    return hipGetLastError();
}

}  // namespace hmk
