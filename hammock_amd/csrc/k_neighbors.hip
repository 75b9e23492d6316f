// k_neighbors.hip -- all-vs-all ShiftedScorer (ShiftedScorer.java:48-95) with threshold -> edge list.
//
// Integer scoring only: no MFMA, no dense contraction.  The bound that matters is the LDS lookup rate
// (ds_read_b64: 32 lanes/clk/CU) and VALU issue, see DESIGN.md "Kernels".
//
//   k_neighbors_swar    the hot kernel (every sequence of length 12, max shift 3): one (row, column) pair per lane
//                       per step; the row peptide is wave-uniform and expanded once per tile into per-position
//                       lookup tables in LDS whose entries hold ALL shifts of one column position as packed 8-bit
//                       lanes, so one ds_read_b64 + one v_add3 per dword advance all 7 shift sums of a pair
//   k_neighbors_planes  the same scheme for arbitrary lengths (conflict-free plane layout of the tables)
//   k_neighbors_direct  literal tier for length classes whose sums fit neither 8- nor 16-bit lanes
#include "hmk_device.h"

namespace hmk {

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
// DEG: also count the CSR degrees while writing edges (NeighborParams::deg, hmk_greedy_cluster); the plain neighbour pass
// is its own instantiation so that it keeps its spill-free 72-VGPR allocation.
template <int NW, int R, int CPL, int LBMAX, bool EXACT, int DEG>   // DEG: EDGES_PLAIN / EDGES_COUNT (hmk_device.h)
// The production tiling (R = 6, CPL = 2) is held to 72 VGPRs = 7 waves/SIMD, which its 18.7 KB of LDS allow as well
// (80 VGPRs would mean 6 waves): +2.4 % measured.
__global__ void __launch_bounds__(256, (R >= 5 && R <= 7 && CPL == 2) ? 7 : 1)
k_neighbors_swar(const NeighborParams P, const uint32_t tile_base) {
    constexpr int ES = NW * 4;                 // table entry bytes
    constexpr int ROWBYTES = LBMAX * 24 * ES;  // one row's tables
    constexpr int TAB_BYTES = R * ROWBYTES;
    constexpr int STAGE_CAP = 256;             // records per wave; flushed when fewer than 64 slots are free
    constexpr int REC_DW = 1;                  // compact records, see flush_stage_compact
    constexpr int LPADW = (LBMAX <= 16) ? 4 : 8;  // residue dwords a lane needs (rows are P.lpad bytes apart)
    static_assert(TAB_BYTES <= 65536, "row tables must stay addressable by the DS immediate offset");
    static_assert(EXACT && NW == 2 && LBMAX == 12 && R <= 16, "this kernel is the length 12, max shift 3, 8-bit lane case");
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
    const int threshold = 128 - Cp->g;         // 8-bit lanes: lane value = 128 - threshold + score
    const uint32_t himask = 0x80808080u;
    const uint32_t shard = tile_shard(tile_base + blockIdx.x);

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + wave * (STAGE_CAP * REC_DW);   // 32-bit LDS pointer: no 64-bit flat pointer held (and spilled) across the tile

    build_begin();
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
    build_end();

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
                if constexpr (LPADW == 8) {
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
                read_phase_begin(true);
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
                read_phase_end(true);
#pragma unroll
                for (int p = 0; p < CPL; p++) {
                    uint32_t any = W[p][0];
#pragma unroll
                    for (int w = 1; w < NW; w++) any |= W[p][w];
                    const bool hit = (any & himask) != 0;  // some shift reached score >= threshold
                    if (__ballot(hit) != 0) {              // wave-uniform, rare
                        if (cnt > (uint32_t)(STAGE_CAP - 64)) {  // keep room for one wave of hits
                            flush_stage_compact<DEG>(stage, cnt, P, T, threshold, shard);
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
                        if (keep)   // score - threshold = best lane - 128, in 0..127
                            stage[cnt + mbcnt64(mask)] = (col - T.col0) | ((uint32_t)r << 16) | ((max_byte_lane(W[p][0], W[p][1]) - 128u) << 20);
                        cnt += (uint32_t)__popcll(mask);
                    }
                }
            }
        }
    }
    flush_stage_compact<DEG>(stage, cnt, P, T, threshold, shard);
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
//           mb 576 B, rowres R x 32 B, stage 4 waves x 192 dwords of hit records (one dword each with 8-bit lanes)
// Plane strides.  The compiler fuses two ds_reads off the same address register into ds_read2[st64]
// when their immediates differ by < 2048 B or by a multiple of 512 B (256 B for 4-byte reads); a fused
// read whose halves are a multiple of 256 B apart hits the same banks with both halves (measured: 43 %
// conflict cycles with strides of 1536 / 3072 B).  Strides of 8 x odd bytes, >= 2048, rule the fusion out
// (static_assert in the kernel).
constexpr int plane64_bytes(int lbmax) { return lbmax * 24 * 8 + 8; }
// Odd NW: the last dword of TWO rows shares one 8-byte entry of a "pair plane", so one ds_read_b64 serves both rows
// (a 4-byte plane per row would cost the same 2 LDS cycles per read as an 8-byte one, MI355X_MICROARCH.md LDS table).
// Every plane -- NW / 2 per row, then (R + 1) / 2 pair planes -- is plane64_bytes() long, i.e. consecutive planes are
// 8 x odd >= 2048 bytes apart and no two of a workgroup's (< 64) planes a multiple of 512 B apart: the compiler cannot
// fuse two table reads off one address register into a same-bank ds_read2[st64] (see "Plane strides" above).
constexpr int planes_per_tile(int r, int nw) { return r * (nw / 2) + ((r + 1) / 2) * (nw & 1); }
constexpr int rows_for(int rowbytes, int nw) {  // rows per tile: table bytes and R x NW accumulator registers, tuned on config 4a
    constexpr int TAB_BUDGET = 24576, ACC_CAP = 16;
    int r = TAB_BUDGET / rowbytes;
    if (r > ACC_CAP / nw) r = ACC_CAP / nw;
    r = r > 16 ? 16 : (r < 1 ? 1 : r);
    // odd NW reads the last dword of two rows at once (pair planes): an odd row count wastes half of one such read per
    // position, measured 0.74 against 0.83 of the LDS ideal for the instantiations with R = 5 / 3: one row less
    if ((nw & 1) && r > 1 && (r & 1)) r -= 1;
    return r;
}
constexpr int planes_rowbytes(int lbmax, int nw) { return (2 * (nw / 2) + (nw & 1)) * plane64_bytes(lbmax) / 2; }   // table bytes per row

template <int NW, int R, int LBMAX>
// the narrow-entry instantiations are held to 80 VGPRs = 6 waves/SIMD, which is what their 25 KB of LDS allow per CU
// (<2, 8, 12> took 85 VGPRs = 5 waves and showed the lowest LDS busy fraction of config 4a; <2, 6, 20> -- 12 accumulators and 20
// offsets -- does not fit 80 and is left alone)
__global__ void __launch_bounds__(256, (NW == 2 && 2 * R + LBMAX <= 30) ? 6 : 1) k_neighbors_planes(const NeighborParams P, const uint32_t tile_base) {
    constexpr int NP = NW / 2, H = NW & 1;
    constexpr int CPL = NW <= 2 ? 2 : 1;
    constexpr int PLANE64 = plane64_bytes(LBMAX);
    constexpr int NPLANES = planes_per_tile(R, NW);
    static_assert(R == rows_for(planes_rowbytes(LBMAX, NW), NW) && NPLANES < 64 && PLANE64 >= 2048 && (PLANE64 / 8) % 2 == 1,
                  "plane stride must keep table reads from being fused into a same-bank ds_read2");
    constexpr int TAB_BYTES = NPLANES * PLANE64;   // plane (r, q) = r * NP + q; pair plane rp = R * NP + rp
    constexpr int STAGE_DW = 192;                      // dwords per wave: 192 one-dword hit records (8-bit lanes) or 96 two-dword ones
    constexpr int LPADW = (LBMAX <= 16) ? 4 : 8;
    // table build scratch: per (row, residue) the row's cells as one byte string (see below); it shares the
    // stage area, which is idle while the tables are built
    // dwords per string, one spare for the funnel shift; an ODD stride, so that the 64 lanes of pass 2 (consecutive
    // residues = consecutive strings) spread over all banks: with 8 dwords per string 3.7-6 % of the LDS cycles of the
    // <4,4,12>, <3,4,16> and <2,5,20> instantiations were bank-conflict cycles (PMC), with 7 or 9 dwords 0.2-0.4 %
    constexpr int WIN_DW = ((LBMAX - 1 + 4 * NW + 3) / 4 + 1) | 1;
    constexpr int STAGE_BYTES = 4 * STAGE_DW * 4, WIN_BYTES = R * 24 * WIN_DW * 4;
    constexpr int AUX_BYTES = STAGE_BYTES > WIN_BYTES ? STAGE_BYTES : WIN_BYTES;
    constexpr int TAB_PAD = (TAB_BYTES + 15) & ~15;
    constexpr int LDS_BYTES = TAB_PAD + 576 + R * 32 + AUX_BYTES;
    static_assert(TAB_BYTES <= 65536 && LDS_BYTES <= 65536, "LDS budget / DS immediate range");
    __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];
    uint8_t *tab = smem;
    uint8_t *mb = smem + TAB_PAD;
    uint8_t *rowres = mb + 576;
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(rowres + R * 32);

    const Tile T = P.tiles[tile_base + blockIdx.x];
    const TileClass *Cp = P.classes + T.cls;
    const int la = Cp->la, lb = Cp->lb, nd = Cp->nd, X = Cp->x;
    const bool case_b = Cp->case_b != 0;
    const bool lane16 = Cp->path == PATH_U16;
    const int g = Cp->g;
    const uint32_t himask = lane16 ? 0x80008000u : 0x80808080u;
    const uint32_t shard = tile_shard(tile_base + blockIdx.x);
    const int tid = threadIdx.x;
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + (tid >> 6) * STAGE_DW;   // 32-bit LDS pointer: no 64-bit flat pointer held (and spilled) across the tile
    // hit records (hmk_device.h flush_stage_packed): one dword when score - threshold fits 12 bits -- always with 8-bit lanes,
    // where it is lane - 128 -- else two
    const uint32_t rec_dw = lane16 ? 2u : 1u;
    const bool prio = true;   // wave priority for the read phase (hmk_device.h)
    const int base_score = (lane16 ? 32768 : 128) - g;   // the score of a lane that just reaches the threshold

    build_begin();
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
    // Entry (r, j, c) holds, lane by lane (t = s + X), the cells of ONE byte string per (r, c):
    //   S:  win[k] = cell(i = k - X),            entry j = win[j .. j + nd)
    //   L:  win[k] = cell(i = lb - 1 + X - k),   entry j = win[lb - 1 - j .. lb - 1 - j + nd)
    // (cell = 0 outside the row).  Pass 1 gathers the strings into LDS, four cells per work item; pass 2 cuts
    // every entry out of its string with funnel shifts (v_alignbyte) -- a handful of instructions per dword
    // instead of one bounds-checked double gather per lane.
    {
        uint32_t *win = stage_all;   // R * 24 strings of WIN_DW dwords (aliases the stage: idle during the build)
        const int l0 = lb - 1 + X;
        for (int e = tid; e < R * 24 * WIN_DW; e += 256) {
            const int rc = e / WIN_DW, kd = e - rc * WIN_DW;
            const int r = rc / 24, c = rc - r * 24;
            uint32_t acc = 0;
            if ((uint32_t)r < T.nrows) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k = kd * 4 + q;
                    const int i = case_b ? k - X : l0 - k;
                    if (i >= 0 && i < la) {
                        const int a = rowres[r * 32 + i];
                        acc |= (uint32_t)(case_b ? mb[c * 24 + a] : mb[a * 24 + c]) << (q * 8);
                    }
                }
            }
            win[e] = acc;
        }
        __syncthreads();
        const int per_row = lb * 24;
        // lanes at and beyond nd stay zero
        uint32_t lane_mask[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int left = nd - w * (lane16 ? 2 : 4);   // live lanes in dword w
            if (lane16) lane_mask[w] = left >= 2 ? 0xFFFFFFFFu : left == 1 ? 0x0000FFFFu : 0u;
            else lane_mask[w] = left >= 4 ? 0xFFFFFFFFu : left <= 0 ? 0u : (1u << (left * 8)) - 1u;
        }
        for (int e = tid; e < R * per_row; e += 256) {
            const int r = e / per_row;
            const int rem = e - r * per_row;
            const int j = rem / 24;
            const int c = rem - j * 24;
            const uint32_t *str = win + (r * 24 + c) * WIN_DW;
            const int s0 = case_b ? j : lb - 1 - j;       // first byte of the entry in its string
            uint32_t dw[NW];
            if (!lane16) {
                uint32_t d[NW + 1];
#pragma unroll
                for (int w = 0; w <= NW; w++) d[w] = str[(s0 >> 2) + w];
#pragma unroll
                for (int w = 0; w < NW; w++) dw[w] = __builtin_amdgcn_alignbyte(d[w + 1], d[w], (uint32_t)(s0 & 3)) & lane_mask[w];
            } else {
#pragma unroll
                for (int w = 0; w < NW; w++) {      // two byte cells widened to two 16-bit lanes
                    const int o = s0 + 2 * w;
                    const uint32_t t2 = __builtin_amdgcn_alignbyte(str[(o >> 2) + 1], str[o >> 2], (uint32_t)(o & 3));
                    dw[w] = ((t2 & 0xFFu) | ((t2 << 8) & 0x00FF0000u)) & lane_mask[w];
                }
            }
            uint8_t *row = tab + (r * NP) * PLANE64;
#pragma unroll
            for (int q = 0; q < NP; q++)
                *reinterpret_cast<u32x2 *>(row + q * PLANE64 + (j * 24 + c) * 8) = u32x2{dw[2 * q], dw[2 * q + 1]};
            if (H) *reinterpret_cast<uint32_t *>(tab + (R * NP + (r >> 1)) * PLANE64 + (j * 24 + c) * 8 + (r & 1) * 4) = dw[NW - 1];
        }
    }
    __syncthreads();
    build_end();

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
                if constexpr (LPADW == 8) {
                    const u32x4 v1 = src[1];
                    words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
                }
            }
            uint32_t off64[LBMAX];   // every plane has 8-byte entries: one offset per position serves them all
#pragma unroll
            for (int j = 0; j < LBMAX; j++) {
                const uint32_t c = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                off64[j] = tab_addr + (uint32_t)(j * 24 * 8) + c * 8;
            }

            // ---- position-major accumulation: the `j < lb` tests are wave-uniform branches; inside one
            // branch the reads of all R rows for two positions are in flight together, and a pair of
            // positions costs one v_add3 per accumulator dword.  Rows past T.nrows read zero tables.
            uint32_t W[R][NW];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int w = 0; w < NW; w++) W[r][w] = cinit[w];
            // all R rows' entries of one position: NP reads per row + one read per row PAIR for the odd dword
            auto read_position = [&](int j, uint32_t (&e)[R][NW]) {
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        const u32x2 v = lds_read<u32x2>(off64[j] + (uint32_t)((r * NP + q) * PLANE64));
                        e[r][2 * q] = v.x; e[r][2 * q + 1] = v.y;
                    }
                if (H) {
#pragma unroll
                    for (int rp = 0; rp < (R + 1) / 2; rp++) {
                        const u32x2 v = lds_read<u32x2>(off64[j] + (uint32_t)((R * NP + rp) * PLANE64));
                        e[2 * rp][NW - 1] = v.x;
                        if (2 * rp + 1 < R) e[2 * rp + 1 < R ? 2 * rp + 1 : 0][NW - 1] = v.y;
                    }
                }
            };
            // Positions are taken in pairs (one v_add3 per accumulator dword and pair).  An odd column length puts its
            // single position FIRST, so that each variant is a chain of nested wave-uniform `if`s around in-place adds --
            // an `if / else if` per pair made the compiler copy all R x NW accumulators on every skipped pair.
            // a fresh scalar copy of lb per batch: the length tests below stay s_cmp + s_cbranch_scc.  Hoisted out of the
            // batch loop they become 64-bit lane masks, more of them than there are SGPRs, and each test then starts with
            // two v_readlane of a spilled mask (14 % of the VALU instructions of the accumulation, <3,4,12> ISA)
            int lbs = lb;
            asm volatile("" : "+s"(lbs));
            auto add_pairs = [&](auto start_tag) {
                constexpr int J0 = decltype(start_tag)::value;
#pragma unroll
                for (int j = J0; j + 1 < LBMAX; j += 2) {
                    if (j + 1 >= lbs) break;                    // wave-uniform: nothing beyond this position
                    uint32_t e0[R][NW], e1[R][NW];
                    read_position(j, e0);
                    read_position(j + 1, e1);
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int w = 0; w < NW; w++) W[r][w] = W[r][w] + e0[r][w] + e1[r][w];
                }
            };
            read_phase_begin(prio);
            if (lbs & 1) {
                uint32_t e0[R][NW];
                read_position(0, e0);
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int w = 0; w < NW; w++) W[r][w] += e0[r][w];
                add_pairs(std::integral_constant<int, 1>{});
            } else {
                add_pairs(std::integral_constant<int, 0>{});
            }

            read_phase_end(prio);
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
                    if ((cnt + 64) * rec_dw > (uint32_t)STAGE_DW) {
                        flush_stage_packed<true>(stage, cnt, P, T, base_score, rec_dw, shard);
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
                        HMK_LDS uint32_t *rec = stage + (cnt + mbcnt64(mask)) * rec_dw;
                        const uint32_t where = (col - T.col0) | ((uint32_t)r << 16);
                        if (lane16) { rec[0] = where; rec[1] = (uint32_t)((int)mx - g); }
                        else rec[0] = where | ((mx - 128u) << 20);
                    }
                    cnt += (uint32_t)__popcll(mask);
                }
            }
        }
    }
    flush_stage_packed<true>(stage, cnt, P, T, base_score, rec_dw, shard);
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
    const uint32_t shard = tile_shard(tile_base + blockIdx.x);
    const int tid = threadIdx.x;
    HMK_LDS uint32_t *stage = (HMK_LDS uint32_t *)stage_all + (tid >> 6) * (STAGE_CAP * REC_DW);   // 32-bit LDS pointer: no 64-bit flat pointer held (and spilled) across the tile

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

// -----------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------
template <int NW, int R, int CPL, int LBMAX, bool EXACT>
static hipError_t launch_swar_t(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, hipStream_t s) {
    if (P.deg)
        hipLaunchKernelGGL((k_neighbors_swar<NW, R, CPL, LBMAX, EXACT, EDGES_COUNT>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
    else
        hipLaunchKernelGGL((k_neighbors_swar<NW, R, CPL, LBMAX, EXACT, EDGES_PLAIN>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
    return hipGetLastError();
}

// The exact length-12, NW = 2 kernel's tiling: 6 rows per tile, 2 columns per lane (of the ten tilings rounds 1-2 timed).
constexpr int SWAR12_ROWS = 6, SWAR12_CPL = 2;

// Generic instantiations: column-length capacity LBMAX x dwords per entry NW.  Rows per tile
// R = what fits a 40 KB table budget (<= 16); 2 columns per lane for the narrow entries.
constexpr int swar_r(int lbmax, int nw) {
    return rows_for(planes_rowbytes(lbmax, nw), nw);
}

int swar_lbmax_for(int lb) { return lb <= 12 ? 12 : lb <= 16 ? 16 : lb <= 20 ? 20 : 32; }  // plane strides need >= 11 positions

int swar_rows_per_tile(int lbmax, int nw, bool exact) {
    if (exact && lbmax == 12 && nw == 2) return SWAR12_ROWS;
    return swar_r(lbmax, nw);
}

hipError_t launch_neighbors_swar(int lbmax, int nw, bool exact, const NeighborParams &P,
                                 uint32_t tile_base, uint32_t n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    if (exact && lbmax == 12 && nw == 2) return launch_swar_t<2, SWAR12_ROWS, SWAR12_CPL, 12, true>(P, tile_base, n_tiles, s);
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

// loads this translation unit's code object (HIP defers that to the first launch: 5-10 ms of the first call otherwise)
hipError_t warm_neighbors_module() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_neighbors_direct));
}

}  // namespace hmk
