// hmk_plan.cpp -- the planner of the neighbour passes: length buckets, score bounds and lane classes (classify), kernel
// selection per class, tiles and launch groups (build_plan); the LocalAlignmentScorer pass's plan (build_plan_local).
#include "hmk_ctx.h"

namespace hmk { namespace impl {

// std::stable_sort's result on several threads: contiguous runs sorted on their own, then merged pairwise (std::merge takes
// from the left run on ties)
template <class T, class Cmp>
void parallel_stable_sort(std::vector<T> &v, Cmp before) {
    const size_t n = v.size();
    const unsigned hw = usable_cpus();
    size_t runs = 1;
    while (runs < 8 && runs < (hw ? hw : 1u) && n / (2 * runs) >= 32768) runs *= 2;
    if (runs == 1) { std::stable_sort(v.begin(), v.end(), before); return; }
    std::vector<size_t> cut(runs + 1);
    for (size_t r = 0; r <= runs; r++) cut[r] = n * r / runs;
    {
        std::vector<std::thread> pool;
        for (size_t r = 1; r < runs; r++) pool.emplace_back([&, r] { std::stable_sort(v.begin() + (long)cut[r], v.begin() + (long)cut[r + 1], before); });
        std::stable_sort(v.begin(), v.begin() + (long)cut[1], before);
        for (std::thread &th : pool) th.join();
    }
    std::vector<T> other(n);
    std::vector<T> *from = &v, *to = &other;
    for (size_t width = 1; width < runs; width *= 2) {
        std::vector<std::thread> pool;
        for (size_t r = 0; r < runs; r += 2 * width) {
            auto job = [&, r] {
                std::merge(from->begin() + (long)cut[r], from->begin() + (long)cut[r + width], from->begin() + (long)cut[r + width],
                           from->begin() + (long)cut[r + 2 * width], to->begin() + (long)cut[r], before);
            };
            if (r + 2 * width < runs) pool.emplace_back(job); else job();
        }
        for (std::thread &th : pool) th.join();
        std::swap(from, to);
    }
    if (from != &v) v.swap(other);
}

void free_plan(Plan &pl) {
    if (pl.d_res_sorted) (void)hipFree(pl.d_res_sorted);
    if (pl.d_perm) (void)hipFree(pl.d_perm);
    if (pl.d_mb) (void)hipFree(pl.d_mb);
    if (pl.d_classes) (void)hipFree(pl.d_classes);
    if (pl.d_tiles) (void)hipFree(pl.d_tiles);
    pl = Plan();
}

// Lane layout of one (row length, column length) class; see DESIGN.md "SWAR tables".
// row_bound < 0: lanes are proven to fit for ANY pair of the class (every cell at the matrix maximum).
// row_bound >= 0: the caller guarantees score(row, anything) <= row_bound for the rows it will put into
// this class (sum of the row residues' best cells), which lets long peptides keep 8-bit lanes.
// *u8_row_limit receives the largest row_bound for which 8-bit lanes fit (or -1 if they never do).
void classify(const hmk_ctx *ctx, int la, int lb, int X, int p, int thr, TileClass *out, long long row_bound,
              long long *u8_row_limit) {
    TileClass c{};
    const int m = std::min(la, lb), nl = std::max(la, lb);
    const int d = nl - m;
    const int nd = 2 * X + d + 1;
    c.la = (uint8_t)la;
    c.lb = (uint8_t)lb;
    c.nd = (uint8_t)std::min(nd, 255);
    c.case_b = lb < la;
    c.x = (uint8_t)X;
    c.d = d;
    const int bias = ctx->min_m < 0 ? -ctx->min_m : 0;
    const long long cell_max = (long long)ctx->max_m + bias;
    c.path = PATH_DIRECT;
    for (int attempt = 0; attempt < 2 && c.path == PATH_DIRECT; attempt++) {
        const bool u16 = attempt == 1;
        const long long lane_max = u16 ? 65535 : 255;
        const long long g = (u16 ? 32768LL : 128LL) - thr;
        const int max_nd = u16 ? 16 : 32;
        if (nd > max_nd || cell_max > 255) continue;
        bool ok = true, lower_ok = true;
        long long ci[32], limit = 1LL << 40;
        for (int t = 0; t < nd; t++) {
            const int s = t - X;
            const long long ncell = s <= 0 ? m + s : std::min(m, nl - s);
            long long pen = (long long)d * p;                       // ShiftedScorer.java:79
            if (s < 0) pen += (long long)(-s) * 2 * p;              // :80-82
            if (s > d) pen += (long long)(s - d) * 2 * p;           // :83-85
            const long long c0 = g + pen - bias * ncell;            // lane value = g + pen + sum of the cells
            if (c0 < 0) lower_ok = false;
            const long long top = row_bound >= 0 ? g + pen + row_bound : c0 + ncell * cell_max;
            if (top > lane_max) ok = false;
            limit = std::min(limit, lane_max - g - pen);
            ci[t] = c0;
        }
        if (!u16 && u8_row_limit) *u8_row_limit = lower_ok ? limit : -1;
        if (!ok || !lower_ok) continue;
        c.path = u16 ? PATH_U16 : PATH_U8;
        c.g = (int32_t)g;
        const int lpd = u16 ? 2 : 4, bits = u16 ? 16 : 8;
        const int ndw = (nd + lpd - 1) / lpd;
        c.nw = (uint8_t)ndw;  // 1..8 dwords per table entry, each count has its own kernel
        for (int t = 0; t < nd; t++) c.cinit[t / lpd] |= (uint32_t)ci[t] << ((t % lpd) * bits);
    }
    *out = c;
}

// band_rows: tiles that touch a sequence with caller index < band_rows are put first in every launch group, so that a
// first launch of only those tiles completes the adjacency rows phase 1 of the greedy merge reads first
// (hmk_greedy_cluster); -1 = the caller does not care (any cached plan with the other parameters will do).
int build_plan(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, int64_t band_rows) {
    Plan &pl = ctx->plan;
    if (pl.valid && pl.X == X && pl.p == p && pl.thr == thr && pl.part == part && pl.n_parts == n_parts &&
        (band_rows < 0 || pl.band_req == band_rows) && pl.no_rows_kernel == ctx->sw.no_rows_kernel)
        return HMK_OK;
    free_plan(pl);
    if (band_rows < 0) band_rows = 0;
    const int64_t band_req = band_rows;
    const bool plan_timing = ctx->sw.greedy_timing;
    const auto plan_t0 = std::chrono::steady_clock::now();
    auto plan_lap = [&](const char *what) {
        if (plan_timing)
            fprintf(stderr, "[hmk plan] %s at %.2f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - plan_t0).count());
    };
    const uint32_t n = ctx->n;
    if (n == 0) return fail(ctx, HMK_ERR_NO_SEQUENCES, "no sequences set (hmk_set_sequences)");
    if (X < 0) return fail(ctx, HMK_ERR_BAD_ARG, "max_shift must be >= 0");
    if (n_parts == 0 || part >= n_parts) return fail(ctx, HMK_ERR_BAD_ARG, "part must be < n_parts");
    if (X >= ctx->min_len)
        return fail(ctx, HMK_ERR_SHIFT_TOO_BIG,
                    "Shift too big: " + std::to_string(ctx->min_len - 1) + " is maximum, but " + std::to_string(X) +
                        " found");  // ShiftedScorer.java:59-62
    if (thr < -30000 || thr > 30000) return fail(ctx, HMK_ERR_BAD_ARG, "threshold outside [-30000, 30000]");
    {   // edge scores travel as int16: the largest score any pair can reach must fit
        const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                              (long long)std::max(0, p) * ((ctx->max_len - ctx->min_len) + 2LL * X);
        if (top > 32767)
            return fail(ctx, HMK_ERR_BAD_ARG, "scores up to " + std::to_string(top) + " are possible with this matrix / shift penalty: "
                                               "they do not fit the int16 score of a packed edge");
    }

    // ---- bucket by length ("sorted order") --------------------------------------
    uint32_t bucket[HMK_MAX_LEN + 2] = {0};
    for (uint32_t k = 0; k < n; k++) bucket[ctx->len[k] + 1]++;
    for (int l = 0; l <= HMK_MAX_LEN; l++) bucket[l + 1] += bucket[l];
    std::vector<uint32_t> perm(n);
    {
        uint32_t fill[HMK_MAX_LEN + 2];
        std::memcpy(fill, bucket, sizeof(fill));
        for (uint32_t k = 0; k < n; k++) perm[fill[ctx->len[k]]++] = k;
    }
    // Per-sequence score bound: no pair involving sequence k scores above bound[k] = sum over its residues
    // of the best (non-negative) cell of that residue's matrix row/column.  If some class does not fit
    // 8-bit lanes for arbitrary pairs, its bucket is ordered by this bound and the rows below the class's
    // limit still run on 8-bit lanes (for BLOSUM62 a 20-mer's bound is its self-score, ~112 +- 8, against
    // a limit of 127 + threshold).
    bool refine = false;
    for (int la = 1; la <= HMK_MAX_LEN && !refine; la++)
        for (int lb = 1; lb <= HMK_MAX_LEN && !refine; lb++) {
            if (bucket[la] == bucket[la + 1] || bucket[lb] == bucket[lb + 1]) continue;
            if (ctx->symmetric && lb > la) continue;
            TileClass tc;
            long long limit = -1;
            classify(ctx, la, lb, X, p, thr, &tc, -1, &limit);
            if (tc.path != PATH_U8 && limit >= 0) refine = true;
        }
    std::vector<uint32_t> bound_sorted;  // bound of the sequence at each sorted position (refine only)
    constexpr uint32_t BCAP = 4095;      // bounds are only compared with limits < 65536; clamped for the counting sort
    // refine, but EVERY row of every class that needs its bound has one within the class's limit (uniform 15- or 20-mers at the
    // reference's default threshold: a 20-mer's bound is ~112 +- 8 against a limit of 161): the buckets keep the caller's order
    // -- no reordering, so the band of a clustering call survives and a one-length set keeps its compile-time-length kernel
    bool all_rows_fit = false;
    if (refine) {
        long long best[HMK_ALPHABET];
        for (int a = 0; a < HMK_ALPHABET; a++) {
            long long b = 0;
            for (int y = 0; y < HMK_ALPHABET; y++)
                b = std::max<long long>(b, std::max(ctx->M[a * HMK_ALPHABET + y], ctx->M[y * HMK_ALPHABET + a]));
            best[a] = b;
        }
        std::vector<uint32_t> bound(n);
        for (uint32_t k = 0; k < n; k++) {
            long long b = 0;
            for (uint32_t q = ctx->off[k]; q < ctx->off[k + 1]; q++) b += best[ctx->res[q]];
            bound[k] = (uint32_t)std::min<long long>(b, BCAP);
        }
        {
            uint32_t bucket_max[HMK_MAX_LEN + 2] = {0};
            for (uint32_t k = 0; k < n; k++) bucket_max[ctx->len[k]] = std::max(bucket_max[ctx->len[k]], bound[k]);
            all_rows_fit = true;
            for (int la = 1; la <= HMK_MAX_LEN && all_rows_fit; la++)
                for (int lb = 1; lb <= HMK_MAX_LEN && all_rows_fit; lb++) {
                    if (bucket[la] == bucket[la + 1] || bucket[lb] == bucket[lb + 1]) continue;
                    if (ctx->symmetric && lb > la) continue;
                    TileClass tc;
                    long long limit = -1;
                    classify(ctx, la, lb, X, p, thr, &tc, -1, &limit);
                    if (tc.path == PATH_U8) continue;
                    if (limit < 0 || (long long)bucket_max[la] > std::min<long long>(limit, BCAP - 1)) all_rows_fit = false;
                }
        }
        // stable counting sort of every length bucket by bound
        std::vector<uint32_t> sorted(n), cnt(BCAP + 2);
        for (int l = 1; l <= HMK_MAX_LEN && !all_rows_fit; l++) {
            const uint32_t b0 = bucket[l], b1 = bucket[l + 1];
            if (b0 == b1) continue;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (uint32_t q = b0; q < b1; q++) cnt[bound[perm[q]] + 1]++;
            for (uint32_t v = 0; v <= BCAP; v++) cnt[v + 1] += cnt[v];
            for (uint32_t q = b0; q < b1; q++) sorted[b0 + cnt[bound[perm[q]]]++] = perm[q];
        }
        if (!all_rows_fit) perm.swap(sorted);
        bound_sorted.resize(n);
        for (uint32_t q = 0; q < n; q++) bound_sorted[q] = bound[perm[q]];
    }
    // band members of a length bucket are its leading sorted positions (the counting sort keeps caller order); a
    // bucket reordered by score bound has no such prefix, so the band is dropped there (phase 1 then waits for the pass)
    uint32_t band_end[HMK_MAX_LEN + 2];
    if (refine && !all_rows_fit) band_rows = 0;
    for (int l = 0; l <= HMK_MAX_LEN; l++) {
        band_end[l] = bucket[l];
        if (band_rows > 0)
            while (band_end[l] < bucket[l + 1] && perm[band_end[l]] < (uint64_t)band_rows) band_end[l]++;
    }
    pl.band_rows = (uint32_t)band_rows;
    pl.band_req = band_req;
    plan_lap("buckets and score bounds");
    pl.lbmax = swar_lbmax_for(ctx->max_len);
    pl.lpad = ctx->max_len <= 16 ? 16 : 32;
    // The exact hot kernel: every sequence has length 12, max shift 3, and the (12, 12) class fits 8-bit
    // lanes in 8-byte entries.  It reads residues pre-multiplied by the entry size (see res_sorted below).
    // Row-packed kernels (k_neighbors_rows.hip) take every 8-bit-lane class they have an instantiation for; a set of one
    // length may have one with the length at compile time.  HMK_NO_ROWS_KERNEL=1: the shift-packed kernels of round 1-2.
    const bool use_rows = !ctx->sw.no_rows_kernel;
    pl.no_rows_kernel = ctx->sw.no_rows_kernel;
    pl.exact = false;
    pl.rows_exact = false;
    if (use_rows && ctx->min_len == ctx->max_len) {
        TileClass t1;
        classify(ctx, ctx->min_len, ctx->min_len, X, p, thr, &t1);
        // (8-bit lanes for any pair of the class, or -- by their score bounds -- for every row the set has)
        pl.rows_exact = (t1.path == PATH_U8 || (refine && all_rows_fit)) && rows_kernel_available(X, ctx->min_len, ctx->min_len, true);
    }
    if (!use_rows && ctx->min_len == 12 && ctx->max_len == 12 && X == 3) {
        TileClass t12;
        classify(ctx, 12, 12, X, p, thr, &t12);
        pl.exact = t12.path == PATH_U8 && t12.nw == 2;
    }
    // Column runs: long runs amortise the table build (65,536 columns: 3.55 ms for the whole 10^5 pass against
    // 3.60 ms with 16,384), short ones keep the tail of a small launch short (a 1/8 shard: 0.478 ms with 16,384,
    // 0.532 ms with 65,536).  Take the longest run that still leaves ~8 rounds of workgroups (256 CUs x 7).
    const uint64_t tile_rows = pl.rows_exact ? (uint64_t)rows_per_tile_rows(X, 0, ctx->min_len, true) : use_rows ? 16 : 6;
    const uint64_t row_groups = (uint64_t)n / tile_rows / n_parts + 1;
    pl.cols_per_tile = 65536;
    while (pl.cols_per_tile > 16384 && row_groups * ((uint64_t)n / (2 * pl.cols_per_tile) + 1) < 8 * 1792)
        pl.cols_per_tile /= 2;
    // (not below 4,096 columns: a tile's dead time -- its chain of dependent loads before the first table read, the flush after
    // the last -- is about four 256-column batches long, and short tiles pay it several times over on every workgroup slot.
    // 10^4 12-mers: 1,024 / 2,048 / 4,096 / 16,384 columns per tile 0.090 / 0.061 / 0.053 / 0.051 ms, although the last leaves
    // a third of the slots empty; 3 x 10^4: 2,048 / 4,096 / 8,192 0.354 / 0.301 / 0.294 ms.)
    while (pl.cols_per_tile > 4096 && ((uint64_t)n / tile_rows + 1) * ((uint64_t)n / (2 * pl.cols_per_tile) + 1) < 4096)
        pl.cols_per_tile /= 2;

    // ---- classes and tiles --------------------------------------------------------
    std::vector<TileClass> classes;
    std::map<int, int> class_of;  // la * 64 + lb
    std::map<std::tuple<int, int, int>, std::vector<Tile>> grouped;  // (path, nw, column capacity)
    const uint32_t COLS = pl.cols_per_tile;
    const bool equal_runs = true;
    hmk_neighbor_stats &S = pl.stats;
    S = hmk_neighbor_stats{};
    S.symmetric = ctx->symmetric;
    uint64_t row_chunk_counter = 0;
    for (int la = 1; la <= HMK_MAX_LEN; la++) {
        const uint32_t rb = bucket[la], re = bucket[la + 1];
        if (rb == re) continue;
        for (int lb = 1; lb <= HMK_MAX_LEN; lb++) {
            const uint32_t cb = bucket[lb], ce = bucket[lb + 1];
            if (cb == ce) continue;
            // unordered pairs: the LONGER bucket supplies the rows, so a pair costs one table lookup per
            // residue of its SHORTER sequence (the column), ShiftedScorer.java:51-57 decides S/L by length anyway
            if (ctx->symmetric && lb > la) continue;
            const bool same = la == lb;
            if (same && re - rb < 2) continue;
            // row ranges of this (la, lb) pair: all rows in one class, or -- when 8-bit lanes do not fit every
            // conceivable pair -- the rows whose score bound fits (8-bit lanes) and the rest (16-bit / literal)
            struct Range { uint32_t lo, hi; TileClass tc; };
            std::vector<Range> ranges;
            {
                TileClass tc0;
                long long limit = -1;
                classify(ctx, la, lb, X, p, thr, &tc0, -1, &limit);
                uint32_t split = rb;  // rows [rb, split) fit 8-bit lanes by their bound
                if (refine && tc0.path != PATH_U8 && limit >= 0) {
                    const uint32_t lim = (uint32_t)std::min<long long>(limit, BCAP - 1);  // a clamped bound never passes
                    split = all_rows_fit ? re   // (caller order kept: every row of the bucket is within the limit)
                                         : (uint32_t)(std::upper_bound(bound_sorted.begin() + rb, bound_sorted.begin() + re, lim) -
                                                      bound_sorted.begin());
                    if (split > rb) {
                        TileClass t8;
                        classify(ctx, la, lb, X, p, thr, &t8, lim);
                        if (t8.path == PATH_U8) ranges.push_back(Range{rb, split, t8});
                        else split = rb;
                    }
                }
                if (split < re) ranges.push_back(Range{split, re, tc0});
            }
            for (const Range &rg : ranges) {
                const TileClass &tc = rg.tc;
                const int cls = (int)classes.size();
                classes.push_back(tc);
                class_of[la * 64 + lb] = cls;
                if (tc.path == PATH_U8) S.classes_u8++;
                else if (tc.path == PATH_U16) S.classes_u16++;
                else S.classes_direct++;
                // launch group: (kernel family, entry dwords | length difference, column capacity)
                const bool rows = use_rows && tc.path == PATH_U8 && la >= lb &&
                                  (pl.rows_exact || rows_kernel_available(X, la, lb, false));
                const int lbk = rows ? (pl.rows_exact ? lb : rows_cap_for(lb)) : pl.exact ? 12 : swar_lbmax_for(lb);
                const uint32_t R = rows ? (uint32_t)rows_per_tile_rows(X, la - lb, lbk, pl.rows_exact)
                                        : tc.path == PATH_DIRECT ? 16u : (uint32_t)swar_rows_per_tile(lbk, tc.nw, pl.exact);
                if (rows) S.classes_rows++;
                std::vector<Tile> &dst = grouped[rows ? std::make_tuple((int)PATH_ROWS, la - lb, lbk)
                                                      : std::make_tuple((int)tc.path, tc.path == PATH_DIRECT ? 0 : (int)tc.nw,
                                                                        tc.path == PATH_DIRECT ? 0 : lbk)];
                for (uint32_t r0 = rg.lo; r0 < rg.hi; r0 += R) {
                    const bool mine = (row_chunk_counter++ % n_parts) == part;
                    if (!mine) continue;
                    const uint32_t nr = std::min(R, rg.hi - r0);
                    uint32_t c_lo = cb, c_hi = ce;
                    if (same && ctx->symmetric) c_lo = r0 + 1;  // triangle: columns after the first row of the chunk
                    // equal column runs (whole 256-column batches) instead of full runs + one short rest:
                    // no tiny tiles whose table build is not amortised, and an even tail
                    uint32_t run = COLS;
                    if (c_hi > c_lo && equal_runs) {
                        const uint32_t k_runs = (c_hi - c_lo + COLS - 1) / COLS;
                        run = ((c_hi - c_lo + k_runs - 1) / k_runs + 255u) & ~255u;
                        run = std::min(run, COLS);
                    }
                    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += run) {
                        Tile t{};
                        t.row0 = r0; t.nrows = nr;
                        t.col0 = c0; t.ncols = std::min(run, c_hi - c0);
                        t.cls = (uint32_t)cls;
                        const bool overlap = same && c0 < r0 + nr && c0 + t.ncols > r0;
                        t.diag = overlap ? (ctx->symmetric ? 1u : 2u) : 0u;
                        uint64_t pairs = (uint64_t)nr * t.ncols;
                        if (t.diag == 1) {
                            pairs = 0;
                            for (uint32_t r = r0; r < r0 + nr; r++) {
                                const uint32_t lo = std::max(c0, r + 1), hi = c0 + t.ncols;
                                if (hi > lo) pairs += hi - lo;
                            }
                        } else if (t.diag == 2) {
                            for (uint32_t r = r0; r < r0 + nr; r++)
                                if (r >= c0 && r < c0 + t.ncols) pairs--;
                        }
                        if (pairs == 0) continue;
                        S.pairs_scored += pairs;
                        t.pad0 = (r0 < band_end[la] || c0 < band_end[lb]) ? 1u : 0u;   // band tile (host-side flag)
                        if (t.pad0) pl.band_pairs += pairs;
                        dst.push_back(t);
                    }
                }
            }
        }
    }
    plan_lap("classes and tiles");
    std::vector<Tile> tiles;
    for (auto &kv : grouped) {
        if (kv.second.empty()) continue;
        // workgroups are dispatched in tile order: biggest tiles first keeps the tail of the launch short
        // (band tiles first: they are launched on their own by hmk_greedy_cluster)
        // (10^6 sequences: a million tiles; the stable sort of them was 30 of the plan's 55 ms on one thread)
        parallel_stable_sort(kv.second, [](const Tile &a, const Tile &b) {
            if (a.pad0 != b.pad0) return a.pad0 > b.pad0;
            return (uint64_t)a.nrows * a.ncols > (uint64_t)b.nrows * b.ncols;
        });
        uint32_t n_band = 0;
        uint64_t work = 0;
        for (const Tile &t : kv.second) {
            n_band += t.pad0;
            const TileClass &tc = classes[t.cls];
            const int lb = std::min((int)tc.la, (int)tc.lb), d = std::abs((int)tc.la - (int)tc.lb);
            work += (uint64_t)t.nrows * t.ncols * (uint64_t)std::max(1, lb * (2 * X + d + 1) - X * (X + 1));
        }
        pl.groups.push_back(Group{std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first),
                                  (uint32_t)tiles.size(), (uint32_t)kv.second.size(), n_band, work});
        tiles.insert(tiles.end(), kv.second.begin(), kv.second.end());
    }
    S.n_tiles = (uint32_t)tiles.size();
    plan_lap("tile order");

    // ---- device copies ------------------------------------------------------------
    std::vector<uint8_t> res_sorted((size_t)n * pl.lpad + 16, 0);   // + 16: the row-packed kernel's unaligned tail loads may touch the bytes after the last row
    {   // (rows are independent: several threads for large sets -- 10 ms on one at 10^6)
        const unsigned hw = usable_cpus();
        const unsigned T = n >= (1u << 18) ? std::max(1u, std::min(8u, hw ? hw : 1u)) : 1u;
        auto fill = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t s = lo; s < hi; s++) {
                const uint32_t k = perm[s];
                for (uint32_t q = 0; q < ctx->len[k]; q++)
                    res_sorted[(size_t)s * pl.lpad + q] = (uint8_t)(ctx->res[ctx->off[k] + q] * (pl.exact ? 8 : 1));
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; t++) pool.emplace_back(fill, (uint32_t)((uint64_t)n * t / T), (uint32_t)((uint64_t)n * (t + 1) / T));
        fill(0, (uint32_t)((uint64_t)n / T));
        for (std::thread &th : pool) th.join();
    }
    const int bias = ctx->min_m < 0 ? -ctx->min_m : 0;
    uint8_t mb[576];
    for (int e = 0; e < 576; e++) {
        const long long v = (long long)ctx->M[e] + bias;
        mb[e] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);  // only read by classes that passed the range check
    }
    HIPCHK(ctx, hipMalloc((void **)&pl.d_res_sorted, res_sorted.size()));
    HIPCHK(ctx, hipMemcpy(pl.d_res_sorted, res_sorted.data(), res_sorted.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void **)&pl.d_perm, (size_t)n * 4));
    HIPCHK(ctx, hipMemcpy(pl.d_perm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    pl.perm_identity = true;
    for (uint32_t q = 0; q < n && pl.perm_identity; q++) pl.perm_identity = perm[q] == q;
    HIPCHK(ctx, hipMalloc((void **)&pl.d_mb, 576));
    HIPCHK(ctx, hipMemcpy(pl.d_mb, mb, 576, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void **)&pl.d_classes, std::max<size_t>(1, classes.size()) * sizeof(TileClass)));
    if (!classes.empty())
        HIPCHK(ctx, hipMemcpy(pl.d_classes, classes.data(), classes.size() * sizeof(TileClass), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void **)&pl.d_tiles, std::max<size_t>(1, tiles.size()) * sizeof(Tile)));
    if (!tiles.empty())
        HIPCHK(ctx, hipMemcpy(pl.d_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice));
    pl.X = X; pl.p = p; pl.thr = thr; pl.part = part; pl.n_parts = n_parts;
    pl.valid = true;
    plan_lap("device copies");
    return HMK_OK;
}
// ---- LocalAlignmentScorer neighbour pass: plan (tiles of ordered length classes) + launch ----------------
void free_plan_local(PlanLocal &pl) {
    if (pl.d_res_sorted) (void)hipFree(pl.d_res_sorted);
    if (pl.d_perm) (void)hipFree(pl.d_perm);
    if (pl.d_classes) (void)hipFree(pl.d_classes);
    if (pl.d_tiles) (void)hipFree(pl.d_tiles);
    pl = PlanLocal();
}

int build_plan_local(hmk_ctx *ctx, uint32_t part, uint32_t n_parts) {
    PlanLocal &pl = ctx->plan_local;
    if (pl.valid && pl.part == part && pl.n_parts == n_parts) return HMK_OK;
    free_plan_local(pl);
    const uint32_t n = ctx->n;
    if (n == 0) return fail(ctx, HMK_ERR_NO_SEQUENCES, "no sequences set (hmk_set_sequences)");
    if (n_parts == 0 || part >= n_parts) return fail(ctx, HMK_ERR_BAD_ARG, "part must be < n_parts");
    uint32_t bucket[HMK_MAX_LEN + 2] = {0};
    for (uint32_t k = 0; k < n; k++) bucket[ctx->len[k] + 1]++;
    for (int l = 0; l <= HMK_MAX_LEN; l++) bucket[l + 1] += bucket[l];
    std::vector<uint32_t> perm(n);
    {
        uint32_t fill[HMK_MAX_LEN + 2];
        std::memcpy(fill, bucket, sizeof(fill));
        for (uint32_t k = 0; k < n; k++) perm[fill[ctx->len[k]]++] = k;
    }
    std::vector<TileClass> classes;
    std::vector<Tile> tiles;
    const uint32_t R = 16, COLS = 16384;
    uint64_t row_chunk_counter = 0;
    pl.pairs = 0;
    for (int la = 1; la <= HMK_MAX_LEN; la++) {          // rows = seq1 (lines)
        const uint32_t rb = bucket[la], re = bucket[la + 1];
        if (rb == re) continue;
        for (int lb = 1; lb <= HMK_MAX_LEN; lb++) {      // columns = seq2
            const uint32_t cb = bucket[lb], ce = bucket[lb + 1];
            if (cb == ce) continue;
            TileClass tc{};
            tc.la = (uint8_t)la;
            tc.lb = (uint8_t)lb;
            const uint32_t cls = (uint32_t)classes.size();
            classes.push_back(tc);
            for (uint32_t r0 = rb; r0 < re; r0 += R) {
                if ((row_chunk_counter++ % n_parts) != part) continue;
                const uint32_t nr = std::min(R, re - r0);
                for (uint32_t c0 = cb; c0 < ce; c0 += COLS) {
                    Tile t{};
                    t.row0 = r0; t.nrows = nr; t.col0 = c0; t.ncols = std::min(COLS, ce - c0); t.cls = cls;
                    const bool overlap = la == lb && c0 < r0 + nr && c0 + t.ncols > r0;
                    t.diag = overlap ? 2u : 0u;
                    uint64_t pairs = (uint64_t)nr * t.ncols;
                    if (overlap)
                        for (uint32_t r = r0; r < r0 + nr; r++)
                            if (r >= c0 && r < c0 + t.ncols) pairs--;
                    if (pairs == 0) continue;
                    pl.pairs += pairs;
                    tiles.push_back(t);
                }
            }
        }
    }
    pl.n_tiles = (uint32_t)tiles.size();
    std::vector<uint8_t> res_sorted((size_t)n * 32, 0);
    for (uint32_t s = 0; s < n; s++) std::memcpy(&res_sorted[(size_t)s * 32], &ctx->res[ctx->off[perm[s]]], ctx->len[perm[s]]);
    HIPCHK(ctx, hipMalloc((void **)&pl.d_res_sorted, res_sorted.size()));
    HIPCHK(ctx, hipMemcpy(pl.d_res_sorted, res_sorted.data(), res_sorted.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void **)&pl.d_perm, (size_t)n * 4));
    HIPCHK(ctx, hipMemcpy(pl.d_perm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    pl.perm_identity = true;
    for (uint32_t q = 0; q < n && pl.perm_identity; q++) pl.perm_identity = perm[q] == q;
    HIPCHK(ctx, hipMalloc((void **)&pl.d_classes, std::max<size_t>(1, classes.size()) * sizeof(TileClass)));
    if (!classes.empty())
        HIPCHK(ctx, hipMemcpy(pl.d_classes, classes.data(), classes.size() * sizeof(TileClass), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void **)&pl.d_tiles, std::max<size_t>(1, tiles.size()) * sizeof(Tile)));
    if (!tiles.empty())
        HIPCHK(ctx, hipMemcpy(pl.d_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice));
    pl.part = part; pl.n_parts = n_parts;
    pl.valid = true;
    return HMK_OK;
}

} }  // namespace hmk::impl
