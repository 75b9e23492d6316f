// hmk_pass.cpp -- launching the neighbour passes (neighbors_dev_locked: groups, side streams, band / rest), the host-buffer
// form that grows its edge buffer (neighbors_grow), the LocalAlignmentScorer pass, the pair and block probes.
#include "hmk_ctx.h"

namespace hmk { namespace impl {

// The side streams of a mixed-length pass: three streams that RUN SIDE BY SIDE.  Streams share the process's few hardware queues
// (4 per priority by default) and two streams on one queue serialise; which queue a new stream lands on depends on how many
// the process has created before (a probe of ten fresh streams: queues a b c d d c b a d c), so "three new streams" may be two
// queues -- BASELINE config 4a took 4.13 ms in a fresh process, 4.57-4.61 ms at the end of a bench.py run and 5.27 ms on three
// streams created one after the other behind the context's own (round 3's sweep: 1 / 2 / 3 streams 5.73 / 4.56 / 4.41 ms).
// Measured with the probe in place: 4.26-4.37 ms inside bench.py and on its own alike (0.82-0.84 of the LDS-byte ideal); four probed
// streams 4.62, one or two of them at high priority 4.31-4.60, five / six 4.50 / 4.62.
// So the streams are PROBED, once per context: a one-wave kernel that spins for 60 us notes its start and end on each of two
// streams; they run side by side iff the second started before the first ended.  New streams are created until three pass
// pairwise (at most 12; whatever was found is used then).
static int make_side_streams(hmk_ctx *ctx) {
    constexpr int n_side = hmk_ctx::N_SIDE, MAX_TRIES = 12;
    unsigned long long *when = nullptr;
    HIPCHK(ctx, hipHostMalloc((void **)&when, 64, hipHostMallocDefault));
    auto side_by_side = [&](hipStream_t a, hipStream_t b, bool *yes) -> hipError_t {
        std::memset(when, 0, 64);
        hipError_t e = launch_probe_spin(when, 6000, a);
        if (e == hipSuccess) e = launch_probe_spin(when + 2, 6000, b);
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess) e = hipStreamSynchronize(b);
        *yes = when[2] < when[1] && when[0] < when[3];
        return e;
    };
    std::vector<hipStream_t> kept, dropped;
    hipError_t e = hipSuccess;
    for (int t = 0; t < MAX_TRIES && (int)kept.size() < n_side && e == hipSuccess; t++) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) break;
        bool ok = true;
        for (size_t k = 0; k < kept.size() && ok && e == hipSuccess; k++) e = side_by_side(kept[k], s, &ok);
        (ok ? kept : dropped).push_back(s);
    }
    while ((int)kept.size() < n_side && !dropped.empty()) { kept.push_back(dropped.back()); dropped.pop_back(); }   // (fewer queues than streams: serialise)
    for (hipStream_t s : dropped) HMK_QUIET(hipStreamDestroy(s));
    HMK_QUIET(hipHostFree(when));
    if (e != hipSuccess || (int)kept.size() < n_side) {
        for (hipStream_t s : kept) HMK_QUIET(hipStreamDestroy(s));
        return fail(ctx, HMK_ERR_DEVICE, std::string("side streams of the mixed-length pass: ") + hipGetErrorString(e != hipSuccess ? e : hipErrorUnknown));
    }
    for (int k = 0; k < n_side; k++) ctx->side[k] = kept[k];
    return HMK_OK;
}

int neighbors_dev_locked(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, void *d_edges,
                         uint64_t capacity, void *d_counts, hipStream_t stream, int which,
                         int64_t band_rows, uint32_t *d_deg, uint32_t *d_deg_lo) {
    int st = need_device(ctx);
    if (st) return st;
    if (!d_edges || !d_counts || capacity < HMK_EDGE_SHARDS)
        return fail(ctx, HMK_ERR_BAD_ARG, "d_edges/d_counts must be device buffers, capacity >= HMK_EDGE_SHARDS");
    st = build_plan(ctx, X, p, thr, part, n_parts, band_rows);
    if (st) return st;
    Plan &pl = ctx->plan;
    if (which != LAUNCH_REST) HIPCHK(ctx, hipMemsetAsync(d_counts, 0, HMK_EDGE_SHARDS * sizeof(unsigned long long), stream));
    NeighborParams P{};
    P.res_sorted = pl.d_res_sorted;
    P.perm = pl.d_perm;
    P.perm_identity = pl.perm_identity ? 1u : 0u;
    P.mb = pl.d_mb;
    P.classes = pl.d_classes;
    P.tiles = pl.d_tiles;
    P.edges = (uint64_t *)d_edges;
    P.counts = (unsigned long long *)d_counts;
    P.cap_per_shard = capacity / HMK_EDGE_SHARDS;
    P.n_tiles = pl.stats.n_tiles;
    P.lpad = (uint32_t)pl.lpad;
    P.symmetric = ctx->symmetric ? 1u : 0u;
    P.deg = d_deg;
    P.deg_m_offset = (d_deg && d_deg_lo) ? (uint32_t)(d_deg_lo - d_deg) : 0u;   // split counters: upper counts, then lower counts
    // one launch per (lane path, entry width, column capacity) group.  A mixed-length plan has two dozen of them: fork them
    // round-robin onto three side streams so that one group's tail overlaps the next group's start, and join back into `stream`,
    // which itself only records the fork event and waits for the joins.
    const bool fork = pl.groups.size() > 2;
    constexpr int n_side = hmk_ctx::N_SIDE;
    hipStream_t *sides = ctx->side;
    if (fork) {
        if (!ctx->ev_fork) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        if (!ctx->side[n_side - 1]) { const int rc = make_side_streams(ctx); if (rc != HMK_OK) return rc; }
        for (int k = 0; k < n_side; k++)
            if (!ctx->ev_join[k]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, stream));
        for (int k = 0; k < n_side; k++) HIPCHK(ctx, hipStreamWaitEvent(sides[k], ctx->ev_fork, 0));
    }
    // biggest groups first
    std::vector<const Group *> order;
    for (const Group &g : pl.groups) order.push_back(&g);
    std::stable_sort(order.begin(), order.end(), [](const Group *a, const Group *b) { return a->work > b->work; });
    // each launch goes to the side stream with the least work so far (cells to add): dealt round-robin by tile count, one stream of
    // BASELINE config 4a ran dry a millisecond before the other two (kernel trace: 3.1 / 4.1 / 4.1 ms of kernels per stream)
    uint64_t load[hmk_ctx::N_SIDE] = {0};
    for (const Group *gp : order) {
        const Group &g = *gp;
        int least = 0;
        for (int k = 1; k < n_side; k++) if (load[k] < load[least]) least = k;
        load[least] += g.work;
        hipStream_t s = fork ? sides[least] : stream;
        const uint32_t t0 = which == LAUNCH_REST ? g.base + g.band : g.base;
        const uint32_t cnt = which == LAUNCH_ALL ? g.count : which == LAUNCH_BAND ? g.band : g.count - g.band;
        if (g.path == PATH_DIRECT)
            HIPCHK(ctx, launch_neighbors_direct(P, t0, cnt, ctx->d_M, X, p, thr, s));
        else if (g.path == PATH_ROWS)
            HIPCHK(ctx, launch_neighbors_rows(X, g.nw, g.lbk, pl.rows_exact, P, t0, cnt, s));
        else
            HIPCHK(ctx, launch_neighbors_swar(g.lbk, g.nw, pl.exact, P, t0, cnt, s));
    }
    if (fork)
        for (int k = 0; k < n_side; k++) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_join[k], sides[k]));
            HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[k], 0));
        }
    return HMK_OK;
}

// Runs the neighbour pass into the context's own device buffer, growing it until
// every segment fits, and returns the per-segment counts.
// the tagged-max SW kernels carry 4 * value + direction in int8 table bytes
bool local_enc(const hmk_ctx *ctx, int gap_open, int gap_extend) {
    return ctx->min_m >= -31 && ctx->max_m <= 31 && gap_open >= -31 && gap_extend >= -31 && gap_open <= 0 && gap_extend <= 0;
}

int neighbors_internal(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, uint64_t want_cap,
                       unsigned long long counts[HMK_EDGE_SHARDS], double *kernel_ms) {
    return neighbors_grow(ctx, want_cap, counts, kernel_ms, [&](uint64_t *d_edges, uint64_t cap, unsigned long long *d_counts) {
        return neighbors_dev_locked(ctx, X, p, thr, part, n_parts, d_edges, cap, d_counts, nullptr);
    });
}
int neighbors_local_dev_locked(hmk_ctx *ctx, int gap_open, int gap_extend, int thr, uint32_t part, uint32_t n_parts,
                               uint64_t *d_edges, uint64_t capacity, unsigned long long *d_counts, hipStream_t stream) {
    int st = need_device(ctx);
    if (st) return st;
    // the striped register kernels take gap penalties <= 0 and int8 matrix entries; anything else (the reference imposes
    // neither, LocalAlignmentScorer.java:43-55) runs the literal DP on the same tiles
    const bool literal = gap_open > 0 || gap_extend > 0 || ctx->min_m < -127 || ctx->max_m > 127 || ctx->sw.local_literal;
    {   // edge scores travel as int16
        const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                              2LL * ctx->max_len * (long long)std::max(0, std::max(gap_open, gap_extend));
        if (top > 32767 || thr < -30000 || thr > 30000)
            return fail(ctx, HMK_ERR_BAD_ARG, "scores up to " + std::to_string(top) + " are possible with this matrix / these gap penalties "
                                               "(or the threshold is outside [-30000, 30000]): they do not fit the int16 score of a packed edge");
    }
    st = build_plan_local(ctx, part, n_parts);
    if (st) return st;
    PlanLocal &pl = ctx->plan_local;
    HIPCHK(ctx, hipMemsetAsync(d_counts, 0, HMK_EDGE_SHARDS * sizeof(unsigned long long), stream));
    NeighborParams P{};
    P.res_sorted = pl.d_res_sorted;
    P.perm = pl.d_perm;
    P.perm_identity = pl.perm_identity ? 1u : 0u;
    P.classes = pl.d_classes;
    P.tiles = pl.d_tiles;
    P.edges = d_edges;
    P.counts = d_counts;
    P.cap_per_shard = capacity / HMK_EDGE_SHARDS;
    P.n_tiles = pl.n_tiles;
    P.lpad = 32;
    P.symmetric = 0;
    P.row_is_m = 1;
    if (literal)
        HIPCHK(ctx, launch_neighbors_local_literal(P, 0, pl.n_tiles, ctx->d_M, gap_open, gap_extend, thr, stream));
    else
        HIPCHK(ctx, launch_neighbors_local(ctx->max_len, local_enc(ctx, gap_open, gap_extend), ctx->sw.local_signed, ctx->sw.local_no_pk, P, 0, pl.n_tiles,
                                           ctx->d_M, gap_open, gap_extend, thr, stream));
    return HMK_OK;
}

// HIP-event bracket around the probe kernels (hmk_last_kernel_ms)
void timer_start(hmk_ctx *ctx) {
    if (!ctx->ev0) { (void)hipEventCreate(&ctx->ev0); (void)hipEventCreate(&ctx->ev1); }
    ctx->last_kernel_ms = 0;
    (void)hipEventRecord(ctx->ev0, nullptr);
}
void timer_stop(hmk_ctx *ctx) {
    float ms = 0;
    if (hipEventRecord(ctx->ev1, nullptr) == hipSuccess && hipEventSynchronize(ctx->ev1) == hipSuccess &&
        hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess)
        ctx->last_kernel_ms += ms;
}

int check_pairs(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, bool shifted, int X) {
    if (ctx->n == 0) return fail(ctx, HMK_ERR_NO_SEQUENCES, "no sequences set (hmk_set_sequences)");
    if (n_pairs && (!i || !j)) return fail(ctx, HMK_ERR_BAD_ARG, "null index array");
    for (uint64_t k = 0; k < n_pairs; k++) {
        if (i[k] >= ctx->n || j[k] >= ctx->n) return fail(ctx, HMK_ERR_BAD_ARG, "pair index out of range");
        if (shifted && X >= std::min(ctx->len[i[k]], ctx->len[j[k]]))
            return fail(ctx, HMK_ERR_SHIFT_TOO_BIG,
                        "Shift too big: " + std::to_string(std::min(ctx->len[i[k]], ctx->len[j[k]]) - 1) +
                            " is maximum, but " + std::to_string(X) + " found");
    }
    return HMK_OK;
}

int score_pairs(hmk_ctx *ctx, int scorer, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, int a, int b,
                int32_t *out, int32_t *out_shift) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    int st = need_device(ctx);
    if (st) return st;
    if (scorer == 0 && a < 0) return fail(ctx, HMK_ERR_BAD_ARG, "max_shift must be >= 0");
    st = check_pairs(ctx, i, j, n_pairs, scorer == 0, a);
    if (st) return st;
    if (n_pairs && !out) return fail(ctx, HMK_ERR_BAD_ARG, "null output");
    st = ensure_res32(ctx);
    if (st) return st;
    const uint64_t CH = 1ull << 24;
    uint32_t *d_i = nullptr, *d_j = nullptr;
    int32_t *d_out = nullptr, *d_shift = nullptr;
    const uint64_t ch = std::min<uint64_t>(CH, std::max<uint64_t>(n_pairs, 1));
    HIPCHK(ctx, hipMalloc((void **)&d_i, ch * 4));
    HIPCHK(ctx, hipMalloc((void **)&d_j, ch * 4));
    HIPCHK(ctx, hipMalloc((void **)&d_out, ch * 4));
    if (out_shift) HIPCHK(ctx, hipMalloc((void **)&d_shift, ch * 4));
    st = HMK_OK;
    for (uint64_t o = 0; o < n_pairs && st == HMK_OK; o += ch) {
        const uint64_t m = std::min(ch, n_pairs - o);
        hipError_t e = hipMemcpy(d_i, i + o, m * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_j, j + o, m * 4, hipMemcpyHostToDevice);
        double acc = ctx->last_kernel_ms;
        timer_start(ctx);
        if (e == hipSuccess) e = launch_pairs(scorer, ctx->d_res32, ctx->d_len, ctx->d_M, d_i, d_j, m, 0, 0, 1, a, b, d_out, d_shift, nullptr);
        timer_stop(ctx);
        ctx->last_kernel_ms += (o == 0 ? 0.0 : acc);
        if (e == hipSuccess) e = hipMemcpy(out + o, d_out, m * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_shift) e = hipMemcpy(out_shift + o, d_shift, m * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(ctx, HMK_ERR_DEVICE, std::string("score_pairs: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_i);
    (void)hipFree(d_j);
    (void)hipFree(d_out);
    if (d_shift) (void)hipFree(d_shift);
    return st;
}

int score_block(hmk_ctx *ctx, int scorer, uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int a, int b,
                int32_t *out) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    int st = need_device(ctx);
    if (st) return st;
    if (ctx->n == 0) return fail(ctx, HMK_ERR_NO_SEQUENCES, "no sequences set (hmk_set_sequences)");
    if (r0 > r1 || c0 > c1 || r1 > ctx->n || c1 > ctx->n) return fail(ctx, HMK_ERR_BAD_ARG, "block outside [0, n)");
    const uint64_t n_pairs = (uint64_t)(r1 - r0) * (c1 - c0);
    if (n_pairs == 0) return HMK_OK;
    if (!out) return fail(ctx, HMK_ERR_BAD_ARG, "null output");
    if (scorer == 0) {
        if (a < 0) return fail(ctx, HMK_ERR_BAD_ARG, "max_shift must be >= 0");
        int mn = 255;
        for (uint32_t r = r0; r < r1; r++) mn = std::min<int>(mn, ctx->len[r]);
        for (uint32_t c = c0; c < c1; c++) mn = std::min<int>(mn, ctx->len[c]);
        if (a >= mn)
            return fail(ctx, HMK_ERR_SHIFT_TOO_BIG, "Shift too big: " + std::to_string(mn - 1) + " is maximum, but " +
                                                        std::to_string(a) + " found");
    }
    {
        const int st32 = ensure_res32(ctx);
        if (st32) return st32;
    }
    int32_t *d_out = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_out, n_pairs * 4));
    hipError_t e;
    timer_start(ctx);
    // LocalAlignmentScorer: the register-resident striped kernel when its preconditions hold
    const bool fast_local = scorer == 1 && a <= 0 && b <= 0 && ctx->min_m >= -127 && ctx->max_m <= 127 && !ctx->sw.local_literal;
    if (fast_local)
        e = launch_local_block(ctx->max_len, local_enc(ctx, a, b), ctx->sw.local_signed, ctx->sw.local_no_pk, ctx->d_res32, ctx->d_len, ctx->d_M, r0, r1,
                               c0, c1, a, b, d_out, nullptr);
    else
        e = launch_pairs(scorer, ctx->d_res32, ctx->d_len, ctx->d_M, nullptr, nullptr, n_pairs, r0, c0, c1 - c0, a, b,
                         d_out, nullptr, nullptr);
    timer_stop(ctx);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n_pairs * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, HMK_ERR_DEVICE, std::string("score_block: ") + hipGetErrorString(e));
    return HMK_OK;
}

} }  // namespace hmk::impl
