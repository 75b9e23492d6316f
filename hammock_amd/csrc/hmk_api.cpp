// hmk_api.cpp -- the C ABI of libhammock_hip.so (include/hammock_hip.h): the extern "C" entry points.  Context, planner, passes,
// the clustering tail and the multi-device call live in hmk_common / hmk_plan / hmk_pass / hmk_cluster / hmk_multi.cpp (hmk_ctx.h).
// Host code only; the kernels live in k_*.hip (launchers declared in hmk_kernels.h).
#include "hmk_ctx.h"

using namespace hmk::impl;

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int hmk_abi_version(void) { return HMK_ABI_VERSION; }

double hmk_last_kernel_ms(const hmk_ctx *ctx) { return ctx ? ctx->last_kernel_ms : 0.0; }

const char *hmk_last_error(const hmk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int hmk_create(const int32_t *matrix, int device, hmk_ctx **out) {
    if (!matrix || !out) return fail(nullptr, HMK_ERR_BAD_ARG, "hmk_create: null argument");
    *out = nullptr;
    hmk_ctx *ctx = new (std::nothrow) hmk_ctx();
    if (!ctx) return fail(nullptr, HMK_ERR_OOM, "out of memory");
    std::memcpy(ctx->M, matrix, sizeof(ctx->M));
    ctx->min_m = ctx->max_m = matrix[0];
    for (int e = 0; e < 576; e++) {
        ctx->min_m = std::min(ctx->min_m, matrix[e]);
        ctx->max_m = std::max(ctx->max_m, matrix[e]);
        if (matrix[e] != matrix[(e % 24) * 24 + e / 24]) ctx->symmetric = false;
        if (matrix[e] < -1000 || matrix[e] > 1000) {
            delete ctx;
            return fail(nullptr, HMK_ERR_BAD_ARG, "matrix entries must lie in [-1000, 1000] (int16 edge scores)");
        }
    }
    ctx->device = device;
    ctx->sw.read();
    const bool timing = ctx->sw.greedy_timing;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[hmk] hmk_create: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    if (device >= 0) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        lap("hipGetDeviceCount (runtime start-up)");
        if (e != hipSuccess || device >= count) {
            delete ctx;
            return fail(nullptr, HMK_ERR_DEVICE,
                        "no HIP device " + std::to_string(device) + " (" +
                            (e != hipSuccess ? hipGetErrorString(e) : "device count " + std::to_string(count)) + ")");
        }
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            std::string arch = e == hipSuccess ? prop.gcnArchName : "?";
            delete ctx;
            return fail(nullptr, HMK_ERR_DEVICE, "libhammock_hip is built for gfx950 (MI355X) only; device is " + arch);
        }
        e = hipSetDevice(device);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_M, sizeof(ctx->M));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_M, ctx->M, sizeof(ctx->M), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            delete ctx;
            return fail(nullptr, HMK_ERR_DEVICE, std::string("hmk_create: ") + hipGetErrorString(e));
        }
        ctx->has_device = true;
        lap("device properties, first hipMalloc + copy");
        // What a first clustering call would otherwise pay: two HSA queues (streams), events, the pinned blocks (16-17 ms) and
        // the deferred load of the kernels' code objects (5-10 ms).  The reference constructs its scorer before it starts the
        // clock of "Clustering time" (Hammock.java:402-406), and a host can create the context while it still reads its input.
        {
            bool ok = greedy_streams(ctx) == HMK_OK;
            lap("streams, events, pinned blocks, first copies");
            ok = ok && warm_neighbors_module() == hipSuccess;
            lap("code objects: k_neighbors");
            ok = ok && warm_neighbors_rows_module() == hipSuccess;
            lap("code objects: k_neighbors_rows");
            ok = ok && warm_edges_module() == hipSuccess;
            lap("code objects: k_edges");
            if (!ok) (void)hipGetLastError();   // not fatal here: the first call tries again and reports
        }
    } else if (device != -1) {
        delete ctx;
        return fail(nullptr, HMK_ERR_BAD_ARG, "device must be >= 0 or -1 (host-only)");
    }
    *out = ctx;
    return HMK_OK;
}

// a call whose failure changes nothing here, and never leaves its error behind as the thread's "last error" for the next launch
// wrapper's hipGetLastError() to pick up
void hmk_destroy(hmk_ctx *ctx) {
    if (!ctx) return;
    for (hmk_ctx *peer : ctx->peers) hmk_destroy(peer);
    ctx->peers.clear();
    if (ctx->has_device) {
        HMK_QUIET(hipSetDevice(ctx->device));
        (void)join_late_buffers(ctx);
        free_plan(ctx->plan);
        free_plan_local(ctx->plan_local);
        if (ctx->d_res32) HMK_QUIET(hipFree(ctx->d_res32));
        if (ctx->d_len) HMK_QUIET(hipFree(ctx->d_len));
        if (ctx->d_M) HMK_QUIET(hipFree(ctx->d_M));
        if (ctx->d_edges) HMK_QUIET(hipFree(ctx->d_edges));
        if (ctx->d_counts) HMK_QUIET(hipFree(ctx->d_counts));
        if (ctx->d_rows_scratch) HMK_QUIET(hipFree(ctx->d_rows_scratch));
        if (ctx->copy_stream) HMK_QUIET(hipStreamDestroy(ctx->copy_stream));
        for (int k = 0; k < hmk_ctx::N_SIDE; k++) {
            if (ctx->side[k]) HMK_QUIET(hipStreamDestroy(ctx->side[k]));
            if (ctx->ev_join[k]) HMK_QUIET(hipEventDestroy(ctx->ev_join[k]));
        }
        if (ctx->ev_fork) HMK_QUIET(hipEventDestroy(ctx->ev_fork));
        if (ctx->xfer_stream) HMK_QUIET(hipStreamDestroy(ctx->xfer_stream));   // (this device's transfers to the others, hmk_multi.cpp)
        if (ctx->h_start) HMK_QUIET(hipHostFree(ctx->h_start));
        if (ctx->h_stage) HMK_QUIET(hipHostFree(ctx->h_stage));
        if (ctx->h_adj) HMK_QUIET(hipHostFree(ctx->h_adj));
        if (ctx->h_counts) HMK_QUIET(hipHostFree(ctx->h_counts));
        if (ctx->h_loop) HMK_QUIET(hipHostFree(ctx->h_loop));
        for (int b = 0; b < SB_N; b++)
            if (ctx->sb[b].p) HMK_QUIET(hipFree(ctx->sb[b].p));
        if (ctx->gstream) HMK_QUIET(hipStreamDestroy(ctx->gstream));
        for (hipEvent_t ev : {ctx->ev_t0, ctx->ev_band, ctx->ev_edges, ctx->ev_csr, ctx->ev_bandcsr})
            if (ev) HMK_QUIET(hipEventDestroy(ev));
        if (ctx->ev0) HMK_QUIET(hipEventDestroy(ctx->ev0));
        if (ctx->ev1) HMK_QUIET(hipEventDestroy(ctx->ev1));
    }
    delete ctx;
}

int hmk_set_sequences(hmk_ctx *ctx, const uint8_t *residues, const uint32_t *offsets, const int32_t *sizes,
                      uint32_t n) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (n > HMK_MAX_SEQUENCES) return fail(ctx, HMK_ERR_BAD_ARG, "more than 2^24 sequences");
    if (n && (!residues || !offsets)) return fail(ctx, HMK_ERR_BAD_ARG, "null residues/offsets");
    if (n && offsets[0] != 0) return fail(ctx, HMK_ERR_BAD_ARG, "offsets[0] must be 0");
    std::vector<uint8_t> len(n);
    int mn = 1 << 30, mx = 0;
    for (uint32_t k = 0; k < n; k++) {
        if (offsets[k + 1] < offsets[k]) return fail(ctx, HMK_ERR_BAD_ARG, "offsets must be non-decreasing");
        const uint32_t l = offsets[k + 1] - offsets[k];
        if (l < 1 || l > HMK_MAX_LEN)
            return fail(ctx, HMK_ERR_BAD_ARG, "sequence " + std::to_string(k) + " has length " + std::to_string(l) +
                                                  "; the GPU kernels take 1.." + std::to_string(HMK_MAX_LEN));
        len[k] = (uint8_t)l;
        mn = std::min<int>(mn, l);
        mx = std::max<int>(mx, l);
        if (sizes && sizes[k] < 1) return fail(ctx, HMK_ERR_BAD_ARG, "sizes must be >= 1");
    }
    const uint32_t total = n ? offsets[n] : 0;
    {   // (12 MB at 10^6 sequences: several threads)
        const unsigned hw = usable_cpus();
        const unsigned T = total >= (1u << 22) ? std::max(1u, std::min(8u, hw ? hw : 1u)) : 1u;
        std::atomic<bool> bad{false};
        auto check = [&](uint32_t lo, uint32_t hi) {
            uint8_t worst = 0;
            for (uint32_t q = lo; q < hi; q++) worst = std::max(worst, residues[q]);
            if (worst >= HMK_ALPHABET) bad.store(true);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; t++) pool.emplace_back(check, (uint32_t)((uint64_t)total * t / T), (uint32_t)((uint64_t)total * (t + 1) / T));
        check(0, (uint32_t)((uint64_t)total / T));
        for (std::thread &th : pool) th.join();
        if (bad.load()) return fail(ctx, HMK_ERR_BAD_ARG, "residue index >= 24");
    }
    if (ctx->has_device) {
        int st = need_device(ctx);
        if (st) return st;
        free_plan(ctx->plan);
        free_plan_local(ctx->plan_local);
        if (ctx->d_res32) (void)hipFree(ctx->d_res32);
        if (ctx->d_len) (void)hipFree(ctx->d_len);
        ctx->d_res32 = nullptr;
        ctx->d_len = nullptr;
        // (the padded copy the pair / block scorers read is made at their first use: ensure_res32)
    }
    ctx->n = n;
    ctx->res.assign(residues, residues + total);
    ctx->off.assign(offsets, offsets + (n ? n + 1 : 0));
    if (!n) ctx->off.assign(1, 0);
    ctx->has_sizes = sizes != nullptr;
    if (sizes) ctx->sizes.assign(sizes, sizes + n);
    else ctx->sizes.clear();
    ctx->len.swap(len);
    ctx->min_len = n ? mn : 0;
    ctx->max_len = mx;
    for (hmk_ctx *peer : ctx->peers) {   // every device holds all residues (row-block sharding)
        const int st = hmk_set_sequences(peer, residues, offsets, sizes, n);
        if (st) return fail(ctx, st, "device " + std::to_string(peer->device) + ": " + peer->err);
    }
    return HMK_OK;
}

int hmk_score_pairs_shifted(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, int max_shift,
                            int shift_penalty, int32_t *out) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    return score_pairs(ctx, 0, i, j, n_pairs, max_shift, shift_penalty, out);
}

int hmk_score_with_shift(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, int max_shift,
                         int shift_penalty, int32_t *score, int32_t *shift) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    if (n_pairs && !shift) return fail(ctx, HMK_ERR_BAD_ARG, "null shift output");
    return score_pairs(ctx, 0, i, j, n_pairs, max_shift, shift_penalty, score, shift);
}

int hmk_score_pairs_local(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, int gap_open,
                          int gap_extend, int32_t *out) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    return score_pairs(ctx, 1, i, j, n_pairs, gap_open, gap_extend, out);
}

int hmk_score_block_shifted(hmk_ctx *ctx, uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int max_shift,
                            int shift_penalty, int32_t *out) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    return score_block(ctx, 0, r0, r1, c0, c1, max_shift, shift_penalty, out);
}

int hmk_score_block_local(hmk_ctx *ctx, uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int gap_open,
                          int gap_extend, int32_t *out) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    return score_block(ctx, 1, r0, r1, c0, c1, gap_open, gap_extend, out);
}

int hmk_neighbors_shifted_dev(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, uint32_t part,
                              uint32_t n_parts, void *d_edges, uint64_t capacity, void *d_counts, void *stream) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    return neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, part, n_parts, d_edges, capacity, d_counts,
                                (hipStream_t)stream);
}

int hmk_compact_edges_dev(hmk_ctx *ctx, const void *d_edges, uint64_t capacity, const void *d_counts, void *d_out,
                          uint64_t out_capacity, void *d_total, void *stream) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    int st = need_device(ctx);
    if (st) return st;
    if (!d_edges || !d_counts || !d_out || !d_total || capacity < HMK_EDGE_SHARDS)
        return fail(ctx, HMK_ERR_BAD_ARG, "hmk_compact_edges_dev: null buffer or capacity < HMK_EDGE_SHARDS");
    HIPCHK(ctx, launch_compact_edges((const uint64_t *)d_edges, capacity / HMK_EDGE_SHARDS,
                                     (const unsigned long long *)d_counts, (uint64_t *)d_out, out_capacity,
                                     (unsigned long long *)d_total, (hipStream_t)stream));
    return HMK_OK;
}

int hmk_pack_rows_dev(hmk_ctx *ctx, const void *d_edges, uint64_t capacity, const void *d_counts, int threshold,
                      void *d_row_start, void *d_adj, uint64_t adj_capacity, void *stream) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    int st = need_device(ctx);
    if (st) return st;
    if (!d_edges || !d_counts || !d_row_start || !d_adj || capacity < HMK_EDGE_SHARDS)
        return fail(ctx, HMK_ERR_BAD_ARG, "hmk_pack_rows_dev: null buffer or capacity < HMK_EDGE_SHARDS");
    if (!ctx->n) return fail(ctx, HMK_ERR_NO_SEQUENCES, "hmk_pack_rows_dev: no sequences set");
    if (adj_capacity > 0xFFFFFFFFull) return fail(ctx, HMK_ERR_BAD_ARG, "hmk_pack_rows_dev: row offsets are 32 bit");
    if (ctx->d_rows_scratch_n < ctx->n) {
        if (ctx->d_rows_scratch) {
            HIPCHK(ctx, hipDeviceSynchronize());  // an earlier call may still be using the old scratch
            (void)hipFree(ctx->d_rows_scratch);
            ctx->d_rows_scratch = nullptr;
            ctx->d_rows_scratch_n = 0;
        }
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_rows_scratch, pack_rows_scratch_bytes(ctx->n)));
        ctx->d_rows_scratch_n = ctx->n;
    }
    HIPCHK(ctx, launch_pack_rows((const uint64_t *)d_edges, capacity / HMK_EDGE_SHARDS, (const unsigned long long *)d_counts,
                                 ctx->n, threshold, ctx->d_rows_scratch, (uint32_t *)d_row_start, (uint32_t *)d_adj,
                                 adj_capacity, (hipStream_t)stream));
    return HMK_OK;
}

int hmk_unpack_rows_dev(hmk_ctx *ctx, const void *d_row_start, const void *d_adj, int threshold, void *d_edges_out,
                        uint64_t out_capacity, void *stream) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    int st = need_device(ctx);
    if (st) return st;
    if (!d_row_start || !d_adj || !d_edges_out) return fail(ctx, HMK_ERR_BAD_ARG, "hmk_unpack_rows_dev: null buffer");
    if (!ctx->n) return fail(ctx, HMK_ERR_NO_SEQUENCES, "hmk_unpack_rows_dev: no sequences set");
    HIPCHK(ctx, launch_unpack_rows((const uint32_t *)d_row_start, (const uint32_t *)d_adj, ctx->n, threshold,
                                   (uint64_t *)d_edges_out, out_capacity, (hipStream_t)stream));
    return HMK_OK;
}

int hmk_neighbors_last_plan(hmk_ctx *ctx, hmk_neighbor_stats *stats) {
    if (!ctx || !stats) return fail(ctx, HMK_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (!ctx->plan.valid) return fail(ctx, HMK_ERR_BAD_ARG, "no neighbour pass has been planned yet");
    *stats = ctx->plan.stats;
    return HMK_OK;
}

int hmk_neighbors_shifted(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, uint32_t part,
                          uint32_t n_parts, uint64_t *edges, uint64_t capacity, uint64_t *n_edges,
                          hmk_neighbor_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (!n_edges) return fail(ctx, HMK_ERR_BAD_ARG, "n_edges must not be null");
    unsigned long long counts[HMK_EDGE_SHARDS];
    double ms = 0;
    int st = neighbors_internal(ctx, max_shift, shift_penalty, threshold, part, n_parts, capacity, counts, &ms);
    if (st) return st;
    uint64_t total = 0;
    for (int s = 0; s < HMK_EDGE_SHARDS; s++) total += counts[s];
    *n_edges = total;
    if (stats) {
        *stats = ctx->plan.stats;
        stats->n_edges = total;
        stats->kernel_ms = ms;
    }
    if (total > capacity) return fail(ctx, HMK_ERR_CAPACITY, "edge buffer too small: " + std::to_string(total) + " needed");
    if (total && !edges) return fail(ctx, HMK_ERR_BAD_ARG, "null edge buffer");
    const uint64_t seg = ctx->d_edges_cap / HMK_EDGE_SHARDS;
    uint64_t o = 0;
    for (int s = 0; s < HMK_EDGE_SHARDS; s++) {
        if (counts[s])
            HIPCHK(ctx, hipMemcpy(edges + o, ctx->d_edges + (uint64_t)s * seg, counts[s] * sizeof(uint64_t), hipMemcpyDeviceToHost));
        o += counts[s];
    }
    return HMK_OK;
}

int hmk_neighbors_local(hmk_ctx *ctx, int gap_open, int gap_extend, int threshold, uint32_t part, uint32_t n_parts,
                        uint64_t *edges, uint64_t capacity, uint64_t *n_edges, hmk_neighbor_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (!n_edges) return fail(ctx, HMK_ERR_BAD_ARG, "n_edges must not be null");
    unsigned long long counts[HMK_EDGE_SHARDS];
    double ms = 0;
    int st = neighbors_grow(ctx, capacity, counts, &ms, [&](uint64_t *d_edges, uint64_t cap, unsigned long long *d_counts) {
        return neighbors_local_dev_locked(ctx, gap_open, gap_extend, threshold, part, n_parts, d_edges, cap, d_counts, nullptr);
    });
    if (st) return st;
    uint64_t total = 0;
    for (int s = 0; s < HMK_EDGE_SHARDS; s++) total += counts[s];
    *n_edges = total;
    if (stats) {
        *stats = hmk_neighbor_stats{};
        stats->n_edges = total;
        stats->pairs_scored = ctx->plan_local.pairs;
        stats->n_tiles = ctx->plan_local.n_tiles;
        stats->kernel_ms = ms;
    }
    if (total > capacity) return fail(ctx, HMK_ERR_CAPACITY, "edge buffer too small: " + std::to_string(total) + " needed");
    if (total && !edges) return fail(ctx, HMK_ERR_BAD_ARG, "null edge buffer");
    const uint64_t seg = ctx->d_edges_cap / HMK_EDGE_SHARDS;
    uint64_t o = 0;
    for (int s = 0; s < HMK_EDGE_SHARDS; s++) {
        if (counts[s])
            HIPCHK(ctx, hipMemcpy(edges + o, ctx->d_edges + (uint64_t)s * seg, counts[s] * sizeof(uint64_t), hipMemcpyDeviceToHost));
        o += counts[s];
    }
    return HMK_OK;
}

int hmk_greedy_from_edges(hmk_ctx *ctx, const uint64_t *edges, uint64_t n_edges, int symmetric, int threshold,
                          int max_clusters, int32_t *cluster_id, int32_t *result_order, int32_t *member_rank,
                          hmk_greedy_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (ctx->n && !cluster_id) return fail(ctx, HMK_ERR_BAD_ARG, "null cluster_id");
    if (n_edges && !edges) return fail(ctx, HMK_ERR_BAD_ARG, "null edge list");
    std::string err;
    int st = greedy_from_edges(ctx->n, ctx->has_sizes ? ctx->sizes.data() : nullptr, edges, n_edges, symmetric != 0,
                               threshold, max_clusters, cluster_id, result_order, member_rank, stats, &err, greedy_options(ctx));
    if (st) return fail(ctx, st, err);
    return HMK_OK;
}

// =============================================================================
// greedy clustering on a device-resident neighbour graph
// =============================================================================
}  // extern "C"

extern "C" {

int hmk_greedy_cluster(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int max_clusters,
                       int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (ctx->n && !cluster_id) return fail(ctx, HMK_ERR_BAD_ARG, "null cluster_id");
    hmk_greedy_stats local;
    if (!stats) stats = &local;
    std::memset(stats, 0, sizeof(*stats));
    ctx->phases = hmk_greedy_phases{};
    if (ctx->n == 0) return HMK_OK;  // cluster() of an empty list returns an empty list
    int st = need_device(ctx);
    if (st) return st;
    const auto t_entry = std::chrono::steady_clock::now();
    g_alloc_ms = 0.0;
    g_allocs = 0;
    st = greedy_streams(ctx);
    if (st) return st;
    if (!ctx->peers.empty())
        return greedy_cluster_multi(ctx, max_shift, shift_penalty, threshold, max_clusters, cluster_id, result_order, member_rank, stats);
    const auto t0 = std::chrono::steady_clock::now();
    const bool call_timing = ctx->sw.greedy_timing;
    auto call_lap = [&](const char *what) {
        if (call_timing) fprintf(stderr, "[hmk greedy] %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    const uint32_t n = ctx->n;
    hipStream_t S = ctx->gstream;
    // Band: phase 1 of the merge (LimitedGreedySequenceClusterer.java:77-120) reads the adjacency rows in order and
    // stops once maxClusters clusters exist, normally a little after row maxClusters.  The tiles that complete the first
    // band_rows rows are launched first, their rows are handed to the host while the rest of the pair space is being scored.
    int64_t band_rows = 0;
    if (max_clusters > 0 && n >= 16384 && !ctx->sw.no_band)
        band_rows = std::min<int64_t>(n, 2LL * max_clusters + 1024);
    if (band_rows * 2 > (int64_t)n) band_rows = 0;   // no point: the band would be most of the pass
    const int64_t band_req = band_rows;
    st = build_plan(ctx, max_shift, shift_penalty, threshold, 0, 1, band_req);
    if (st) return st;
    band_rows = ctx->plan.band_rows;   // 0 if the plan could not order its tiles by band
    ctx->phases.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // adjacency entries are 4 bytes (m << 8 | score - threshold) when no score can exceed threshold + 255
    const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                          (long long)std::max(0, shift_penalty) * ((ctx->max_len - ctx->min_len) + 2LL * max_shift);
    EdgeSource src;
    src.symmetric = ctx->symmetric;
    src.format_known = true;
    src.packed = top - threshold <= 255 && !ctx->sw.adj_8byte;
    src.base = threshold;
    src.check_overflow = true;
    if (!ctx->d_counts) HIPCHK(ctx, hipMalloc((void **)&ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
    HIPCHK(ctx, ensure_buf(ctx, SB_BCOUNTS, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
    uint64_t cap = first_edge_capacity(ctx, n);
    for (int attempt = 0; attempt < 4; attempt++) {
        st = grow_edge_buffer(ctx, cap);
        if (st) return st;
        // Everything cluster_on_device will want, BEFORE the pass is enqueued: a hipMalloc issued while the pass runs returns
        // when the pass is over (seen at 10^6: the band hand-over of a context's first call was enqueued 340 ms late, i.e.
        // after the scoring it is meant to overlap).  Grow-only buffers: steady-state calls find them all in place.
        st = reserve_tail_buffers(ctx, n, src.packed, (uint32_t)std::max<int64_t>(band_rows, 0));
        if (st) return st;
        call_lap("edge buffer ready");
        const uint64_t seg = ctx->d_edges_cap / HMK_EDGE_SHARDS;
        src.seg_cap = seg;
        src.segs = shard_segments(ctx->d_edges, seg, ctx->d_counts);
        src.adj_bound = (ctx->symmetric ? 2 : 1) * ctx->d_edges_cap;
        src.band_rows = (uint32_t)band_rows;
        // (Two launches, one after the other: the band launch's END is what lets the hand-over's kernels in -- a kernel of another
        // stream gets no workgroup slot while a launch still has workgroups waiting.  Band and rest side by side on two streams, one
        // launch with a counter the band tiles bump, CUs kept free by a mask: all measured, all slower)
        src.band_segs = shard_segments(ctx->d_edges, seg, buf<unsigned long long>(ctx, SB_BCOUNTS));
        // the neighbour kernel counts the rows' degrees while it writes the edges (the CSR build's first pass): symmetric scores: the
        // smaller end counts into up[], the larger into lo[] -- the same number of atomics as one total per row, and the lower counts
        // give the bucket sizes of the CSR's dealing pass without a pass over the edges (k_lower_count, 2 ms at 10^6)
        const bool split = ctx->symmetric;
        HIPCHK(ctx, ensure_buf(ctx, SB_DEG, (size_t)n * (split ? 8 : 4)));
        uint32_t *d_deg = buf<uint32_t>(ctx, SB_DEG), *d_deg_lo = split ? d_deg + n : nullptr;
        HIPCHK(ctx, hipMemsetAsync(d_deg, 0, (size_t)n * (split ? 8 : 4), S));
        src.deg_fused = true;
        src.deg_split = split;
        HIPCHK(ctx, hipEventRecord(ctx->ev_t0, S));
        if (band_rows > 0) {
            st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, 1, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S,
                                      LAUNCH_BAND, band_req, d_deg, d_deg_lo);
            if (st) return st;
            HIPCHK(ctx, hipMemcpyAsync(buf<void>(ctx, SB_BCOUNTS), ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long),
                                       hipMemcpyDeviceToDevice, S));
            HIPCHK(ctx, hipEventRecord(ctx->ev_band, S));
            call_lap("band tiles enqueued");
        }
        st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, 1, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S,
                                  band_rows > 0 ? LAUNCH_REST : LAUNCH_ALL, band_req, d_deg, d_deg_lo);
        call_lap("all tiles enqueued");
        if (st) { (void)hipStreamSynchronize(S); return st; }
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, S));
        HIPCHK(ctx, hipEventRecord(ctx->ev_edges, S));
        st = cluster_on_device(ctx, src, max_clusters, cluster_id, result_order, member_rank, stats, t0);
        if (st != ST_RETRY_OVERFLOW) break;
        unsigned long long mx = 0;
        for (int q = 0; q < HMK_EDGE_SHARDS; q++) mx = std::max(mx, ctx->h_counts[q]);
        cap = (uint64_t)HMK_EDGE_SHARDS * (mx + mx / 8 + 1024);  // a segment overflowed: grow and rescore
    }
    if (st == ST_RETRY_OVERFLOW) return fail(ctx, HMK_ERR_DEVICE, "internal edge buffer kept overflowing");
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_edges) == hipSuccess) ctx->phases.score_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev_edges, ctx->ev_csr) == hipSuccess) ctx->phases.csr_ms = ms;
    (void)hipGetLastError();   // (a call that left early never recorded these events: "invalid resource handle" must not stay behind as the thread's last error)
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stats->neighbors_ms = ctx->phases.score_ms;
    if (call_timing)
        fprintf(stderr, "[hmk greedy] call %.2f ms: streams/events/pinned block %.2f, plan %.2f, %d buffer (re)allocations %.2f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(),
                std::chrono::duration<double, std::milli>(t0 - t_entry).count(), ctx->phases.plan_ms, g_allocs, g_alloc_ms);
    return st;
}

}  // extern "C"

extern "C" {

int hmk_create_multi(const int32_t *matrix, const int *devices, int n_devices, hmk_ctx **out) {
    if (!matrix || !devices || !out || n_devices < 1) return fail(nullptr, HMK_ERR_BAD_ARG, "hmk_create_multi: null argument or no device");
    if ((uint32_t)n_devices > HMK_MAX_SEGS - HMK_EDGE_SHARDS + 1)
        return fail(nullptr, HMK_ERR_BAD_ARG, "hmk_create_multi: at most " + std::to_string(HMK_MAX_SEGS - HMK_EDGE_SHARDS + 1) + " devices");
    *out = nullptr;
    hmk_ctx *root = nullptr;
    int st = hmk_create(matrix, devices[0], &root);
    if (st) return st;
    if (!root->has_device) { hmk_destroy(root); return fail(nullptr, HMK_ERR_BAD_ARG, "hmk_create_multi: devices must be HIP ordinals >= 0"); }
    for (int d = 1; d < n_devices; d++) {
        hmk_ctx *peer = nullptr;
        st = devices[d] >= 0 ? hmk_create(matrix, devices[d], &peer) : fail(nullptr, HMK_ERR_BAD_ARG, "hmk_create_multi: devices must be HIP ordinals >= 0");
        if (st) { hmk_destroy(root); return st; }
        root->peers.push_back(peer);
        if (devices[d] != devices[0]) {   // direct xGMI copies peer -> root (a refusal leaves the staged path, still correct)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[d]) == hipSuccess && can) {
                (void)hipSetDevice(devices[0]);
                (void)hipDeviceEnablePeerAccess(devices[d], 0);
                (void)hipSetDevice(devices[d]);
                (void)hipDeviceEnablePeerAccess(devices[0], 0);
                (void)hipGetLastError();   // "already enabled" is fine
            }
        }
    }
    (void)hipSetDevice(devices[0]);
    *out = root;
    return HMK_OK;
}

int hmk_reserve(hmk_ctx *ctx, uint32_t n_sequences) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    const auto t_call = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    if (!ctx->has_device || n_sequences < 2) return HMK_OK;   // nothing to size
    const auto t_lock = std::chrono::steady_clock::now();
    int st = need_device(ctx);
    if (st == HMK_OK) st = greedy_streams(ctx);
    if (st) return st;
    const auto t_streams = std::chrono::steady_clock::now();
    if (!ctx->d_counts) HIPCHK(ctx, hipMalloc((void **)&ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
    HIPCHK(ctx, ensure_buf(ctx, SB_BCOUNTS, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
    {
        // (a multi-device root scores 1 / devices of the pair space: greedy_cluster_multi takes the larger of its own guess and
        // what is there, and sizes the root's adjacency from it -- the single-device guess made those twice as large as needed)
        const uint64_t G = 1 + ctx->peers.size();
        uint64_t cap = first_edge_capacity(ctx, n_sequences);
        if (G > 1 && ctx->sw.edge_guess == 0)
            cap = std::max<uint64_t>({(uint64_t)((double)cap / G * 1.25), (uint64_t)1 << 20, ctx->d_edges_cap});
        st = grow_edge_buffer(ctx, (cap + HMK_EDGE_SHARDS - 1) / HMK_EDGE_SHARDS * HMK_EDGE_SHARDS);
    }
    if (st) return st;
    const int64_t maxc = (int64_t)(n_sequences * 0.025 + 0.5);      // Hammock.java:398-401, the default cluster limit
    int64_t band = n_sequences >= 16384 ? std::min<int64_t>(n_sequences, 2 * maxc + 1024) : 0;
    if (band * 2 > (int64_t)n_sequences) band = 0;
    st = reserve_tail_buffers(ctx, n_sequences, true, (uint32_t)band, true, true);
    if (ctx->sw.greedy_timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[hmk] hmk_reserve(%u): waited for the context %.1f ms, device + streams %.1f ms, buffers %.1f ms\n", n_sequences,
                     ms(t_call, t_lock), ms(t_lock, t_streams), ms(t_streams, std::chrono::steady_clock::now()));
    }
    return st;
}

int hmk_set_java_hashset(hmk_ctx *ctx, int version) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    if (version != 8 && version != 7 && version != 6) return fail(ctx, HMK_ERR_BAD_ARG, "hmk_set_java_hashset: 8 (Java 8+), 7 (JDK 7u6+) or 6 (JDK 6 / 7 before 7u6)");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    ctx->java_hashset = version;
    for (hmk_ctx *peer : ctx->peers) peer->java_hashset = version;
    return HMK_OK;
}

int hmk_device_count(const hmk_ctx *ctx) { return ctx ? (ctx->has_device ? 1 + (int)ctx->peers.size() : 0) : 0; }

int hmk_greedy_from_edges_dev(hmk_ctx *ctx, const void *d_edges, uint64_t n_edges, int symmetric, int max_clusters,
                              int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    int st = need_device(ctx);
    if (st) return st;
    if (ctx->n && !cluster_id) return fail(ctx, HMK_ERR_BAD_ARG, "null cluster_id");
    if (n_edges && !d_edges) return fail(ctx, HMK_ERR_BAD_ARG, "null edge buffer");
    hmk_greedy_stats local;
    if (!stats) stats = &local;
    std::memset(stats, 0, sizeof(*stats));
    ctx->phases = hmk_greedy_phases{};
    if (ctx->n == 0) return HMK_OK;
    st = greedy_streams(ctx);
    if (st) return st;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t S = ctx->gstream;
    // the caller's block may have been written on any stream of its own (an all-gather, a copy): wait for the device
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, ensure_buf(ctx, SB_PEERCNT, HMK_MAX_SEGS * sizeof(unsigned long long)));
    ctx->h_counts[HC_PEER] = n_edges;
    HIPCHK(ctx, hipMemcpyAsync(buf<void>(ctx, SB_PEERCNT), &ctx->h_counts[HC_PEER], sizeof(unsigned long long), hipMemcpyHostToDevice, S));
    EdgeSource src;
    src.symmetric = symmetric != 0;
    src.segs.n = 1;
    src.segs.s[0] = EdgeSeg{(const uint64_t *)d_edges, buf<unsigned long long>(ctx, SB_PEERCNT), n_edges};
    src.total_known = 0;   // invalid edges are not counted by the degree pass; the adjacency size is checked against it
    HIPCHK(ctx, hipEventRecord(ctx->ev_t0, S));
    HIPCHK(ctx, hipEventRecord(ctx->ev_edges, S));
    st = cluster_on_device(ctx, src, max_clusters, cluster_id, result_order, member_rank, stats, t0);
    if (st == HMK_OK || st == HMK_ERR_REFERENCE_WOULD_CRASH) stats->n_edges = n_edges;
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

int hmk_clinkage_cluster(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    hmk_clinkage_stats local;
    if (!stats) stats = &local;
    std::memset(stats, 0, sizeof(*stats));
    ctx->phases = hmk_greedy_phases{};
    if (ctx->n == 0)
        return fail(ctx, HMK_ERR_REFERENCE_WOULD_CRASH,
                    "the reference throws NoSuchElementException here (ClinkageSequenceClusterer.java:118): empty input");
    if (!cluster_id) return fail(ctx, HMK_ERR_BAD_ARG, "null cluster_id");
    if (!ctx->symmetric)
        return fail(ctx, HMK_ERR_BAD_ARG, "clinkage needs a symmetric scoring matrix: the reference caches cluster scores by unordered "
                                          "pair (CachedClusterScorer.java:43-53), so its result depends on the evaluation order otherwise");
    int st = need_device(ctx);
    if (st) return st;
    st = greedy_streams(ctx);
    if (st) return st;
    if (!ctx->peers.empty())
        return greedy_cluster_multi(ctx, max_shift, shift_penalty, threshold, 0, cluster_id, result_order, member_rank, nullptr, stats);
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = ctx->n;
    unsigned long long counts[HMK_EDGE_SHARDS];
    double ms = 0;
    // clinkage inputs are small (the reference switches to greedy above 10,000 sequences) but dense: MUSI has 16 % of
    // its pairs above the default threshold for some rows; neighbors_internal grows the buffer until the pass fits
    const uint64_t guess = (uint64_t)((double)n * (n - 1) / 2 * 0.02) + (1u << 20);
    st = neighbors_internal(ctx, max_shift, shift_penalty, threshold, 0, 1, std::min<uint64_t>(guess, 1ull << 31), counts, &ms);
    if (st) return st;
    uint64_t total = 0;
    for (int q = 0; q < HMK_EDGE_SHARDS; q++) total += counts[q];
    hipStream_t S = ctx->gstream;
    const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                          (long long)std::max(0, shift_penalty) * ((ctx->max_len - ctx->min_len) + 2LL * max_shift);
    EdgeSource src;
    src.symmetric = true;
    src.segs = shard_segments(ctx->d_edges, ctx->d_edges_cap / HMK_EDGE_SHARDS, ctx->d_counts);
    src.format_known = true;
    src.packed = top - threshold <= 255 && !ctx->sw.adj_8byte;
    src.base = threshold;
    src.total_known = total;
    src.adj_bound = 2 * total;
    src.clink = stats;
    HIPCHK(ctx, hipEventRecord(ctx->ev_t0, S));
    HIPCHK(ctx, hipEventRecord(ctx->ev_edges, S));
    st = cluster_on_device(ctx, src, 0, cluster_id, result_order, member_rank, nullptr, t0);
    if (st == ST_RETRY_OVERFLOW) return fail(ctx, HMK_ERR_DEVICE, "internal edge buffer overflow");
    stats->neighbors_ms = ms;
    ctx->phases.score_ms = ms;
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

int hmk_clinkage_from_edges(hmk_ctx *ctx, const uint64_t *edges, uint64_t n_edges, int32_t *cluster_id, int32_t *result_order,
                            int32_t *member_rank, hmk_clinkage_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    refresh_switches(ctx);
    hmk_clinkage_stats local;
    if (!stats) stats = &local;
    std::memset(stats, 0, sizeof(*stats));
    const uint32_t n = ctx->n;
    if (n == 0)
        return fail(ctx, HMK_ERR_REFERENCE_WOULD_CRASH,
                    "the reference throws NoSuchElementException here (ClinkageSequenceClusterer.java:118): empty input");
    if (!cluster_id) return fail(ctx, HMK_ERR_BAD_ARG, "null cluster_id");
    if (n_edges && !edges) return fail(ctx, HMK_ERR_BAD_ARG, "null edge list");
    // symmetric CSR on the host: every edge under both ends
    std::vector<uint64_t> start((size_t)n + 1, 0);
    for (uint64_t e = 0; e < n_edges; e++) {
        const uint32_t x = HMK_EDGE_X(edges[e]), m = HMK_EDGE_M(edges[e]);
        if (x >= n || m >= n || x == m) return fail(ctx, HMK_ERR_BAD_ARG, "edge list references a sequence outside [0, n) or a self pair");
        start[x + 1]++;
        start[m + 1]++;
    }
    for (uint32_t k = 0; k < n; k++) start[k + 1] += start[k];
    std::vector<Nbr> adj(start[n]);
    {
        std::vector<uint64_t> fill(start.begin(), start.end() - 1);
        for (uint64_t e = 0; e < n_edges; e++) {
            const uint32_t x = HMK_EDGE_X(edges[e]), m = HMK_EDGE_M(edges[e]);
            const int32_t sc = HMK_EDGE_SCORE(edges[e]);
            adj[fill[x]++] = Nbr{m, sc};
            adj[fill[m]++] = Nbr{x, sc};
        }
    }
    std::string err;
    const int st = clinkage_from_csr(ctx->java_hashset, n, ctx->has_sizes ? ctx->sizes.data() : nullptr, start.data(), adj.data(), cluster_id, result_order,
                                     member_rank, stats, &err);
    stats->n_edges = n_edges;
    if (st) return fail(ctx, st, err);
    return HMK_OK;
}

int hmk_greedy_last_phases(const hmk_ctx *ctx, hmk_greedy_phases *out) {
    if (!ctx || !out) return fail(nullptr, HMK_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);   // the clustering calls write it under the same lock: never a torn struct
    *out = ctx->phases;
    return HMK_OK;
}

}  // extern "C"
