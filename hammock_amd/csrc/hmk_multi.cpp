// hmk_multi.cpp -- hmk_greedy_cluster / hmk_clinkage_cluster on a context of several devices (hmk_create_multi): one worker thread
// per peer, band first on every device, peer copies to the root, the usual tail on the root.
#include "hmk_ctx.h"

namespace hmk { namespace impl {

// hmk_greedy_cluster on a multi-device context: every device scores its row-block shard (no collective in the scoring),
// the peers' edges travel to the root over xGMI as direct peer copies, every peer over its own link to the root, and the
// root runs the usual tail on the union (CSR on the device, merge).
//
// One worker thread per peer plans, uploads and launches its shard (the plan alone is tens of milliseconds at 10^6) while the
// calling thread does the same for the root and then goes straight into cluster_on_device.  Every device launches the tiles
// that touch a band row first (LimitedGreedySequenceClusterer.java:77-120 reads the first rows only), compacts the band's
// edges into one block and ships it as soon as its own band launch is over; the root builds the band's adjacency from its own
// band segments + the peers' band blocks, and phase 1 runs on the host while every device is still scoring and the rest of
// the edges travel.  A peer's copies are ordered behind that peer's own events only; nothing waits for "all devices".
struct PeerJob {
    hmk_ctx *c = nullptr;
    uint32_t part = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int band_state = 0;      // 0 pending, 1 gathered (ev_bandgather recorded), -1 no band block from this peer
    int full_state = 0;      // 0 pending, 1 gathered (ev_gather recorded), -1 failed, -2 a segment overflowed
    int status = HMK_OK;
    std::string err;
    uint64_t total = 0, band_total = 0;
    uint64_t region = 0, band_region = 0;      // capacity of its blocks on the root (entries)
    uint64_t off = 0, band_off = 0;            // where they start inside SB_PEER / SB_PEERBAND
};

int greedy_cluster_multi(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int max_clusters, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats, hmk_clinkage_stats *clink) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t G = 1 + (uint32_t)ctx->peers.size();
    const uint32_t n = ctx->n;
    if (HMK_EDGE_SHARDS + G - 1 > HMK_MAX_SEGS) return fail(ctx, HMK_ERR_BAD_ARG, "too many devices for one context");
    hipStream_t S = ctx->gstream, C = ctx->copy_stream;
    // HMK_MULTI_SERIAL=1: the conservative form, for a machine where the overlapped one misbehaves -- no worker threads, no band, no
    // peer copies: the peers score their shards one after the other from the calling thread, each is synchronised, and its block
    // and row degrees travel through host memory (two plain hipMemcpy); then the root scores its shard and runs the same tail.
    const bool serial = ctx->sw.multi_serial;
    int64_t band_req = 0;
    if (!serial && !clink && max_clusters > 0 && n >= 16384 && !ctx->sw.no_band) band_req = std::min<int64_t>(n, 2LL * max_clusters + 1024);
    if (band_req * 2 > (int64_t)n) band_req = 0;
    uint64_t guess = (uint64_t)((double)n * (n - 1) / 2 * (ctx->symmetric ? 0.003 : 0.006) / G * 1.25) + (1u << 20);
    if (ctx->sw.edge_guess) guess = ctx->sw.edge_guess;   // tests: force the overflow / retry path
    const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                          (long long)std::max(0, shift_penalty) * ((ctx->max_len - ctx->min_len) + 2LL * max_shift);
    // every device counts the row degrees of the edges it writes (the CSR's first pass, fused into the scoring as in the
    // single-device call); the peers' counters travel with their blocks and are added to the root's
    const bool fuse = ctx->symmetric;
    int st = HMK_OK;
    for (int attempt = 0; attempt < 4; attempt++) {
        // ---- edge buffers (grown to the counts of the last attempt if a segment overflowed) and the root-side regions -----
        std::vector<std::unique_ptr<PeerJob>> jobs;
        for (uint32_t d = 0; d < G; d++) {
            hmk_ctx *c = d ? ctx->peers[d - 1] : ctx;
            st = need_device(c);
            if (st == HMK_OK) st = greedy_streams(c);
            if (st) return d ? fail(ctx, st, c->err) : st;
            if (!c->d_counts) HIPCHK(ctx, hipMalloc((void **)&c->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
            uint64_t cap = std::max<uint64_t>({std::min<uint64_t>(guess, 1ull << 31), (uint64_t)1 << 20, c->d_edges_cap});
            if (attempt > 0) {
                unsigned long long mx = 0;
                for (int q = 0; q < HMK_EDGE_SHARDS; q++) mx = std::max(mx, c->h_counts[q]);
                cap = std::max<uint64_t>(cap, (uint64_t)HMK_EDGE_SHARDS * (mx + mx / 8 + 1024));
            }
            cap = (cap + HMK_EDGE_SHARDS - 1) / HMK_EDGE_SHARDS * HMK_EDGE_SHARDS;
            if (c->d_edges_cap < cap) {
                if (c->d_edges) (void)hipFree(c->d_edges);
                c->d_edges = nullptr;
                c->d_edges_cap = 0;
                HIPCHK(ctx, hipMalloc((void **)&c->d_edges, cap * sizeof(uint64_t)));
                c->d_edges_cap = cap;
            }
            HIPCHK(ctx, ensure_buf(c, SB_BCOUNTS, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
            if (d) {   // the peer's own compacted blocks (band, everything) and their totals
                HIPCHK(ctx, ensure_buf(c, SB_PEER, c->d_edges_cap * sizeof(uint64_t)));
                HIPCHK(ctx, ensure_buf(c, SB_PEERBAND, (c->d_edges_cap / 2 + 1) * sizeof(uint64_t)));
                HIPCHK(ctx, ensure_buf(c, SB_PEERCNT, 64));
                if (fuse) HIPCHK(ctx, ensure_buf(c, SB_DEG, (size_t)n * 4));
                jobs.emplace_back(new PeerJob());
                PeerJob &J = *jobs.back();
                J.c = c;
                J.part = d;
                J.region = c->d_edges_cap;
                J.band_region = c->d_edges_cap / 2 + 1;
            }
        }
        st = need_device(ctx);
        if (st) return st;
        uint64_t off = 0, boff = 0;
        for (auto &jp : jobs) { jp->off = off; off += jp->region; jp->band_off = boff; boff += jp->band_region; }
        HIPCHK(ctx, ensure_buf(ctx, SB_PEER, std::max<uint64_t>(off, 1) * sizeof(uint64_t)));
        HIPCHK(ctx, ensure_buf(ctx, SB_PEERBAND, std::max<uint64_t>(boff, 1) * sizeof(uint64_t)));
        HIPCHK(ctx, ensure_buf(ctx, SB_PEERCNT, 2 * HMK_MAX_SEGS * sizeof(unsigned long long)));   // [d]: a peer's total, [HMK_MAX_SEGS + d]: its band total
        if (fuse) HIPCHK(ctx, ensure_buf(ctx, SB_PEERDEG, std::max<size_t>(jobs.size(), 1) * (size_t)n * 4));
        for (auto &jp : jobs) {   // root-side stream and events of the peer's transfers
            hmk_ctx *c = jp->c;
            if (!c->gather_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&c->gather_stream, hipStreamNonBlocking));
            if (!c->ev_bandgather) HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_bandgather, hipEventDisableTiming));
            if (!c->ev_gather) HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));
        }
        // everything the tail will want on the root, before anything is enqueued (a hipMalloc waits for running kernels)
        const bool packed = top - threshold <= 255 && !ctx->sw.adj_8byte;
        {
            uint64_t all_cap = ctx->d_edges_cap + off;
            const size_t esz0 = packed ? sizeof(NbrPacked) : sizeof(Nbr);
            HIPCHK(ctx, ensure_buf(ctx, SB_ADJ, std::max<uint64_t>((ctx->symmetric ? 2 : 1) * all_cap, 1) * esz0));
            HIPCHK(ctx, ensure_buf(ctx, SB_DEG, (size_t)n * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_CURSOR, (size_t)n * 8));
            HIPCHK(ctx, ensure_buf(ctx, SB_START, ((size_t)n + 1) * 8));
            HIPCHK(ctx, ensure_buf(ctx, SB_SCAN, scan_scratch_bytes(n)));
            HIPCHK(ctx, ensure_buf(ctx, SB_RANGE, 64));
            HIPCHK(ctx, ensure_pinned(&ctx->h_start, &ctx->h_start_cap, ((size_t)n + 1) * 8 + (size_t)n * 4 + 64, 0));
            if (csr_by_bucket(ctx->symmetric, packed)) {
                HIPCHK(ctx, ensure_buf(ctx, SB_PART, (all_cap + 1) * 8));
                HIPCHK(ctx, ensure_buf(ctx, SB_PARTSCR, csr_partition_scratch_bytes()));
            }
        }
        const int root_dev = ctx->device;
        uint64_t *root_peer = buf<uint64_t>(ctx, SB_PEER), *root_band = buf<uint64_t>(ctx, SB_PEERBAND);
        unsigned long long *root_cnt = buf<unsigned long long>(ctx, SB_PEERCNT);

        // ---- a peer's whole share: plan, band tiles, band block, the rest, the whole block; each hand-over as soon as it can go ----
        auto peer_body = [&](PeerJob &J) {
            hmk_ctx *c = J.c;
            auto set_band = [&](int v) { { std::lock_guard<std::mutex> l(J.mu); J.band_state = v; } J.cv.notify_all(); };
            auto set_full = [&](int v, int code, const std::string &msg) {
                { std::lock_guard<std::mutex> l(J.mu); J.full_state = v; J.status = code; J.err = msg; if (J.band_state == 0) J.band_state = -1; }
                J.cv.notify_all();
            };
            auto hip_fail = [&](const char *what, hipError_t e) {
                set_full(-1, e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
            };
            std::lock_guard<std::mutex> lock(c->mu);
            int r = need_device(c);
            if (r) { set_full(-1, r, c->err); return; }
            hipStream_t Q = c->gstream;
            const uint64_t seg = c->d_edges_cap / HMK_EDGE_SHARDS;
            unsigned long long *d_tot = buf<unsigned long long>(c, SB_PEERCNT);        // [0] everything, [1] the band
            unsigned long long *h_tot = c->h_counts + HC_PEER;                          // pinned: [0] everything, [1] the band
            hipError_t e = hipSuccess;
            r = build_plan(c, max_shift, shift_penalty, threshold, J.part, G, band_req);
            if (r) { set_full(-1, r, c->err); return; }
            const bool band = c->plan.band_rows > 0;
            uint32_t *p_deg = fuse ? buf<uint32_t>(c, SB_DEG) : nullptr;
            if (p_deg && (e = hipMemsetAsync(p_deg, 0, (size_t)n * 4, Q)) != hipSuccess) { hip_fail("degree counters", e); return; }
            if (band) {
                r = neighbors_dev_locked(c, max_shift, shift_penalty, threshold, J.part, G, c->d_edges, c->d_edges_cap, c->d_counts, Q, LAUNCH_BAND, band_req, p_deg);
                if (r) { set_full(-1, r, c->err); return; }
                e = hipMemcpyAsync(buf<void>(c, SB_BCOUNTS), c->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToDevice, Q);
                if (e == hipSuccess) e = launch_compact_edges(c->d_edges, seg, buf<unsigned long long>(c, SB_BCOUNTS), buf<uint64_t>(c, SB_PEERBAND), J.band_region, d_tot + 1, Q);
                if (e == hipSuccess) e = hipMemcpyAsync(h_tot + 1, d_tot + 1, 8, hipMemcpyDeviceToHost, Q);
                if (e == hipSuccess) e = hipEventRecord(c->ev_band, Q);
                if (e != hipSuccess) { hip_fail("band launch", e); return; }
            }
            r = neighbors_dev_locked(c, max_shift, shift_penalty, threshold, J.part, G, c->d_edges, c->d_edges_cap, c->d_counts, Q,
                                     band ? LAUNCH_REST : LAUNCH_ALL, band_req, p_deg);
            if (r) { (void)hipStreamSynchronize(Q); set_full(-1, r, c->err); return; }
            e = hipMemcpyAsync(c->h_counts, c->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, Q);
            if (e == hipSuccess) e = launch_compact_edges(c->d_edges, seg, c->d_counts, buf<uint64_t>(c, SB_PEER), J.region, d_tot, Q);
            if (e == hipSuccess) e = hipMemcpyAsync(h_tot, d_tot, 8, hipMemcpyDeviceToHost, Q);
            if (e == hipSuccess) e = hipEventRecord(c->ev_edges, Q);
            if (e != hipSuccess) { hip_fail("shard launch", e); return; }
            // -- band hand-over: its size is known once the band launch is over --
            if (band) {
                e = hipEventSynchronize(c->ev_band);
                if (e != hipSuccess) { hip_fail("band launch", e); return; }
                J.band_total = h_tot[1];
                if (J.band_total > J.band_region) set_band(-1);
                else {
                    h_tot[3] = J.band_total;
                    e = hipSetDevice(root_dev);
                    if (e == hipSuccess && J.band_total)
                        e = hipMemcpyPeerAsync(root_band + J.band_off, root_dev, buf<uint64_t>(c, SB_PEERBAND), c->device, J.band_total * sizeof(uint64_t), c->gather_stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(root_cnt + HMK_MAX_SEGS + J.part, h_tot + 3, 8, hipMemcpyHostToDevice, c->gather_stream);
                    if (e == hipSuccess) e = hipEventRecord(c->ev_bandgather, c->gather_stream);
                    if (e != hipSuccess) { hip_fail("band hand-over", e); return; }
                    set_band(1);
                    (void)hipSetDevice(c->device);
                }
            } else set_band(-1);
            // -- everything: once the shard is scored --
            e = hipEventSynchronize(c->ev_edges);
            if (e != hipSuccess) { hip_fail("shard", e); return; }
            for (int q = 0; q < HMK_EDGE_SHARDS; q++)
                if (c->h_counts[q] > seg) { set_full(-2, HMK_OK, ""); return; }
            J.total = h_tot[0];
            h_tot[2] = J.total;
            if (serial) {   // through the host, every step synchronous
                std::vector<uint64_t> h_block(std::max<uint64_t>(J.total, 1));
                std::vector<uint32_t> h_deg(p_deg ? n : 0);
                e = hipStreamSynchronize(Q);
                if (e == hipSuccess && J.total) e = hipMemcpy(h_block.data(), buf<uint64_t>(c, SB_PEER), J.total * sizeof(uint64_t), hipMemcpyDeviceToHost);
                if (e == hipSuccess && p_deg) e = hipMemcpy(h_deg.data(), p_deg, (size_t)n * 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipSetDevice(root_dev);
                if (e == hipSuccess && J.total) e = hipMemcpy(root_peer + J.off, h_block.data(), J.total * sizeof(uint64_t), hipMemcpyHostToDevice);
                if (e == hipSuccess && p_deg) e = hipMemcpy(buf<uint32_t>(ctx, SB_PEERDEG) + (size_t)(J.part - 1) * n, h_deg.data(), (size_t)n * 4, hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMemcpy(root_cnt + J.part, h_tot + 2, 8, hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipEventRecord(c->ev_gather, c->gather_stream);   // (nothing is pending on that stream: complete at once)
                (void)hipSetDevice(c->device);
                if (e != hipSuccess) { hip_fail("edge hand-over through the host", e); return; }
                set_full(1, HMK_OK, "");
                return;
            }
            e = hipSetDevice(root_dev);
            if (e == hipSuccess && J.total)
                e = hipMemcpyPeerAsync(root_peer + J.off, root_dev, buf<uint64_t>(c, SB_PEER), c->device, J.total * sizeof(uint64_t), c->gather_stream);
            if (e == hipSuccess && p_deg)
                e = hipMemcpyPeerAsync(buf<uint32_t>(ctx, SB_PEERDEG) + (size_t)(J.part - 1) * n, root_dev, p_deg, c->device, (size_t)n * 4, c->gather_stream);
            if (e == hipSuccess) e = hipMemcpyAsync(root_cnt + J.part, h_tot + 2, 8, hipMemcpyHostToDevice, c->gather_stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_gather, c->gather_stream);
            (void)hipSetDevice(c->device);
            if (e != hipSuccess) { hip_fail("edge hand-over", e); return; }
            set_full(1, HMK_OK, "");
        };
        // (an exception inside a worker -- bad_alloc from the plan, from a staging vector -- must end as this call's error, not
        // as std::terminate: the job is marked failed, which also releases whoever waits for its band or its edges)
        auto peer_main = [&](PeerJob &J) {
            try { peer_body(J); }
            catch (const std::bad_alloc &) { std::lock_guard<std::mutex> l(J.mu); J.full_state = -1; J.status = HMK_ERR_OOM; J.err = "out of host memory in a peer's worker"; if (J.band_state == 0) J.band_state = -1; J.cv.notify_all(); }
            catch (const std::exception &ex) { std::lock_guard<std::mutex> l(J.mu); J.full_state = -1; J.status = HMK_ERR_DEVICE; J.err = ex.what(); if (J.band_state == 0) J.band_state = -1; J.cv.notify_all(); }
        };
        struct Joiner {   // on every way out: the workers are done and nothing they enqueued is still in flight before their state goes away
            std::vector<std::unique_ptr<PeerJob>> &jobs;
            hipStream_t S, C;
            const hmk_ctx *root;
            ~Joiner() {
                for (auto &jp : jobs) if (jp->th.joinable()) jp->th.join();
                if (root->wedged) return;   // (a device that stopped making progress: nothing waits for it any more)
                for (auto &jp : jobs) if (jp->c->gather_stream) (void)hipStreamSynchronize(jp->c->gather_stream);   // peer copies into the root's blocks
                (void)hipStreamSynchronize(S);
                (void)hipStreamSynchronize(C);
                (void)hipGetLastError();
            }
        } joiner{jobs, S, C, ctx};   // (before the first thread starts: a std::thread constructor that throws leaves no joinable thread behind)
        if (serial) {
            for (auto &jp : jobs) peer_main(*jp);
            st = need_device(ctx);   // (back on the root's device)
            if (st) return st;
        } else {
            for (auto &jp : jobs) { PeerJob *J = jp.get(); J->th = std::thread([&peer_main, J]() { peer_main(*J); }); }
        }

        // ---- the root's own shard, on the calling thread ----------------------------------------------------------
        st = build_plan(ctx, max_shift, shift_penalty, threshold, 0, G, band_req);
        if (st) return st;
        const int64_t band_rows = ctx->plan.band_rows;
        ctx->phases.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const uint64_t seg0 = ctx->d_edges_cap / HMK_EDGE_SHARDS;
        uint32_t *r_deg = fuse ? buf<uint32_t>(ctx, SB_DEG) : nullptr;
        if (r_deg) HIPCHK(ctx, hipMemsetAsync(r_deg, 0, (size_t)n * 4, S));
        HIPCHK(ctx, hipEventRecord(ctx->ev_t0, S));
        if (band_rows > 0) {
            st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, G, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S, LAUNCH_BAND, band_req, r_deg);
            if (st) return st;
            HIPCHK(ctx, hipMemcpyAsync(buf<void>(ctx, SB_BCOUNTS), ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToDevice, S));
            HIPCHK(ctx, hipEventRecord(ctx->ev_band, S));
        }
        st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, G, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S,
                                  band_rows > 0 ? LAUNCH_REST : LAUNCH_ALL, band_req, r_deg);
        if (st) { (void)hipStreamSynchronize(S); return st; }
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, S));

        EdgeSource src;
        src.deg_fused = fuse;
        src.symmetric = ctx->symmetric;
        src.format_known = true;
        src.packed = packed;
        src.base = threshold;
        src.check_overflow = true;
        src.seg_cap = seg0;
        src.segs = shard_segments(ctx->d_edges, seg0, ctx->d_counts);
        src.band_segs = shard_segments(ctx->d_edges, seg0, buf<unsigned long long>(ctx, SB_BCOUNTS));
        src.band_rows = (uint32_t)band_rows;
        uint64_t bound = ctx->d_edges_cap;
        for (auto &jp : jobs) {
            src.segs.s[src.segs.n++] = EdgeSeg{root_peer + jp->off, root_cnt + jp->part, jp->region};
            src.band_segs.s[src.band_segs.n++] = EdgeSeg{root_band + jp->band_off, root_cnt + HMK_MAX_SEGS + jp->part, jp->band_region};
            bound += jp->region;
        }
        src.adj_bound = (ctx->symmetric ? 2 : 1) * bound;
        src.clink = clink;
        bool overflow = false;
        src.before_band = [&]() -> int {
            for (auto &jp : jobs) {
                std::unique_lock<std::mutex> l(jp->mu);
                jp->cv.wait(l, [&]() { return jp->band_state != 0; });
                if (jp->band_state < 0) return -1;
            }
            for (auto &jp : jobs)
                if (hipStreamWaitEvent(C, jp->c->ev_bandgather, 0) != hipSuccess) return -1;
            return HMK_OK;
        };
        src.before_full = [&]() -> int {
            int bad = HMK_OK;
            for (auto &jp : jobs) {
                std::unique_lock<std::mutex> l(jp->mu);
                jp->cv.wait(l, [&]() { return jp->full_state != 0; });
                if (jp->full_state == -2) overflow = true;
                else if (jp->full_state < 0 && bad == HMK_OK) { bad = jp->status ? jp->status : HMK_ERR_DEVICE; ctx->err = jp->err; }
            }
            if (bad) return bad;
            if (overflow) return ST_RETRY_OVERFLOW;
            for (auto &jp : jobs) {
                if (hipStreamWaitEvent(S, jp->c->ev_gather, 0) != hipSuccess) { ctx->err = "hipStreamWaitEvent (peer gather)"; return HMK_ERR_DEVICE; }
                if (fuse && launch_add_u32(buf<uint32_t>(ctx, SB_DEG), buf<uint32_t>(ctx, SB_PEERDEG) + (size_t)(jp->part - 1) * n, n, S) != hipSuccess) {
                    ctx->err = "adding a peer's row degrees";
                    return HMK_ERR_DEVICE;
                }
            }
            if (hipEventRecord(ctx->ev_edges, S) != hipSuccess) { ctx->err = "hipEventRecord"; return HMK_ERR_DEVICE; }
            return HMK_OK;
        };
        st = cluster_on_device(ctx, src, max_clusters, cluster_id, result_order, member_rank, stats, t0);
        for (auto &jp : jobs) if (jp->th.joinable()) jp->th.join();
        for (auto &jp : jobs) {   // a crash-parity exit during phase 1 never reached before_full
            if (jp->full_state == -2) overflow = true;
            (void)hipStreamSynchronize(jp->c->gather_stream);
        }
        if (st == ST_RETRY_OVERFLOW || (overflow && (st == HMK_OK || st == HMK_ERR_REFERENCE_WOULD_CRASH))) { st = ST_RETRY_OVERFLOW; continue; }
        break;
    }
    if (st == ST_RETRY_OVERFLOW) return fail(ctx, HMK_ERR_DEVICE, "internal edge buffer kept overflowing");
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_edges) == hipSuccess) ctx->phases.score_ms = ms;   // root shard + gather
    if (hipEventElapsedTime(&ms, ctx->ev_edges, ctx->ev_csr) == hipSuccess) ctx->phases.csr_ms = ms;
    (void)hipGetLastError();   // (a call that left early never recorded these events: "invalid resource handle" must not stay behind as the thread's last error)
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) stats->neighbors_ms = ctx->phases.score_ms;
    if (clink) clink->neighbors_ms = ctx->phases.score_ms;
    return st;
}

} }  // namespace hmk::impl
