// hmk_multi.cpp -- hmk_greedy_cluster / hmk_clinkage_cluster on a context of several devices (hmk_create_multi): every device scores a
// row-block shard of the pair space AND builds the adjacency of the rows it owns; only the band of phase 1, the candidate lists of the
// second loop and the joiners' rows ever reach the root.
#include "hmk_ctx.h"

namespace hmk { namespace impl {

// Device d of G owns the rows [d * rows_per, (d + 1) * rows_per).  One worker thread per device (the root's included; the calling thread
// keeps the host's part: band hand-over, phase 1, second loop):
//
//   plan -> BAND tiles -> (band block to the root) -> REST of its shard, degrees counted -> edges dealt into one block per owning device
//   -> blocks + degree slices to their owners, device to device, all pairs at once (2 / G of a shard per link, where gathering at a
//      root put every shard on the root's links)
//   -> CSR of its own rows from the blocks it received (k_lower_* with a row range)
//   -> after phase 1: the second loop's pre-check for the leftovers whose rows it holds; candidate entries to the root
//
// The root then runs the second loop (LimitedGreedySequenceClusterer.java:59-66) over all candidate lists and reads a joiner's row in
// the memory of the device that holds it (peer access; a copy of the piece where there is none).
// Every cross-device dependency goes through the HOST: a worker synchronises the stream its copies travel on and then publishes a state
// under one mutex; nobody waits on another device's event.  Every wait is for a state that always leaves 0 (a worker's exit sets what is
// still pending to "failed").
struct DevJob {
    hmk_ctx *c = nullptr;
    uint32_t d = 0, r0 = 0, r1 = 0;
    std::thread th;
    // states: 0 pending, 1 done, < 0 failed (-2: a buffer was too small -- grown for the next attempt)
    int band_state = 0;    // the band launch is enqueued (root) / the band block has landed on the root (peers); -1: no band from this device
    int sent_state = 0;    // its blocks and degree slices have landed on every other device
    int csr_state = 0;     // its piece of the CSR is enqueued (c->ev_csr recorded)
    int pre_state = 0;     // its part of the pre-check is done and on the root; -1: not usable (the host's pre-check)
    int status = HMK_OK;
    std::string err;
    uint64_t edges = 0, band_total = 0, band_region = 0, band_off = 0;
    unsigned long long pre_total = 0;
    uint64_t need_edges = 0, need_inbox = 0;   // (-2) what the next attempt must hold
    double t_band = 0, t_scored = 0, t_sent = 0, t_csr = 0, t_pre = 0;   // HMK_GREEDY_TIMING: ms since the call began
    unsigned long long route_off[HMK_MAX_DEVICES + 1] = {0};   // (valid once sent_state is 1) where its block for owner t begins in its SB_ROUTE
};

int greedy_cluster_multi(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int max_clusters, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats, hmk_clinkage_stats *clink) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t G = 1 + (uint32_t)ctx->peers.size();
    const uint32_t n = ctx->n;
    if (G > HMK_MAX_DEVICES || HMK_EDGE_SHARDS + G - 1 > HMK_MAX_SEGS) return fail(ctx, HMK_ERR_BAD_ARG, "too many devices for one context");
    const uint32_t rows_per = (n + G - 1) / G;
    hipStream_t S = ctx->gstream;
    auto dev = [&](uint32_t d) { return d ? ctx->peers[d - 1] : ctx; };
    int64_t band_req = 0;
    if (!clink && max_clusters > 0 && n >= 16384 && !ctx->sw.no_band) band_req = std::min<int64_t>(n, 2LL * max_clusters + 1024);
    if (band_req * 2 > (int64_t)n) band_req = 0;
    uint64_t guess = (uint64_t)((double)n * (n - 1) / 2 * (ctx->symmetric ? 0.003 : 0.006) / G * 1.25) + (1u << 20);
    if (ctx->sw.edge_guess) guess = ctx->sw.edge_guess;   // tests: force the overflow / retry path
    const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                          (long long)std::max(0, shift_penalty) * ((ctx->max_len - ctx->min_len) + 2LL * max_shift);
    const bool packed = top - threshold <= 255 && !ctx->sw.adj_8byte;
    const bool symmetric = ctx->symmetric;
    const bool fuse = symmetric;   // the pass counts upper and lower degrees itself (asymmetric scores: a counting pass over the blocks)
    std::vector<uint64_t> want_edges(G, 0), want_inbox(G, 0);   // grown by an attempt that overflowed
    int st = HMK_OK;
    for (int attempt = 0; attempt < 4; attempt++) {
        // ---- every buffer of every device, before anything is enqueued (a hipMalloc waits for running kernels) ------------------
        std::vector<std::unique_ptr<DevJob>> jobs;
        std::vector<uint64_t> inbox_cap(G, 0);   // entries a device's inbox holds PER SENDER
        for (uint32_t d = 0; d < G; d++) {
            hmk_ctx *c = dev(d);
            st = need_device(c);
            if (st == HMK_OK) st = greedy_streams(c);
            if (st) return d ? fail(ctx, st, c->err) : st;
            c->sw = ctx->sw;
            if (!c->d_counts) HIPCHK(ctx, hipMalloc((void **)&c->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
            uint64_t cap = std::max<uint64_t>({std::min<uint64_t>(guess, 1ull << 31), (uint64_t)1 << 20, c->d_edges_cap, want_edges[d]});
            cap = (cap + HMK_EDGE_SHARDS - 1) / HMK_EDGE_SHARDS * HMK_EDGE_SHARDS;
            if (c->d_edges_cap < cap) {
                if (c->d_edges) (void)hipFree(c->d_edges);
                c->d_edges = nullptr;
                c->d_edges_cap = 0;
                HIPCHK(ctx, hipMalloc((void **)&c->d_edges, cap * sizeof(uint64_t)));
                c->d_edges_cap = cap;
            }
            if (!c->xfer_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&c->xfer_stream, hipStreamNonBlocking));
            HIPCHK(ctx, ensure_buf(c, SB_BCOUNTS, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
            HIPCHK(ctx, ensure_buf(c, SB_DEG, (size_t)n * 8));
            // an edge goes to the owner of each end: at most two blocks hold it
            HIPCHK(ctx, ensure_buf(c, SB_ROUTE, 2 * c->d_edges_cap * sizeof(uint64_t)));
            HIPCHK(ctx, ensure_buf(c, SB_ROUTECNT, 3 * (HMK_MAX_DEVICES + 1) * sizeof(unsigned long long)));
            // what one sender deals to one owner: 2 / G of its shard where the rows are spread evenly, + a half
            inbox_cap[d] = std::max<uint64_t>({(uint64_t)((double)c->d_edges_cap * 2.0 / G * 1.5) + 65536, want_inbox[d]});
            jobs.emplace_back(new DevJob());
            DevJob &J = *jobs.back();
            J.c = c;
            J.d = d;
            J.r0 = (uint32_t)std::min<uint64_t>(n, (uint64_t)d * rows_per);
            J.r1 = (uint32_t)std::min<uint64_t>(n, (uint64_t)(d + 1) * rows_per);
            if (d) {   // the peer's own compacted band block
                HIPCHK(ctx, ensure_buf(c, SB_PEERBAND, (c->d_edges_cap / 2 + 1) * sizeof(uint64_t)));
                J.band_region = c->d_edges_cap / 2 + 1;
            }
        }
        for (uint32_t d = 0; d < G; d++) {   // (the inboxes: every device's capacity is known now)
            hmk_ctx *c = dev(d);
            st = need_device(c);
            if (st) return d ? fail(ctx, st, c->err) : st;
            HIPCHK(ctx, ensure_buf(c, SB_PEER, std::max<uint64_t>((G - 1) * inbox_cap[d], 1) * sizeof(uint64_t)));
            HIPCHK(ctx, ensure_buf(c, SB_PEERCNT, 2 * HMK_MAX_SEGS * sizeof(unsigned long long)));   // [s]: sender s's block size; (root) [HMK_MAX_SEGS + s]: its band block's
            HIPCHK(ctx, ensure_buf(c, SB_PEERDEG, std::max<size_t>(G - 1, 1) * 2 * (size_t)rows_per * 4));
            // the piece's CSR and pre-check buffers (grow-only: steady-state calls find them in place)
            const uint64_t records = 2 * c->d_edges_cap + (G - 1) * inbox_cap[d] + 1;
            const size_t esz0 = packed ? sizeof(NbrPacked) : sizeof(Nbr);
            HIPCHK(ctx, ensure_buf(c, SB_ADJ, (symmetric ? 2 : 1) * records * esz0));
            HIPCHK(ctx, ensure_buf(c, SB_CURSOR, (size_t)n * 8));
            HIPCHK(ctx, ensure_buf(c, SB_START, ((size_t)n + 1) * 8));
            HIPCHK(ctx, ensure_buf(c, SB_SCAN, scan_scratch_bytes(n)));
            HIPCHK(ctx, ensure_buf(c, SB_RANGE, 64));
            if (csr_by_bucket(symmetric, packed)) {
                HIPCHK(ctx, ensure_buf(c, SB_PART, records * 8));
                HIPCHK(ctx, ensure_buf(c, SB_PARTSCR, csr_partition_scratch_bytes()));
            }
            HIPCHK(ctx, ensure_buf(c, SB_COF, (size_t)n * 4));
            HIPCHK(ctx, ensure_buf(c, SB_BITMAP, ((size_t)n + 31) / 32 * 4));
            HIPCHK(ctx, ensure_buf(c, SB_LEFT, (size_t)n * 4));
            HIPCHK(ctx, ensure_buf(c, SB_CNT, (size_t)n * 4));
            HIPCHK(ctx, ensure_buf(c, SB_CSTART, ((size_t)n + 1) * 4));
            HIPCHK(ctx, ensure_buf(c, SB_CAND, (size_t)n * 24 * sizeof(GreedyCand)));
            HIPCHK(ctx, ensure_buf(c, SB_RETRY, (size_t)n * 4));
        }
        st = need_device(ctx);
        if (st) return st;
        uint64_t boff = 0;
        for (auto &jp : jobs) { jp->band_off = boff; boff += jp->band_region; }
        HIPCHK(ctx, ensure_buf(ctx, SB_PEERBAND, std::max<uint64_t>(boff, 1) * sizeof(uint64_t)));
        HIPCHK(ctx, ensure_pinned(&ctx->h_start, &ctx->h_start_cap, ((size_t)n + 1) * 8 + (size_t)n * 4 + 64, 0));
        uint64_t *root_band = buf<uint64_t>(ctx, SB_PEERBAND);
        unsigned long long *root_cnt = buf<unsigned long long>(ctx, SB_PEERCNT);

        // ---- shared state -----------------------------------------------------------------------------------------------------
        std::mutex mu;
        std::condition_variable cv;
        int pre_request = 0;          // 0: phase 1 is still running, 1: run the pre-check (pre_in), -1: this call has none
        PreIn pre_in;
        auto set_state = [&](int DevJob::*field, DevJob &J, int v) { { std::lock_guard<std::mutex> l(mu); J.*field = v; } cv.notify_all(); };
        auto fail_job = [&](DevJob &J, int code, const std::string &msg) {
            std::lock_guard<std::mutex> l(mu);
            if (J.status == HMK_OK) { J.status = code; J.err = msg; }
        };
        auto wait_all = [&](int DevJob::*field) {   // until that state of every device has left 0; -> the worst of them
            std::unique_lock<std::mutex> l(mu);
            int worst = 1;
            for (auto &jp : jobs) {
                DevJob *J = jp.get();
                cv.wait(l, [&]() { return J->*field != 0; });
                worst = std::min(worst, J->*field);
            }
            return worst;
        };

        // ---- one device's whole share --------------------------------------------------------------------------------------------
        auto worker_body = [&](DevJob &J) {
            hmk_ctx *c = J.c;
            const uint32_t d = J.d;
            auto hip_fail = [&](const char *what, hipError_t e) {
                fail_job(J, e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
            };
            int r = need_device(c);
            if (r) { fail_job(J, r, c->err); return; }
            hipStream_t Q = c->gstream, X = c->xfer_stream;
            const uint64_t seg = c->d_edges_cap / HMK_EDGE_SHARDS;
            unsigned long long *d_rcnt = buf<unsigned long long>(c, SB_ROUTECNT), *d_roff = d_rcnt + HMK_MAX_DEVICES + 1, *d_rcur = d_roff + HMK_MAX_DEVICES + 1;
            unsigned long long *h_rcnt = c->h_counts + HC_PEER;                          // pinned: the G block sizes, then [16] the band block's
            uint64_t *d_route = buf<uint64_t>(c, SB_ROUTE);
            hipError_t e = hipSuccess;
            auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
            r = build_plan(c, max_shift, shift_penalty, threshold, d, G, band_req);
            if (r) { fail_job(J, r, c->err); return; }
            if (d == 0) ctx->phases.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const bool band = c->plan.band_rows > 0;
            uint32_t *p_deg = fuse ? buf<uint32_t>(c, SB_DEG) : nullptr, *p_deg_lo = fuse ? p_deg + n : nullptr;
            if (p_deg && (e = hipMemsetAsync(p_deg, 0, (size_t)n * 8, Q)) != hipSuccess) { hip_fail("degree counters", e); return; }
            if (d == 0 && (e = hipEventRecord(c->ev_t0, Q)) != hipSuccess) { hip_fail("hipEventRecord", e); return; }
            if (band) {
                r = neighbors_dev_locked(c, max_shift, shift_penalty, threshold, d, G, c->d_edges, c->d_edges_cap, c->d_counts, Q, LAUNCH_BAND, band_req, p_deg, p_deg_lo);
                if (r) { fail_job(J, r, c->err); return; }
                e = hipMemcpyAsync(buf<void>(c, SB_BCOUNTS), c->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToDevice, Q);
                if (e == hipSuccess && d) e = launch_compact_edges(c->d_edges, seg, buf<unsigned long long>(c, SB_BCOUNTS), buf<uint64_t>(c, SB_PEERBAND), J.band_region, d_rcnt + HMK_MAX_DEVICES, Q);
                if (e == hipSuccess && d) e = hipMemcpyAsync(h_rcnt + 16, d_rcnt + HMK_MAX_DEVICES, 8, hipMemcpyDeviceToHost, Q);
                if (e == hipSuccess) e = hipEventRecord(c->ev_band, Q);
                if (e != hipSuccess) { hip_fail("band launch", e); return; }
                if (d == 0) set_state(&DevJob::band_state, J, 1);   // (the root's own band segments: its copy stream waits for ev_band itself)
            } else if (d == 0) set_state(&DevJob::band_state, J, -1);
            r = neighbors_dev_locked(c, max_shift, shift_penalty, threshold, d, G, c->d_edges, c->d_edges_cap, c->d_counts, Q,
                                     band ? LAUNCH_REST : LAUNCH_ALL, band_req, p_deg, p_deg_lo);
            if (r) { (void)hipStreamSynchronize(Q); fail_job(J, r, c->err); return; }
            e = hipMemcpyAsync(c->h_counts, c->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, Q);
            // the shard's edges, one block per owning device
            if (e == hipSuccess) e = launch_route_edges(shard_segments(c->d_edges, seg, c->d_counts), rows_per, G, d_rcnt, d_roff, d_rcur, d_route, 2 * c->d_edges_cap, Q);
            if (e == hipSuccess) e = hipMemcpyAsync(h_rcnt, d_rcnt, G * sizeof(unsigned long long), hipMemcpyDeviceToHost, Q);
            if (e == hipSuccess) e = hipEventRecord(c->ev_edges, Q);
            if (e != hipSuccess) { hip_fail("shard launch", e); return; }
            // -- band hand-over (peers): its size is known once the band launch is over --
            if (d) {
                if (band) {
                    e = hipEventSynchronize(c->ev_band);
                    if (e != hipSuccess) { hip_fail("band launch", e); return; }
                    J.band_total = h_rcnt[16];
                    if (J.band_total > J.band_region) set_state(&DevJob::band_state, J, -1);
                    else {
                        if (c->device != ctx->device) {   // (a context on the root's own device is read where its block lies: below, band_segs)
                            if (J.band_total)
                                e = hipMemcpyPeerAsync(root_band + J.band_off, ctx->device, buf<uint64_t>(c, SB_PEERBAND), c->device, J.band_total * sizeof(uint64_t), X);
                            if (e == hipSuccess) e = hipMemcpyPeerAsync(root_cnt + HMK_MAX_SEGS + d, ctx->device, d_rcnt + HMK_MAX_DEVICES, c->device, 8, X);
                            if (e == hipSuccess) e = hipStreamSynchronize(X);   // (landed: the root's copy stream needs no event of another device)
                        }
                        if (e != hipSuccess) { hip_fail("band hand-over", e); return; }
                        J.t_band = now_ms();
                        set_state(&DevJob::band_state, J, 1);
                    }
                } else set_state(&DevJob::band_state, J, -1);
            }
            // -- the blocks to their owners: once the shard is scored and dealt --
            e = hipEventSynchronize(c->ev_edges);
            if (e != hipSuccess) { hip_fail("shard", e); return; }
            J.t_scored = now_ms();
            {
                unsigned long long mx = 0;
                J.edges = 0;
                for (int q = 0; q < HMK_EDGE_SHARDS; q++) { mx = std::max(mx, c->h_counts[q]); J.edges += std::min<unsigned long long>(c->h_counts[q], seg); }
                if (mx > seg) { J.need_edges = (uint64_t)HMK_EDGE_SHARDS * (mx + mx / 8 + 1024); set_state(&DevJob::sent_state, J, -2); return; }
            }
            {
                bool too_small = false;
                for (uint32_t t = 0; t < G; t++)
                    if (t != d && dev(t)->device != c->device && h_rcnt[t] > inbox_cap[t]) {   // (the owner's inbox is too small for this block: grown for the next attempt)
                        std::lock_guard<std::mutex> l(mu);
                        jobs[t]->need_inbox = std::max<uint64_t>(jobs[t]->need_inbox, h_rcnt[t] + h_rcnt[t] / 8 + 65536);
                        too_small = true;
                    }
                if (too_small) { set_state(&DevJob::sent_state, J, -2); return; }
            }
            std::vector<unsigned long long> off(G + 1, 0);
            for (uint32_t t = 0; t < G; t++) { off[t + 1] = off[t] + h_rcnt[t]; J.route_off[t + 1] = off[t + 1]; }
            for (uint32_t t = 0; t < G && e == hipSuccess; t++) {
                if (t == d) continue;
                hmk_ctx *o = dev(t);
                const uint32_t slot = d < t ? d : d - 1;   // this sender's place among the owner's G - 1 senders
                const uint32_t o_r0 = jobs[t]->r0, o_len = jobs[t]->r1 - jobs[t]->r0;
                // (an owner on this very device -- a device list that names a GPU several times -- reads the block where it lies: a copy
                // inside one device is a blit kernel that waits for workgroup slots beside every context's pass)
                if (o->device != c->device) {
                    if (h_rcnt[t])
                        e = hipMemcpyPeerAsync(buf<uint64_t>(o, SB_PEER) + (uint64_t)slot * inbox_cap[t], o->device, d_route + off[t], c->device, h_rcnt[t] * sizeof(uint64_t), X);
                    if (e == hipSuccess) e = hipMemcpyPeerAsync(buf<unsigned long long>(o, SB_PEERCNT) + slot, o->device, d_rcnt + t, c->device, 8, X);
                }
                if (e == hipSuccess && p_deg && o_len) {
                    uint32_t *slice = buf<uint32_t>(o, SB_PEERDEG) + (size_t)slot * 2 * rows_per;
                    e = hipMemcpyPeerAsync(slice, o->device, p_deg + o_r0, c->device, (size_t)o_len * 4, X);
                    if (e == hipSuccess) e = hipMemcpyPeerAsync(slice + o_len, o->device, p_deg_lo + o_r0, c->device, (size_t)o_len * 4, X);
                }
            }
            if (e == hipSuccess) e = hipStreamSynchronize(X);
            if (e != hipSuccess) { hip_fail("edge blocks to their owners", e); return; }
            J.t_sent = now_ms();
            set_state(&DevJob::sent_state, J, 1);
            // -- its piece of the CSR: the rows [r0, r1) from its own block and the G - 1 it received --
            if (wait_all(&DevJob::sent_state) < 0) return;
            {
                EdgeSegs in{};
                in.s[in.n++] = EdgeSeg{d_route + off[d], d_rcnt + d, h_rcnt[d]};
                const uint32_t *slices[HMK_MAX_DEVICES] = {nullptr};
                for (uint32_t k = 0; k + 1 < G; k++) {
                    const uint32_t sender = k < d ? k : k + 1;
                    hmk_ctx *sc = dev(sender);
                    if (sc->device == c->device) {   // (its block, where the sender dealt it)
                        const DevJob &SJ = *jobs[sender];
                        in.s[in.n++] = EdgeSeg{buf<uint64_t>(sc, SB_ROUTE) + SJ.route_off[d], buf<unsigned long long>(sc, SB_ROUTECNT) + d, SJ.route_off[d + 1] - SJ.route_off[d]};
                    } else
                        in.s[in.n++] = EdgeSeg{buf<uint64_t>(c, SB_PEER) + (uint64_t)k * inbox_cap[d], buf<unsigned long long>(c, SB_PEERCNT) + k, inbox_cap[d]};
                    slices[k] = buf<uint32_t>(c, SB_PEERDEG) + (size_t)k * 2 * rows_per;
                }
                if (p_deg) e = launch_owned_degrees(p_deg, n, J.r0, J.r1, slices, G - 1, Q);
                if (e == hipSuccess) e = piece_enqueue_csr(c, in, symmetric, packed, threshold, n, J.r0, J.r1, fuse, fuse, Q);
                if (e != hipSuccess) { hip_fail("CSR piece", e); return; }
                set_state(&DevJob::csr_state, J, 1);
                if (ctx->sw.greedy_timing && hipEventSynchronize(c->ev_csr) == hipSuccess) J.t_csr = now_ms();
            }
            // -- the second loop's pre-check for the leftovers whose rows live here, once phase 1 is over on the host --
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&]() { return pre_request != 0; });
                if (pre_request < 0) return;
            }
            const uint32_t per = HMK_PRE_REGIONS / G, rb = d * per, rc = d + 1 == G ? HMK_PRE_REGIONS - rb : per;
            const int fit = piece_precheck(c, pre_in, J.r0, J.r1, rb, rc, Q, true, &J.pre_total);
            if (fit != 0) { set_state(&DevJob::pre_state, J, -1); return; }
            if (d) {   // its regions of the candidate buffer and its leftovers' (first entry, count) to the root
                e = hipMemcpyPeerAsync(buf<GreedyCand>(ctx, SB_CAND) + (size_t)rb * pre_in.region_cap, ctx->device,
                                       buf<GreedyCand>(c, SB_CAND) + (size_t)rb * pre_in.region_cap, c->device, (size_t)rc * pre_in.region_cap * sizeof(GreedyCand), X);
                // the leftover list is [orphans | the sequences phase 1 never reached], ids ascending in each part: its leftovers are (at most)
                // one run of each
                const uint32_t *left = (const uint32_t *)(pre_in.h_block + pre_in.b_cof + pre_in.b_us);
                uint32_t split = 0;   // first index of the second part: where the ids stop ascending
                while (split + 1 < pre_in.nl && left[split] < left[split + 1]) split++;
                split = pre_in.nl ? split + 1 : 0;
                for (int part = 0; part < 2 && e == hipSuccess; part++) {
                    const uint32_t *pb = left + (part ? split : 0), *pe = left + (part ? pre_in.nl : split);
                    const uint32_t q0 = (uint32_t)(std::lower_bound(pb, pe, J.r0) - left), q1 = (uint32_t)(std::lower_bound(pb, pe, J.r1) - left);
                    if (q1 <= q0) continue;
                    e = hipMemcpyPeerAsync(buf<uint32_t>(ctx, SB_CNT) + q0, ctx->device, buf<uint32_t>(c, SB_CNT) + q0, c->device, (size_t)(q1 - q0) * 4, X);
                    if (e == hipSuccess)
                        e = hipMemcpyPeerAsync(buf<uint32_t>(ctx, SB_CSTART) + q0, ctx->device, buf<uint32_t>(c, SB_CSTART) + q0, c->device, (size_t)(q1 - q0) * 4, X);
                }
                if (e == hipSuccess) e = hipStreamSynchronize(X);
                if (e != hipSuccess) { hip_fail("candidate lists to the root", e); set_state(&DevJob::pre_state, J, -1); return; }
            }
            J.t_pre = now_ms();
            set_state(&DevJob::pre_state, J, 1);
        };
        // (an exception inside a worker -- bad_alloc from the plan -- must end as this call's error, not as std::terminate; whatever
        // way a worker leaves, every state of its job that is still pending becomes "failed", which releases whoever waits for it)
        auto worker = [&](DevJob &J) {
            try { worker_body(J); }
            catch (const std::bad_alloc &) { fail_job(J, HMK_ERR_OOM, "out of host memory in a device's worker"); }
            catch (const std::exception &ex) { fail_job(J, HMK_ERR_DEVICE, ex.what()); }
            {
                std::lock_guard<std::mutex> l(mu);
                for (int DevJob::*f : {&DevJob::band_state, &DevJob::sent_state, &DevJob::csr_state, &DevJob::pre_state})
                    if (J.*f == 0) J.*f = -1;
            }
            cv.notify_all();
        };
        struct Joiner {   // on every way out: the pre-check request is settled, the workers are done and nothing they enqueued is still in flight
            std::vector<std::unique_ptr<DevJob>> &jobs;
            std::mutex &mu; std::condition_variable &cv; int &pre_request;
            const hmk_ctx *root;
            ~Joiner() {
                { std::lock_guard<std::mutex> l(mu); if (pre_request == 0) pre_request = -1; }
                cv.notify_all();
                for (auto &jp : jobs) if (jp->th.joinable()) jp->th.join();
                if (root->wedged) return;   // (a device that stopped making progress: nothing waits for it any more)
                for (auto &jp : jobs) {
                    if (jp->c->xfer_stream) (void)hipStreamSynchronize(jp->c->xfer_stream);
                    if (jp->c->gstream) (void)hipStreamSynchronize(jp->c->gstream);
                    if (jp->c->copy_stream) (void)hipStreamSynchronize(jp->c->copy_stream);
                }
                (void)hipGetLastError();
            }
        } joiner{jobs, mu, cv, pre_request, ctx};   // (before the first thread starts: a std::thread constructor that throws leaves no joinable thread behind)
        for (auto &jp : jobs) { DevJob *J = jp.get(); J->th = std::thread([&worker, J]() { worker(*J); }); }

        // ---- the host's part, on the calling thread -----------------------------------------------------------------------------
        auto first_error = [&]() -> int {   // the first failed job's error becomes the call's
            std::lock_guard<std::mutex> l(mu);
            for (auto &jp : jobs)
                if (jp->status != HMK_OK) { ctx->err = jp->err; g_last_error = jp->err; return jp->status; }
            return HMK_OK;
        };
        EdgeSource src;
        src.symmetric = symmetric;
        src.format_known = true;
        src.packed = packed;
        src.base = threshold;
        src.deg_fused = fuse;
        src.deg_split = fuse;
        src.rows_per = rows_per;
        for (auto &jp : jobs) src.pieces.push_back(EdgeSource::Piece{jp->c, jp->r0, jp->r1});
        src.seg_cap = ctx->d_edges_cap / HMK_EDGE_SHARDS;
        src.band_segs = shard_segments(ctx->d_edges, src.seg_cap, buf<unsigned long long>(ctx, SB_BCOUNTS));
        for (uint32_t d = 1; d < G; d++) {
            hmk_ctx *pc = dev(d);
            if (pc->device == ctx->device)   // (a context on the root's own device: its compacted band block and that block's size, where they lie)
                src.band_segs.s[src.band_segs.n++] = EdgeSeg{buf<uint64_t>(pc, SB_PEERBAND), buf<unsigned long long>(pc, SB_ROUTECNT) + HMK_MAX_DEVICES, jobs[d]->band_region};
            else
                src.band_segs.s[src.band_segs.n++] = EdgeSeg{root_band + jobs[d]->band_off, root_cnt + HMK_MAX_SEGS + d, jobs[d]->band_region};
        }
        src.band_rows = (uint32_t)band_req;
        src.clink = clink;
        src.before_band = [&]() -> int { return wait_all(&DevJob::band_state) > 0 ? HMK_OK : -1; };
        bool overflow = false;
        src.before_full = [&]() -> int {
            const int worst = std::min(wait_all(&DevJob::sent_state), wait_all(&DevJob::csr_state));
            {
                std::lock_guard<std::mutex> l(mu);
                for (auto &jp : jobs) if (jp->sent_state == -2) overflow = true;
            }
            const int err = first_error();
            if (err) return err;
            if (overflow) return ST_RETRY_OVERFLOW;
            if (worst < 0) { ctx->err = "a device's worker stopped without a reason"; return HMK_ERR_DEVICE; }
            src.total_edges = 0;
            for (auto &jp : jobs) {
                if (hipEventSynchronize(jp->c->ev_csr) != hipSuccess) { ctx->err = "a piece of the CSR failed on its device"; (void)hipGetLastError(); return HMK_ERR_DEVICE; }
                src.total_edges += jp->edges;
            }
            (void)hipEventRecord(ctx->ev_edges, S);   // (timing: scoring, exchange and the pieces end here for the root)
            return HMK_OK;
        };
        src.precheck_pieces = [&](const PreIn &in, unsigned long long *total) -> bool {
            { std::lock_guard<std::mutex> l(mu); pre_in = in; pre_request = 1; }
            cv.notify_all();
            const int worst = wait_all(&DevJob::pre_state);
            *total = 0;
            for (auto &jp : jobs) *total += jp->pre_total;
            return worst > 0;
        };
        st = cluster_on_device(ctx, src, max_clusters, cluster_id, result_order, member_rank, stats, t0);
        {
            std::lock_guard<std::mutex> l(mu);
            if (pre_request == 0) pre_request = -1;   // (a crash-parity exit during phase 1, the host's own loop: the workers go home)
        }
        cv.notify_all();
        for (auto &jp : jobs) if (jp->th.joinable()) jp->th.join();
        if (ctx->sw.greedy_timing)
            for (auto &jp : jobs)
                fprintf(stderr, "[hmk greedy] device %u of %u (HIP device %d, rows %u..%u): band block on the root at %.2f ms, shard scored and dealt at %.2f, blocks on their "
                                "owners at %.2f, its CSR piece done at %.2f (%llu entries), its pre-check on the root at %.2f (%llu candidate entries)\n",
                        jp->d, G, jp->c->device, jp->r0, jp->r1, jp->t_band, jp->t_scored, jp->t_sent, jp->t_csr, (unsigned long long)jp->c->h_counts[HC_TOTAL], jp->t_pre,
                        jp->pre_total);
        for (auto &jp : jobs) {
            if (jp->sent_state == -2) overflow = true;
            want_edges[jp->d] = std::max(want_edges[jp->d], jp->need_edges);
            want_inbox[jp->d] = std::max(want_inbox[jp->d], jp->need_inbox);
        }
        if (st == HMK_OK || st == HMK_ERR_REFERENCE_WOULD_CRASH) {   // (a crash-parity exit during phase 1 never reached before_full)
            const int err = first_error();
            if (err && !overflow) st = err;
        }
        if (st == ST_RETRY_OVERFLOW || (overflow && (st == HMK_OK || st == HMK_ERR_REFERENCE_WOULD_CRASH))) { st = ST_RETRY_OVERFLOW; continue; }
        break;
    }
    if (st == ST_RETRY_OVERFLOW) return fail(ctx, HMK_ERR_DEVICE, "internal edge buffer kept overflowing");
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_edges) == hipSuccess) ctx->phases.score_ms = ctx->phases.exchange_ms = ms;   // the root's shard, the exchange, the pieces
    (void)hipGetLastError();   // (a call that left early never recorded these events: "invalid resource handle" must not stay behind as the thread's last error)
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) stats->neighbors_ms = ctx->phases.score_ms;
    if (clink) clink->neighbors_ms = ctx->phases.score_ms;
    return st;
}

} }  // namespace hmk::impl
