// hmk_cluster.cpp -- cluster_on_device: edges -> CSR on the device, band hand-over, rows to the host merge on demand, pre-check and
// the device-side second loop (LimitedGreedySequenceClusterer.java:39-120 / ClinkageSequenceClusterer.java:43-124 on the graph).
#include "hmk_ctx.h"

namespace hmk { namespace impl {

// CSR of the rows [r0, r1) of the graph from `segs`, on c's device, enqueued on q.  The whole graph ([0, n)) in a single-device call; in a
// multi-device call the piece this device owns -- start[] / up[] are indexed by the row itself, rows outside the piece stay empty.
hipError_t piece_enqueue_csr(hmk_ctx *c, const EdgeSegs &segs, bool symmetric, bool packed, int base, uint32_t n, uint32_t r0, uint32_t r1,
                             bool deg_fused, bool deg_split, hipStream_t q) {
    const size_t esz = packed ? sizeof(NbrPacked) : sizeof(Nbr);
    uint64_t records = 1;   // one per edge: at most what the segments hold
    for (uint32_t k = 0; k < segs.n; k++) records += segs.s[k].cap;
    hipError_t e = ensure_buf(c, SB_DEG, (size_t)n * 8);
    if (e == hipSuccess) e = ensure_buf(c, SB_CURSOR, (size_t)n * 8);
    if (e == hipSuccess) e = ensure_buf(c, SB_START, ((size_t)n + 1) * 8);
    if (e == hipSuccess) e = ensure_buf(c, SB_SCAN, scan_scratch_bytes(n));
    if (e == hipSuccess) e = ensure_buf(c, SB_RANGE, 64);
    if (e == hipSuccess) e = ensure_buf(c, SB_ADJ, (symmetric ? 2 : 1) * records * esz);
    if (e != hipSuccess) return e;
    uint32_t *d_deg = buf<uint32_t>(c, SB_DEG), *d_cursor = buf<uint32_t>(c, SB_CURSOR);
    uint64_t *d_start = buf<uint64_t>(c, SB_START);
    const bool whole = r0 == 0 && r1 >= n;
    if ((e = hipMemsetAsync(d_cursor, 0, (size_t)n * 8, q)) != hipSuccess) return e;
    if (deg_fused) {
        e = launch_csr_scan_only(d_deg, deg_split ? d_deg + n : nullptr, d_start, n, buf<uint64_t>(c, SB_SCAN), buf<int>(c, SB_RANGE), q);
    } else {
        if ((e = hipMemsetAsync(d_deg, 0, (size_t)n * 4, q)) != hipSuccess) return e;
        e = launch_csr_degree_scan(segs, n, whole ? n : r1, symmetric, d_deg, d_start, buf<uint64_t>(c, SB_SCAN), buf<int>(c, SB_RANGE), q, whole ? 0 : r0, n);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_counts + HC_RANGE, buf<int>(c, SB_RANGE), 3 * sizeof(int), hipMemcpyDeviceToHost, q);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_counts + HC_TOTAL, d_start + n, 8, hipMemcpyDeviceToHost, q);
    if (e != hipSuccess) return e;
    if (csr_by_bucket(symmetric, packed)) {   // lower sections dealt by bucket
        e = ensure_buf(c, SB_PART, records * 8);
        if (e == hipSuccess) e = ensure_buf(c, SB_PARTSCR, csr_partition_scratch_bytes());
        if (e == hipSuccess)
            e = launch_csr_scatter_partitioned(segs, d_start, d_cursor, buf<void>(c, SB_ADJ), base, n, buf<uint64_t>(c, SB_PART), buf<void>(c, SB_PARTSCR),
                                               deg_fused && deg_split ? d_deg + n : nullptr, c->sw.csr_bucket_shift, q, whole ? 0 : r0, whole ? n : r1);
    } else {
        e = launch_csr_scatter(segs, symmetric, d_start, d_cursor, buf<void>(c, SB_ADJ), packed, base, whole ? n : r1, q, 0, whole ? 0 : r0);
    }
    if (e == hipSuccess) e = hipEventRecord(c->ev_csr, q);
    return e;
}

// Phase 1's result -- cluster_of, the clusters' member counts, the leftover list -- to c's device, on its copy stream (while the pass and
// the CSR build are still running on the clustering stream: three copies, the bitmap kernel and their launch latencies leave the call's
// critical path); records c->ev_bandcsr (the band hand-over's event: that hand-over is long over, and in stream order before this).
hipError_t piece_precheck_upload(hmk_ctx *c, const PreIn &in) {
    hipError_t r = ensure_buf(c, SB_COF, (size_t)in.n * 4);
    if (r == hipSuccess) r = ensure_buf(c, SB_BITMAP, ((size_t)in.n + 31) / 32 * 4);
    if (r == hipSuccess) r = ensure_buf(c, SB_USIZE, std::max<size_t>(in.ncl, 1) * 4);
    if (r == hipSuccess) r = ensure_buf(c, SB_LEFT, std::max<size_t>(in.nl, 1) * 4);
    if (r != hipSuccess) return r;
    hipStream_t C = c->copy_stream;
    r = hipMemcpyAsync(buf<int32_t>(c, SB_COF), in.h_block, in.b_cof, hipMemcpyHostToDevice, C);
    if (r == hipSuccess) r = launch_cluster_bitmap(buf<int32_t>(c, SB_COF), in.n, buf<uint32_t>(c, SB_BITMAP), C);
    if (r == hipSuccess && in.b_us) r = hipMemcpyAsync(buf<int32_t>(c, SB_USIZE), in.h_block + in.b_cof, in.b_us, hipMemcpyHostToDevice, C);
    if (r == hipSuccess && in.b_left) r = hipMemcpyAsync(buf<uint32_t>(c, SB_LEFT), in.h_block + in.b_cof + in.b_us, in.b_left, hipMemcpyHostToDevice, C);
    if (r == hipSuccess) r = hipEventRecord(c->ev_bandcsr, C);
    return r;
}

// The single-pass pre-check of the leftovers whose rows c's device holds ([r0, r1)), entries into the regions [region_base, + region_count)
// of the candidate buffer.  One pass: every wave takes its block of entries from the counter of its workgroup's region; rows with few
// neighbours inside clusters go through small tables first (in.two_stage).
int piece_precheck(hmk_ctx *c, const PreIn &in, uint32_t r0, uint32_t r1, uint32_t region_base, uint32_t region_count, hipStream_t q,
                   bool upload, unsigned long long *total) {
    *total = 0;
    if (upload && piece_precheck_upload(c, in) != hipSuccess) return -1;
    const uint32_t nl = in.nl;
    hipError_t r = ensure_buf(c, SB_CNT, std::max<size_t>(nl, 1) * 4);
    if (r == hipSuccess) r = ensure_buf(c, SB_CSTART, ((size_t)nl + 1) * 4);
    if (r == hipSuccess) r = ensure_buf(c, SB_OVER, 64);
    if (r == hipSuccess) r = ensure_buf(c, SB_CAND, (size_t)in.region_cap * HMK_PRE_REGIONS * sizeof(GreedyCand));
    if (r == hipSuccess) r = ensure_buf(c, SB_PRECNT, HMK_PRE_REGIONS * sizeof(unsigned long long));
    if (r == hipSuccess) r = ensure_pinned(&c->h_stage, &c->h_stage_cap, HMK_PRE_REGIONS * sizeof(unsigned long long) + 64, 0);
    if (r == hipSuccess && in.two_stage) r = ensure_buf(c, SB_RETRY, std::max<size_t>(nl, 1) * 4);
    if (r != hipSuccess) return -1;
    uint32_t *d_over = buf<uint32_t>(c, SB_OVER);                      // [0] table overflows, [1] rows of the second stage
    unsigned long long *d_regions = buf<unsigned long long>(c, SB_PRECNT);
    uint32_t *h_misc = (uint32_t *)(c->h_counts + HC_MISC);
    unsigned long long *h_regions = (unsigned long long *)c->h_stage;   // (the block's first HMK_PRE_REGIONS words; the root's PreIn block lies behind them)
    r = hipStreamWaitEvent(q, c->ev_bandcsr, 0);
    if (r == hipSuccess) r = hipMemsetAsync(d_over, 0, 16, q);
    if (r == hipSuccess) r = hipMemsetAsync(d_regions, 0, HMK_PRE_REGIONS * sizeof(unsigned long long), q);
    if (r == hipSuccess)
        r = launch_greedy_precheck(2, in.packed, buf<uint64_t>(c, SB_START), buf<void>(c, SB_ADJ), buf<int32_t>(c, SB_COF), buf<uint32_t>(c, SB_BITMAP),
                                   buf<int32_t>(c, SB_USIZE), buf<uint32_t>(c, SB_LEFT), nl, buf<uint32_t>(c, SB_CNT), buf<uint32_t>(c, SB_CSTART),
                                   buf<GreedyCand>(c, SB_CAND), d_over, d_regions, in.region_cap, in.two_stage ? buf<uint32_t>(c, SB_RETRY) : nullptr,
                                   d_over + 1, in.first_slots, q, r0, r1, region_base, region_count);
    if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[0], d_over, 4, hipMemcpyDeviceToHost, q);
    if (r == hipSuccess) r = hipMemcpyAsync(h_regions, d_regions, HMK_PRE_REGIONS * sizeof(unsigned long long), hipMemcpyDeviceToHost, q);
    if (r == hipSuccess) r = hipStreamSynchronize(q);
    if (r != hipSuccess || h_misc[0] != 0) return -1;   // a row overflowed its hash table: host pre-check
    bool fits = true;
    for (uint32_t g = region_base; g < region_base + region_count; g++) { *total += h_regions[g]; fits = fits && h_regions[g] <= in.region_cap; }
    return fits ? 0 : 1;
}

// Builds the CSR adjacency on the device, hands rows to the host merge on demand, runs the merge.
int cluster_on_device(hmk_ctx *ctx, const EdgeSource &src, int max_clusters, int32_t *cluster_id, int32_t *result_order,
                      int32_t *member_rank, hmk_greedy_stats *stats, std::chrono::steady_clock::time_point t0) {
    const uint32_t n = ctx->n;
    hipStream_t S = ctx->gstream, C = ctx->copy_stream;
    hmk_greedy_phases &ph = ctx->phases;
    auto ms_since = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    const Switches &sw = ctx->sw;
    const bool timing = sw.greedy_timing;
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[hmk greedy] %s at %.2f ms\n", what, ms_since(t0));
    };
    bool packed = src.packed;
    int base = src.base;
    size_t esz = packed ? sizeof(NbrPacked) : sizeof(Nbr);
    const bool symmetric = src.symmetric;
    // multi-device calls: the adjacency is built in pieces by the devices' workers (hmk_multi.cpp); piece_base[d] = the entries of
    // the pieces before d (where piece d's rows begin in the host's single adj[] when rows are fetched)
    const bool multi = src.pieces.size() > 1;
    std::vector<EdgeSource::Piece> pieces = src.pieces;
    if (pieces.empty()) pieces.push_back(EdgeSource::Piece{ctx, 0, n});
    std::vector<uint64_t> piece_base(pieces.size() + 1, 0);

    // ---- full CSR on the device, enqueued behind the scoring on S ---------------------------------------
    HIPCHK(ctx, ensure_buf(ctx, SB_DEG, (size_t)n * 4));
    HIPCHK(ctx, ensure_buf(ctx, SB_CURSOR, (size_t)n * 8));
    HIPCHK(ctx, ensure_buf(ctx, SB_START, ((size_t)n + 1) * 8));
    HIPCHK(ctx, ensure_buf(ctx, SB_SCAN, scan_scratch_bytes(n)));
    HIPCHK(ctx, ensure_buf(ctx, SB_RANGE, 64));
    const bool late_buffers = late_buffers_pending(ctx);   // hmk_reserve's thread is still getting SB_ADJ / SB_PART: the CSR is enqueued later
    if (src.format_known && !late_buffers) HIPCHK(ctx, ensure_buf(ctx, SB_ADJ, std::max<uint64_t>(src.adj_bound, 1) * esz));
    HIPCHK(ctx, ensure_pinned(&ctx->h_start, &ctx->h_start_cap, ((size_t)n + 1) * 8 + (size_t)n * 4 + 64, 0));
    uint64_t *h_start = (uint64_t *)ctx->h_start;
    uint32_t *h_up = (uint32_t *)((char *)ctx->h_start + ((size_t)n + 1) * 8);
    int *h_range = (int *)(ctx->h_counts + HC_RANGE);

    // (multi-device calls enqueue it later, from wait_full(): the peers' edges are not there yet)
    bool scatter_enqueued = false;
    auto enqueue_scatter = [&]() -> hipError_t {
        scatter_enqueued = true;
        if (late_buffers && src.format_known) {
            const hipError_t e = ensure_buf(ctx, SB_ADJ, std::max<uint64_t>(src.adj_bound, 1) * esz);   // (joins the thread)
            if (e != hipSuccess) return e;
        }
        if (csr_by_bucket(symmetric, packed)) {   // lower sections dealt by bucket
            uint64_t records = 1;   // one per edge: at most what the segments hold
            for (uint32_t q = 0; q < src.segs.n; q++) records += src.segs.s[q].cap;
            hipError_t e = ensure_buf(ctx, SB_PART, records * 8);   // in place already when hmk_greedy_cluster scored the edges itself
            if (e == hipSuccess) e = ensure_buf(ctx, SB_PARTSCR, csr_partition_scratch_bytes());
            if (e == hipSuccess)
                e = launch_csr_scatter_partitioned(src.segs, buf<uint64_t>(ctx, SB_START), buf<uint32_t>(ctx, SB_CURSOR), buf<void>(ctx, SB_ADJ),
                                                   base, n, buf<uint64_t>(ctx, SB_PART), buf<void>(ctx, SB_PARTSCR),
                                                   src.deg_fused && src.deg_split ? buf<uint32_t>(ctx, SB_DEG) + n : nullptr, sw.csr_bucket_shift, S);
            if (e == hipSuccess) e = hipEventRecord(ctx->ev_csr, S);
            return e;
        }
        hipError_t e = launch_csr_scatter(src.segs, symmetric, buf<uint64_t>(ctx, SB_START), buf<uint32_t>(ctx, SB_CURSOR),
                                          buf<void>(ctx, SB_ADJ), packed, base, n, S);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_csr, S);
        return e;
    };
    auto enqueue_counts = [&]() -> hipError_t {
        hipError_t e_;
        if (src.deg_fused) {
            if ((e_ = (hipMemsetAsync(buf<void>(ctx, SB_CURSOR), 0, (size_t)n * 8, S))) != hipSuccess) return e_;
            if ((e_ = (launch_csr_scan_only(buf<uint32_t>(ctx, SB_DEG), src.deg_split ? buf<uint32_t>(ctx, SB_DEG) + n : nullptr,
                                             buf<uint64_t>(ctx, SB_START), n, buf<uint64_t>(ctx, SB_SCAN), buf<int>(ctx, SB_RANGE), S))) != hipSuccess) return e_;
        } else {
            if ((e_ = (hipMemsetAsync(buf<void>(ctx, SB_CURSOR), 0, (size_t)n * 8, S))) != hipSuccess) return e_;
            if ((e_ = (hipMemsetAsync(buf<void>(ctx, SB_DEG), 0, (size_t)n * 4, S))) != hipSuccess) return e_;
            if ((e_ = (launch_csr_degree_scan(src.segs, n, n, symmetric, buf<uint32_t>(ctx, SB_DEG), buf<uint64_t>(ctx, SB_START),
                                               buf<uint64_t>(ctx, SB_SCAN), buf<int>(ctx, SB_RANGE), S))) != hipSuccess) return e_;
        }
        if ((e_ = (hipMemcpyAsync(h_range, buf<int>(ctx, SB_RANGE), 3 * sizeof(int), hipMemcpyDeviceToHost, S))) != hipSuccess) return e_;
        if ((e_ = (hipMemcpyAsync(&h_start[n], buf<uint64_t>(ctx, SB_START) + n, 8, hipMemcpyDeviceToHost, S))) != hipSuccess) return e_;
    return hipSuccess;
    };
    bool full_enqueued = false;
    auto enqueue_full = [&]() -> hipError_t {
        full_enqueued = true;
        hipError_t e = enqueue_counts();
        if (e == hipSuccess && src.format_known) e = enqueue_scatter();
        return e;
    };
    if (!multi && !src.before_full && !late_buffers) HIPCHK(ctx, enqueue_full());

    // ---- band: the first rows' adjacency from the edges of the band launch, on the copy stream --------------
    uint32_t rows_lo = 0, rows_here = 0;   // rows [rows_lo, rows_here) are valid in h_start / the host adjacency block
    uint64_t adj_off = 0;                  // ... which starts at entry adj_off of the CSR's adj[] (a window that begins past row 0)
    bool band_pending = false, band_used = false;
    uint32_t R1 = src.band_rows;
    if (R1 > 0 && src.before_band && src.before_band() != HMK_OK) R1 = 0;   // (the peers' band blocks did not make it: no band)
    // symmetric scores in 4-byte entries: the band reaches the host PREPARED for phase 1 (BandPack: near rows, best far
    // candidates, transposed lists -- k_band_*); else as whole rows
    const bool prepared = R1 > 0 && src.format_known && symmetric && packed;
    constexpr uint32_t FAR_T = 8;
    if (R1 > 0 && src.format_known) {
        HIPCHK(ctx, ensure_buf(ctx, SB_BDEG, (size_t)R1 * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_BCURSOR, (size_t)R1 * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_BSTART, ((size_t)R1 + 1) * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_BSCAN, scan_scratch_bytes(prepared ? n : R1)));
        HIPCHK(ctx, ensure_buf(ctx, SB_BRANGE, 64));
        if (prepared) {
            HIPCHK(ctx, ensure_buf(ctx, SB_BNCNT, (size_t)R1 * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_BNUP, (size_t)R1 * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_BNSTART, ((size_t)R1 + 1) * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_BFTOP, (size_t)R1 * (FAR_T + BandPack::NEAR_T) * 4));   // far_top, then near_top
            HIPCHK(ctx, ensure_buf(ctx, SB_BFMORE, (size_t)R1 + 64));
            HIPCHK(ctx, ensure_buf(ctx, SB_FDEG, (size_t)n * 12 + 64));   // fdeg | fcur | owner_of (one fill), then the two totals
            HIPCHK(ctx, ensure_buf(ctx, SB_FSTART, (size_t)n * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_TRCNT, (size_t)R1 * BandPack::TR_PER_ROW * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_TRSTART, ((size_t)R1 * BandPack::TR_PER_ROW + 1) * 4));
            if (ctx->has_sizes) HIPCHK(ctx, ensure_buf(ctx, SB_SEQSZ, (size_t)n * 4));
        }
        // (the fills first: they run beside the band launch, whose end the rest waits for)
        HIPCHK(ctx, hipMemsetAsync(buf<void>(ctx, SB_BDEG), 0, (size_t)R1 * 4, C));
        HIPCHK(ctx, hipMemsetAsync(buf<void>(ctx, SB_BCURSOR), 0, (size_t)R1 * 8, C));
        if (prepared) {
            HIPCHK(ctx, hipMemsetAsync(buf<void>(ctx, SB_FDEG), 0, (size_t)n * 12 + 16, C));   // (+ the totals and the list allocator's counter)
            if (ctx->has_sizes) HIPCHK(ctx, hipMemcpyAsync(buf<void>(ctx, SB_SEQSZ), ctx->sizes.data(), (size_t)n * 4, hipMemcpyHostToDevice, C));
        }
        HIPCHK(ctx, hipStreamWaitEvent(C, ctx->ev_band, 0));
        HIPCHK(ctx, launch_csr_degree_scan(src.band_segs, n, R1, symmetric, buf<uint32_t>(ctx, SB_BDEG), buf<uint64_t>(ctx, SB_BSTART),
                                           buf<uint64_t>(ctx, SB_BSCAN), buf<int>(ctx, SB_BRANGE), C));
        HIPCHK(ctx, hipMemcpyAsync(h_start, buf<uint64_t>(ctx, SB_BSTART), ((size_t)R1 + 1) * 8, hipMemcpyDeviceToHost, C));

        // (the band's own segments only: beside a pass that runs at the same time the other cursors are in motion)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts + HC_BAND, src.band_segs.s[0].count,
                                   std::min<uint32_t>(src.band_segs.n, HMK_EDGE_SHARDS) * sizeof(unsigned long long), hipMemcpyDeviceToHost, C));
        HIPCHK(ctx, hipEventRecord(ctx->ev_bandcsr, C));
        band_pending = true;
    }
    lap("scoring, CSR and band hand-over enqueued");

    int status_inside = HMK_OK;   // failure inside a hook (the merge then stops with its own error)
    std::string hook_err;
    auto hook_fail = [&](int code, const std::string &msg) { status_inside = code; hook_err = msg; };

    // the full CSR is complete (and trustworthy: no segment overflowed)
    bool full_ready = false;
    auto wait_full = [&]() -> bool {
        if (full_ready) return true;
        if (multi) {
            const int r = src.before_full ? src.before_full() : HMK_ERR_DEVICE;
            if (r != HMK_OK) { hook_fail(r, ctx->err.empty() ? "building the adjacency pieces failed" : ctx->err); return false; }
            uint64_t sum = 0;
            for (size_t d = 0; d < pieces.size(); d++) {
                const int *rg = (const int *)(pieces[d].c->h_counts + HC_RANGE);
                if (rg[2] != 0) { hook_fail(HMK_ERR_BAD_ARG, "edge list references a sequence outside [0, n) or a self pair"); return false; }
                piece_base[d] = sum;
                sum += pieces[d].c->h_counts[HC_TOTAL];
            }
            piece_base[pieces.size()] = sum;
            h_start[n] = sum;
            const uint64_t want = symmetric ? 2 * src.total_edges : src.total_edges;
            if (sum != want) { hook_fail(HMK_ERR_DEVICE, "CSR build: the pieces' entries do not add up to the edges scored"); return false; }
            full_enqueued = scatter_enqueued = full_ready = true;
            lap("every piece of the CSR on its device");
            return true;
        }
        if (!full_enqueued) {   // (the late buffers are ready only now)
            const int r = src.before_full ? src.before_full() : HMK_OK;
            if (r != HMK_OK) { hook_fail(r, ctx->err.empty() ? "gathering the peers' edges failed" : ctx->err); return false; }
            const hipError_t e0 = enqueue_full();
            if (e0 != hipSuccess) { hook_fail(e0 == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string("CSR build: ") + hipGetErrorString(e0)); return false; }
        }
        hipError_t e = hipEventSynchronize(ctx->ev_edges);
        if (e == hipSuccess && src.check_overflow) {
            for (int q = 0; q < HMK_EDGE_SHARDS; q++)
                if (ctx->h_counts[q] > src.seg_cap) { hook_fail(ST_RETRY_OVERFLOW, "edge segment overflow"); return false; }
        }
        if (e == hipSuccess && !scatter_enqueued) {
            // the adjacency format depends on the scores found: 4-byte entries when they span at most 255
            e = hipStreamSynchronize(S);
            if (e == hipSuccess && h_range[2] != 0) {
                hook_fail(HMK_ERR_BAD_ARG, "edge list references a sequence outside [0, n) or a self pair");
                return false;
            }
            if (e == hipSuccess) {
                packed = h_start[n] == 0 || ((long long)h_range[1] - h_range[0] <= 255 && !sw.adj_8byte);
                base = h_range[0];
                esz = packed ? sizeof(NbrPacked) : sizeof(Nbr);
                e = ensure_buf(ctx, SB_ADJ, std::max<uint64_t>(h_start[n], 1) * esz);
            }
            if (e == hipSuccess) e = enqueue_scatter();
        }
        if (e == hipSuccess) e = hipEventSynchronize(ctx->ev_csr);
        if (e != hipSuccess) { hook_fail(e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string("CSR build: ") + hipGetErrorString(e)); return false; }
        if (h_range[2] != 0) { hook_fail(HMK_ERR_BAD_ARG, "edge list references a sequence outside [0, n) or a self pair"); return false; }
        const uint64_t want = src.total_known ? (symmetric ? 2 * src.total_known : src.total_known) : h_start[n];
        if (h_start[n] != want) { hook_fail(HMK_ERR_DEVICE, "CSR build: adjacency size mismatch"); return false; }
        full_ready = true;
        lap("full CSR on the device");
        return true;
    };

    GreedyHooks hooks;
    double t_rows = 0;   // host time spent waiting for rows
    // The band's row starts have landed and no band segment overflowed: its CSR is built on the device (copy stream).
    // -> the number of entries, or -1: no band in this call (phase 1 then waits for the full graph)
    auto band_csr = [&]() -> int64_t {
        hipError_t e = hipEventSynchronize(ctx->ev_bandcsr);
        bool ok = e == hipSuccess;
        for (uint32_t q = 0; q < std::min<uint32_t>(src.band_segs.n, HMK_EDGE_SHARDS) && ok; q++) ok = ctx->h_counts[HC_BAND + q] <= src.seg_cap;
        if (!ok) {
            if (e != hipSuccess) hook_fail(HMK_ERR_DEVICE, std::string("band hand-over: ") + hipGetErrorString(e));
            return -1;
        }
        const uint64_t entries = h_start[R1];
        e = ensure_buf(ctx, SB_BADJ, std::max<uint64_t>(entries, 1) * esz);
        if (e == hipSuccess)
            e = launch_csr_scatter(src.band_segs, symmetric, buf<uint64_t>(ctx, SB_BSTART), buf<uint32_t>(ctx, SB_BCURSOR),
                                   buf<void>(ctx, SB_BADJ), packed, base, R1, C, n);
        if (e != hipSuccess) { hook_fail(e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string("band hand-over: ") + hipGetErrorString(e)); return -1; }
        return (int64_t)entries;
    };
    // ---- the band, prepared for phase 1 (BandPack, hmk_internal.h) ---------------------------------------------------------
    BandPack pack;
    if (prepared) {
        hooks.band_pack = [&]() -> const BandPack * {
            if (!band_pending || ctx->wedged) return nullptr;
            band_pending = false;
            const auto tw = std::chrono::steady_clock::now();
            const int64_t entries = band_csr();
            if (entries < 0) return nullptr;
            const uint32_t TRN = R1 * BandPack::TR_PER_ROW;
            const uint64_t cap = std::max<uint64_t>((uint64_t)entries, 1);   // no list is longer than the band has entries
            hipError_t e = ensure_buf(ctx, SB_FADJ, cap * 4);
            // the host's block (pinned; the kernels store into it themselves: no copy launches, no second round trip for sizes):
            //   near_start [R1 + 1] | near_up [R1] | far_top [R1 x FAR_T] | tr_cnt [TRN] | tr_start [TRN + 1] | near_top [R1 x NEAR_T] | far_more [R1, padded] | near [cap] | tr [cap]
            const size_t o_nstart = 0, o_nup = o_nstart + ((size_t)R1 + 1) * 4, o_ftop = o_nup + (size_t)R1 * 4, o_trowner = o_ftop + (size_t)R1 * FAR_T * 4,
                         o_trstart = o_trowner + (size_t)TRN * 4, o_ntop = o_trstart + ((size_t)TRN + 1) * 4, o_fmore = o_ntop + (size_t)R1 * BandPack::NEAR_T * 4,
                         o_near = (o_fmore + R1 + 63) / 64 * 64,
                         o_tr = o_near + (cap * 4 + 63) / 64 * 64;
            if (e == hipSuccess) e = ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, o_tr + cap * 4 + 64, 0);
            char *hb = (char *)ctx->h_adj, *db = nullptr;
            if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&db, hb, 0);
            uint32_t *d_fdeg = buf<uint32_t>(ctx, SB_FDEG), *d_fcur = d_fdeg + n, *d_totals = d_fdeg + 3 * (size_t)n;
            if (e == hipSuccess)
                e = launch_band_prepare(buf<uint64_t>(ctx, SB_BSTART), buf<uint32_t>(ctx, SB_BCURSOR), buf<void>(ctx, SB_BADJ), R1, (uint64_t)entries, n, FAR_T,
                                        BandPack::TR_PER_ROW, ctx->has_sizes ? buf<int32_t>(ctx, SB_SEQSZ) : nullptr, buf<uint32_t>(ctx, SB_BNCNT),
                                        buf<uint32_t>(ctx, SB_BNUP), buf<uint32_t>(ctx, SB_BNSTART), buf<uint32_t>(ctx, SB_BFTOP), buf<uint8_t>(ctx, SB_BFMORE),
                                        d_fdeg, d_fcur, d_totals, buf<uint32_t>(ctx, SB_FSTART), buf<uint32_t>(ctx, SB_FADJ),
                                        buf<uint32_t>(ctx, SB_TRCNT), buf<uint32_t>(ctx, SB_TRSTART), cap, (uint32_t *)(db + o_nstart), (uint32_t *)(db + o_nup),
                                        (uint32_t *)(db + o_ftop), (uint8_t *)(db + o_fmore), (uint32_t *)(db + o_near), (uint32_t *)(db + o_trowner),
                                        (uint32_t *)(db + o_trstart), (uint32_t *)(db + o_tr), BandPack::NEAR_T, buf<uint32_t>(ctx, SB_BFTOP) + (size_t)R1 * FAR_T,
                                        (uint32_t *)(db + o_ntop), C);
            if (e == hipSuccess) e = hipStreamSynchronize(C);
            if (e != hipSuccess) { hook_fail(e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string("band hand-over: ") + hipGetErrorString(e)); return nullptr; }
            const uint64_t n_near = ((const uint32_t *)(hb + o_nstart))[R1];
            uint64_t n_tr = 0;   // what was written of the intersections (their room, tr_start, is an upper bound)
            for (uint32_t u = 0; u < TRN; u++) { const uint32_t c = ((const uint32_t *)(hb + o_trowner))[u]; if (c != ~0u) n_tr += c; }
            if (n_near > cap || n_tr > cap) { hook_fail(HMK_ERR_DEVICE, "band hand-over: list sizes beyond the band's entries"); return nullptr; }
            pack.rows = R1;
            pack.far_t = FAR_T;
            pack.near_start = (const uint32_t *)(hb + o_nstart);
            pack.near_up = (const uint32_t *)(hb + o_nup);
            pack.near = (const uint32_t *)(hb + o_near);
            pack.far_top = (const uint32_t *)(hb + o_ftop);
            pack.near_top = (const uint32_t *)(hb + o_ntop);
            pack.far_more = (const uint8_t *)(hb + o_fmore);
            pack.tr_cnt = (const uint32_t *)(hb + o_trowner);
            pack.tr_start = (const uint32_t *)(hb + o_trstart);
            pack.tr = (const uint32_t *)(hb + o_tr);
            ph.band_bytes = (uint64_t)(o_near + n_near * 4 + n_tr * 4);
            t_rows += ms_since(tw);
            lap("band prepared and on the host");
            return &pack;
        };
        // a far sequence's band neighbours / a band row's far part, straight from the device (rare: a candidate beyond the lists sent)
        // (into PINNED memory: a copy into pageable memory issued while the pass runs took 8 ms -- 12 such fetches made phase 1 of
        // the 10^6 call 200 ms longer)
        hooks.far_row = [&](uint32_t id, std::vector<uint32_t> &out) -> bool {
            out.clear();
            uint32_t *se = (uint32_t *)(ctx->h_counts + HC_MISC) + 10;   // pinned: the list's start and length
            hipError_t e = hipMemcpyAsync(se, buf<uint32_t>(ctx, SB_FSTART) + id, 4, hipMemcpyDeviceToHost, C);
            if (e == hipSuccess) e = hipMemcpyAsync(se + 1, buf<uint32_t>(ctx, SB_FDEG) + id, 4, hipMemcpyDeviceToHost, C);
            if (e == hipSuccess) e = hipStreamSynchronize(C);
            if (e == hipSuccess && se[1] > 0) {
                const size_t len = se[1], from = se[0];
                e = ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, len * 4 + 64, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_stage, buf<uint32_t>(ctx, SB_FADJ) + from, len * 4, hipMemcpyDeviceToHost, C);
                if (e == hipSuccess) e = hipStreamSynchronize(C);
                if (e == hipSuccess) out.assign((const uint32_t *)ctx->h_stage, (const uint32_t *)ctx->h_stage + len);
            }
            if (e != hipSuccess) { hook_fail(HMK_ERR_DEVICE, std::string("band hand-over (a far sequence's list): ") + hipGetErrorString(e)); return false; }
            return true;
        };
        hooks.band_far = [&](uint32_t x, std::vector<uint32_t> &out) -> bool {
            out.clear();
            uint32_t *h_one = (uint32_t *)(ctx->h_counts + HC_MISC) + 12;   // pinned: the row's upper-section size
            hipError_t e = hipMemcpyAsync(h_one, buf<uint32_t>(ctx, SB_BCURSOR) + x, 4, hipMemcpyDeviceToHost, C);
            if (e == hipSuccess) e = hipStreamSynchronize(C);
            const uint32_t up = e == hipSuccess ? *h_one : 0u;
            if (up) {
                e = ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, (size_t)up * 4 + 64, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_stage, buf<uint32_t>(ctx, SB_BADJ) + h_start[x], (size_t)up * 4, hipMemcpyDeviceToHost, C);
                if (e == hipSuccess) e = hipStreamSynchronize(C);
            }
            if (e != hipSuccess) { hook_fail(HMK_ERR_DEVICE, std::string("band hand-over (a row's far part): ") + hipGetErrorString(e)); return false; }
            for (uint32_t q = 0; q < up; q++)
                if ((((const uint32_t *)ctx->h_stage)[q] >> 8) >= R1) out.push_back(((const uint32_t *)ctx->h_stage)[q]);
            return true;
        };
    }
    hooks.need_rows = [&](uint32_t k, uint32_t from) -> uint32_t {
        if (ctx->wedged) return 0;   // (the second loop gave the device up: the merge stops here instead of waiting for it again)
        if (from >= rows_lo && k < rows_here) return rows_here;
        const auto tw = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        if (band_pending) {   // (the band as whole rows: asymmetric scores or 8-byte entries)
            band_pending = false;
            const int64_t entries = band_csr();
            if (entries < 0 && status_inside != HMK_OK) return 0;
            if (entries >= 0) {
                e = ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, std::max<uint64_t>((uint64_t)entries, 1) * esz, 0);
                if (e == hipSuccess && entries)
                    e = hipMemcpyAsync(ctx->h_adj, buf<void>(ctx, SB_BADJ), (uint64_t)entries * esz, hipMemcpyDeviceToHost, C);
                // the band rows' upper-section sizes travel with them: upper[] must never hold a previous call's values for rows
                // the merge may read
                if (e == hipSuccess && symmetric)
                    e = hipMemcpyAsync(h_up, buf<uint32_t>(ctx, SB_BCURSOR), (size_t)R1 * 4, hipMemcpyDeviceToHost, C);
                if (e == hipSuccess) e = hipStreamSynchronize(C);
                if (e != hipSuccess) { hook_fail(HMK_ERR_DEVICE, std::string("band hand-over: ") + hipGetErrorString(e)); return 0; }
                rows_lo = 0;
                rows_here = R1;
                adj_off = 0;
                band_used = true;
                lap("band rows on the host");
                if (from == 0 && k < rows_here) { t_rows += ms_since(tw); return rows_here; }
            }
        }
        // rows from the full CSR (which must be complete by now): the window [rows_lo, rows_here) grows at its end, or begins anew
        if (!wait_full()) return 0;
        if (band_used || from < rows_lo || rows_here == rows_lo) { rows_lo = rows_here = from; band_used = false; }   // (band rows come again, in the full CSR's layout)
        uint32_t r_end = n;
        if (k + 1 < n) r_end = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>({(uint64_t)k + 1, (uint64_t)rows_here + (rows_here - rows_lo), (uint64_t)rows_here + 8192}));
        // rows [rows_here, r_end) from the piece(s) that hold them: row starts and upper sizes first (a piece's starts count from its own
        // first entry: + piece_base), then the entries, into the host's one adj[]
        for (size_t d = 0; d < pieces.size() && e == hipSuccess; d++) {
            const uint32_t a = std::max(rows_here, pieces[d].r0), b = std::min(r_end, pieces[d].r1);
            if (a >= b) continue;
            hmk_ctx *pc = pieces[d].c;
            if (multi) (void)hipSetDevice(pc->device);
            e = hipMemcpyAsync(h_start + a, buf<uint64_t>(pc, SB_START) + a, ((size_t)(b - a) + 1) * 8, hipMemcpyDeviceToHost, pc->copy_stream);
            if (e == hipSuccess && symmetric)
                e = hipMemcpyAsync(h_up + a, buf<uint32_t>(pc, SB_CURSOR) + a, (size_t)(b - a) * 4, hipMemcpyDeviceToHost, pc->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(pc->copy_stream);
            if (e != hipSuccess) break;
            const uint64_t l0 = h_start[a], l1 = h_start[b];                       // (the piece's own offsets)
            const uint64_t a0 = piece_base[d] + l0, a1 = piece_base[d] + l1;
            if (a == rows_lo && rows_here == rows_lo) adj_off = a0;
            e = ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, std::max<uint64_t>(a1 - adj_off, 1) * esz, (a0 - adj_off) * esz);
            if (e == hipSuccess && a1 > a0)
                e = hipMemcpyAsync((char *)ctx->h_adj + (a0 - adj_off) * esz, (const char *)buf<void>(pc, SB_ADJ) + l0 * esz, (a1 - a0) * esz,
                                   hipMemcpyDeviceToHost, pc->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(pc->copy_stream);
            if (piece_base[d]) for (uint32_t x = a; x <= b; x++) h_start[x] += piece_base[d];
        }
        if (multi) (void)hipSetDevice(ctx->device);
        if (e != hipSuccess) { hook_fail(e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string("adjacency copy: ") + hipGetErrorString(e)); return 0; }
        rows_here = r_end;
        t_rows += ms_since(tw);
        return rows_here;
    };

    // ---- second loop on the device-resident CSR ---------------------------------------------------------------
    // (1) pre-check (k_greedy_precheck): per leftover the clusters that are feasible after phase 1 -> cand CSR on the device.
    // (2) the loop itself on the device in optimistic rounds (k_loop_*); where that does not apply (asymmetric scores, a
    // table overflow, HMK_SECOND_LOOP=host) the host's sequential loop takes the candidate lists.
    // pre_mode: 0 nothing yet, 1 = two passes done (cand_start[] are prefix sums: what the host-side consumers read),
    // 2 = one pass done (every leftover's block lies where the global counter put it: the device loop takes either)
    int pre_mode = 0;
    uint32_t pre_total_c = 0;
    auto device_precheck = [&](const int32_t *cluster_of, const std::vector<int32_t> &usize, const std::vector<uint32_t> &leftover,
                               bool single_pass) -> bool {
        if (pre_mode == 1 || (pre_mode == 2 && single_pass)) return true;
        if (sw.precheck == 1) single_pass = false;   // (HMK_PRECHECK=two_passes: the count + fill form a region overrun falls back to)
        if (multi && !single_pass) return false;     // (the pieces run the single pass only: the host's pre-check otherwise)
        const uint32_t nl = (uint32_t)leftover.size();
        PreIn in;
        in.n = n; in.nl = nl; in.ncl = (uint32_t)usize.size(); in.packed = packed;
        in.b_cof = (size_t)n * 4; in.b_us = usize.size() * 4; in.b_left = (size_t)nl * 4;
        bool uploaded_early = false;
        hipError_t r = hipSuccess;
        if (pre_mode == 0) {
            // Phase 1's result into a pinned block (an "async" upload from pageable memory is staged by the runtime chunk by chunk and
            // the stream waits for it: 0.3 ms for these 0.8 MB at 10^5); it goes up NOW, on the copy stream, while the pass and the CSR
            // build are still running (phase 1 ends before the scoring does at every size)
            r = ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, HMK_PRE_REGIONS * sizeof(unsigned long long) + in.b_cof + in.b_us + in.b_left + 64, 0);
            if (r != hipSuccess) return false;
            char *hs = (char *)ctx->h_stage + HMK_PRE_REGIONS * sizeof(unsigned long long);   // (the block starts with the single pass's region counters)
            std::memcpy(hs, cluster_of, in.b_cof);
            std::memcpy(hs + in.b_cof, usize.data(), in.b_us);
            std::memcpy(hs + in.b_cof + in.b_us, leftover.data(), in.b_left);
            in.h_block = hs;
            if (!multi) {
                if (piece_precheck_upload(ctx, in) != hipSuccess) return false;
                uploaded_early = true;
            }
        }
        in.h_block = (char *)ctx->h_stage + HMK_PRE_REGIONS * sizeof(unsigned long long);
        if (!wait_full()) return false;
        const auto tp = std::chrono::steady_clock::now();
        // The candidate buffer is sized from what previous calls needed (or 24 entries per leftover), HMK_PRE_REGIONS regions; a call that
        // overruns a region falls back to the two passes below.
        const size_t want = std::max<size_t>({ctx->sb[SB_CAND].cap / sizeof(GreedyCand), (size_t)nl * 24, (size_t)HMK_PRE_REGIONS * 64});
        in.region_cap = std::min<unsigned long long>(want, 0xFFFFFFFFull) / HMK_PRE_REGIONS;
        // rows with few neighbours inside clusters (the estimate: average degree x the clustered share of the sequences) go through small
        // tables first.  (10^6 default-threshold 12-mers give an estimate of 130; small tables first for them too -- 512 slots, five
        // workgroups per CU instead of two -- was measured and loses: 22.0 against 18.7 ms, 38.9 against 25.8 ms in the reference's
        // default order, where many rows see far more clusters than the average and are scanned twice)
        size_t in_clusters = 0;
        for (int32_t u : usize) in_clusters += (size_t)u;
        const double est = (double)h_start[n] / std::max<uint32_t>(n, 1) * (double)in_clusters / std::max<uint32_t>(n, 1);
        in.first_slots = est <= 24.0 ? 128 : 512;   // ~5 x the expected number of distinct clusters in a row
        in.two_stage = est <= 100.0 && sw.precheck != 2;   // (HMK_PRECHECK=one_stage: what dense rows run)
        if (multi) {
            unsigned long long total = 0;
            if (!src.precheck_pieces || !src.precheck_pieces(in, &total) || total > 0x7FFFFFFFull) return false;
            pre_total_c = (uint32_t)total;
            pre_mode = 2;
            ph.cand_entries = pre_total_c;
            ph.precheck_ms = ms_since(tp);
            return true;
        }
        if (single_pass) {
            unsigned long long total = 0;
            const int fit = piece_precheck(ctx, in, 0, n, 0, HMK_PRE_REGIONS, S, !uploaded_early && pre_mode == 0, &total);
            if (fit < 0) return false;
            if (fit == 0 && total > 0x7FFFFFFFull) return false;
            if (fit == 0) {
                pre_total_c = (uint32_t)total;
                pre_mode = 2;
                ph.cand_entries = pre_total_c;
                ph.precheck_ms = ms_since(tp);
                return true;
            }
            if (total > 0x7FFFFFFFull) return false;   // (entry indices: 31 bits, k_loop_subscribers' record keeps a flag in the 32nd)
            uploaded_early = true;                      // (more entries than a region holds: count, size, fill -- the inputs are up)
        }
        if (!uploaded_early && pre_mode == 0 && piece_precheck_upload(ctx, in) != hipSuccess) return false;
        r = ensure_buf(ctx, SB_CNT, std::max<size_t>(nl, 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_CSTART, ((size_t)nl + 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_OVER, 64);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SCAN2, scan_scratch_bytes(std::max<uint32_t>(nl, n)));
        if (r != hipSuccess) return false;
        int32_t *d_cof = buf<int32_t>(ctx, SB_COF), *d_usize = buf<int32_t>(ctx, SB_USIZE);
        uint32_t *d_left = buf<uint32_t>(ctx, SB_LEFT), *d_cnt = buf<uint32_t>(ctx, SB_CNT), *d_cstart = buf<uint32_t>(ctx, SB_CSTART);
        uint32_t *d_over = buf<uint32_t>(ctx, SB_OVER);                      // [0] table overflows, [2..3] the count pass's entry counter
        unsigned long long *d_total = (unsigned long long *)(d_over + 2);
        uint64_t *d_scan = buf<uint64_t>(ctx, SB_SCAN2);
        const uint64_t *d_start = buf<uint64_t>(ctx, SB_START);
        const void *d_adj = buf<void>(ctx, SB_ADJ);
        uint32_t *h_misc = (uint32_t *)(ctx->h_counts + HC_MISC);
        r = hipStreamWaitEvent(S, ctx->ev_bandcsr, 0);
        if (r == hipSuccess) r = hipMemsetAsync(d_over, 0, 16, S);
        if (r == hipSuccess) r = launch_greedy_precheck(0, packed, d_start, d_adj, d_cof, buf<uint32_t>(ctx, SB_BITMAP), d_usize, d_left, nl,
                                                        d_cnt, nullptr, nullptr, d_over, d_total, 0, nullptr, nullptr, 0, S);
        if (r == hipSuccess) r = launch_scan_u32(d_cnt, d_cstart, nl, d_scan, S);
        if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[0], d_over, 4, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[1], d_cstart + nl, 4, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess) r = hipStreamSynchronize(S);
        if (r != hipSuccess || h_misc[0] != 0) return false;   // a row overflowed its hash table: host pre-check
        pre_total_c = h_misc[1];
        if (pre_total_c > 0x7FFFFFFFu) return false;
        if (pre_total_c) {
            r = ensure_buf(ctx, SB_CAND, (size_t)pre_total_c * sizeof(GreedyCand));
            if (r == hipSuccess) r = launch_greedy_precheck(1, packed, d_start, d_adj, d_cof, buf<uint32_t>(ctx, SB_BITMAP), d_usize, d_left, nl,
                                                            d_cnt, d_cstart, buf<GreedyCand>(ctx, SB_CAND), d_over, d_total, 0, nullptr, nullptr, 0, S);
            if (r != hipSuccess) return false;
        }
        pre_mode = 1;
        ph.cand_entries = pre_total_c;
        ph.precheck_ms = ms_since(tp);   // enqueue + count pass; the fill pass completes under the consumer's first wait
        return true;
    };
    auto fetch_cand = [&](uint32_t nl, std::vector<uint32_t> &cand_start, std::vector<GreedyCand> &cand) -> bool {
        cand_start.assign((size_t)nl + 1, 0);
        cand.resize(pre_total_c);
        hipError_t r = hipMemcpyAsync(cand_start.data(), buf<uint32_t>(ctx, SB_CSTART), ((size_t)nl + 1) * 4, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess && pre_total_c)
            r = hipMemcpyAsync(cand.data(), buf<GreedyCand>(ctx, SB_CAND), (size_t)pre_total_c * sizeof(GreedyCand), hipMemcpyDeviceToHost, S);
        if (r == hipSuccess) r = hipStreamSynchronize(S);
        return r == hipSuccess;
    };
    const bool forbid_device = sw.second_loop == 2;   // HMK_SECOND_LOOP=host (tests: the host's sequential loop over the candidate lists)

    hooks.device_loop = [&](const int32_t *cluster_of, const std::vector<int32_t> &usize, const std::vector<int64_t> &csize,
                            const std::vector<int32_t> &cids, const std::vector<uint32_t> &leftover,
                            std::vector<int32_t> &join_slot) -> bool {
        if (forbid_device || !symmetric || !ctx->h_loop) return false;   // (no coherent host block for the progress word: the host loop)
        if (!device_precheck(cluster_of, usize, leftover, true)) return false;
        // (measured: the device-side loop beats the host loop over device-built lists at every size -- 1e5 uniform 12-mers
        // 7.5 against 9.5 ms end to end, the antibodies example 14 against 18 ms; the lists stay as the second path)
        const auto tl = std::chrono::steady_clock::now();
        const uint32_t nl = (uint32_t)leftover.size();
        const uint32_t ncl = (uint32_t)usize.size();
        hipError_t r = ensure_buf(ctx, SB_JOINED, std::max<size_t>(ncl, 1) * 16);   // {joined, id, size} per cluster
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SUBSTART, ((size_t)ncl + 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SUBS, std::max<size_t>(pre_total_c, 1) * 8);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SUBS2, std::max<size_t>(pre_total_c, 1) * 8);   // merge scratch of the subscriber sort
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SCAN2, scan_scratch_bytes(std::max<uint32_t>({nl, n, ncl})));
        if (r == hipSuccess) r = ensure_buf(ctx, SB_CSIZE, std::max<size_t>(ncl, 1) * 8);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_CID, std::max<size_t>(ncl, 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_FIRST, std::max<size_t>(ncl, 1) * 12);   // first[], taken[], list cursor[] per cluster
        if (r == hipSuccess) r = ensure_buf(ctx, SB_STATUS, std::max<size_t>(nl, 1));
        if (r == hipSuccess) r = ensure_buf(ctx, SB_ACTIVE, std::max<size_t>(nl, 1) * 8);   // two eval lists
        if (r == hipSuccess) r = ensure_buf(ctx, SB_DIRTY, std::max<size_t>(nl, 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_CHOICE, std::max<size_t>(nl, 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_ACCEPTED, std::max<size_t>(nl, 1) * 4);
        // which cluster a leftover joins: the kernels STORE it into the host's pinned block (a handful of writes per round; nothing on the
        // device reads it) -- no copy and no second synchronise when the loop is over
        int32_t *d_jslot = nullptr;
        if (r == hipSuccess) r = ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, (size_t)std::max<size_t>(nl, 1) * 4 + 64, 0);
        if (r == hipSuccess) r = hipHostGetDevicePointer((void **)&d_jslot, ctx->h_stage, 0);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_LCOUNT, 64);
        if (r == hipSuccess && ctx->has_sizes) r = ensure_buf(ctx, SB_SEQSZ, (size_t)n * 4);
        if (r != hipSuccess) return false;
        // subscriber lists (count into FIRST as scratch, scan, fill, sort by leftover)
        r = hipMemsetAsync(buf<void>(ctx, SB_FIRST), 0, (size_t)ncl * 4, S);
        if (r == hipSuccess) r = launch_loop_subscribers(false, nl, buf<uint32_t>(ctx, SB_CSTART), buf<uint32_t>(ctx, SB_CNT), buf<GreedyCand>(ctx, SB_CAND),
                                                         buf<uint32_t>(ctx, SB_FIRST), nullptr, nullptr, S);
        if (r == hipSuccess) r = launch_scan_u32(buf<uint32_t>(ctx, SB_FIRST), buf<uint32_t>(ctx, SB_SUBSTART), ncl, buf<uint64_t>(ctx, SB_SCAN2), S);
        if (r == hipSuccess) r = hipMemsetAsync(buf<void>(ctx, SB_FIRST), 0, (size_t)ncl * 4, S);
        if (r == hipSuccess) r = launch_loop_subscribers(true, nl, buf<uint32_t>(ctx, SB_CSTART), buf<uint32_t>(ctx, SB_CNT), buf<GreedyCand>(ctx, SB_CAND),
                                                         buf<uint32_t>(ctx, SB_FIRST), buf<uint32_t>(ctx, SB_SUBSTART), buf<uint64_t>(ctx, SB_SUBS), S);
        if (r == hipSuccess) r = launch_loop_sort_subscribers(ncl, buf<uint32_t>(ctx, SB_SUBSTART), buf<uint64_t>(ctx, SB_SUBS), buf<uint64_t>(ctx, SB_SUBS2), S);
        uint32_t *d_first = buf<uint32_t>(ctx, SB_FIRST), *d_taken = d_first + ncl, *d_clcursor = d_first + 2 * (size_t)ncl;
        // (taken[], the statuses, the join slots and the counters are set by k_loop_init)
        if (r == hipSuccess) r = hipMemcpyAsync(buf<void>(ctx, SB_CSIZE), csize.data(), (size_t)ncl * 8, hipMemcpyHostToDevice, S);
        if (r == hipSuccess) r = hipMemcpyAsync(buf<void>(ctx, SB_CID), cids.data(), (size_t)ncl * 4, hipMemcpyHostToDevice, S);
        if (r == hipSuccess) r = launch_loop_init(ncl, buf<long long>(ctx, SB_CSIZE), buf<int32_t>(ctx, SB_CID), buf<void>(ctx, SB_JOINED),
                                                  buf<uint32_t>(ctx, SB_SUBSTART), d_clcursor, nl, buf<uint32_t>(ctx, SB_ACTIVE),
                                                  buf<uint32_t>(ctx, SB_DIRTY), buf<uint32_t>(ctx, SB_LCOUNT), d_taken,
                                                  buf<uint8_t>(ctx, SB_STATUS), d_jslot, S);
        if (r == hipSuccess && ctx->has_sizes)
            r = hipMemcpyAsync(buf<void>(ctx, SB_SEQSZ), ctx->sizes.data(), (size_t)n * 4, hipMemcpyHostToDevice, S);
        uint32_t rounds = 0;
        bool done = false;
        // a second first/accept pass per round saves a third of the rounds; it pays once a round's apply and eval are big enough
        // first/accept passes per round: an accepted leftover stops blocking the other clusters it lists, so a second pass lets more joins into
        // the round -- fewer, longer rounds.  Measured (loop ms, generator / default order): 10,000 clusters 3.46 / 4.58 with two passes, 3.66 / 4.72
        // with three; 15,000: 6.93 / 9.59 against 7.04 / 9.68; 25,000 (10^6 sequences): 19.3 / 29.6 against 19.0 / 27.9 (one pass: 36.8 in the default order)
        const int accept_passes = sw.loop_passes > 0 ? sw.loop_passes : ncl >= 20000 ? 3 : ncl >= 8192 ? 2 : 1;
        // Every round accepts at least the earliest open leftover that has a feasible cluster, so nl + 1 rounds always suffice
        // and a round without a join is the end.  The host keeps enqueuing rounds while it watches the progress word that
        // k_loop_apply stores into pinned host memory (round << 32 | joins of that round), at most LOOKAHEAD rounds ahead of
        // the device; rounds enqueued after the end find nothing to do.
        // where the joiners' rows are read: this device's CSR, or (multi-device) every piece where it lives -- in the peers' memory, or in
        // the copies the root made of their pieces (no peer access to that device; HMK_MULTI_REPLICATE)
        RowPieces rowp{};
        rowp.rows_per = multi ? src.rows_per : 0;
        for (size_t d = 0; d < pieces.size() && r == hipSuccess; d++) {
            hmk_ctx *pc = pieces[d].c;
            rowp.start[d] = buf<uint64_t>(pc, SB_START);
            rowp.up[d] = buf<uint32_t>(pc, SB_CURSOR);
            rowp.adj[d] = buf<void>(pc, SB_ADJ);
            if (multi && d > 0 && (!pc->peer_loads_ok || sw.multi_replicate)) {
                const uint64_t entries = pc->h_counts[HC_TOTAL];
                const size_t o_start = 0, o_up = ((size_t)n + 1) * 8, o_adj = (o_up + (size_t)n * 4 + 63) / 64 * 64, bytes = o_adj + std::max<uint64_t>(entries, 1) * esz;
                DevBuf &rb = pc->sb[SB_REPL];   // (allocated on the ROOT's device, kept in the peer's context)
                if (rb.cap < bytes) {
                    if (rb.p) (void)hipFree(rb.p);
                    rb.p = nullptr; rb.cap = 0;
                    r = hipMalloc(&rb.p, bytes + bytes / 8);
                    if (r == hipSuccess) rb.cap = bytes + bytes / 8;
                }
                char *rp = (char *)rb.p;
                if (r == hipSuccess) r = hipMemcpyPeerAsync(rp + o_start, ctx->device, rowp.start[d], pc->device, ((size_t)n + 1) * 8, S);
                if (r == hipSuccess) r = hipMemcpyPeerAsync(rp + o_up, ctx->device, rowp.up[d], pc->device, (size_t)n * 4, S);
                if (r == hipSuccess && entries) r = hipMemcpyPeerAsync(rp + o_adj, ctx->device, rowp.adj[d], pc->device, entries * esz, S);
                rowp.start[d] = (const uint64_t *)(rp + o_start);
                rowp.up[d] = (const uint32_t *)(rp + o_up);
                rowp.adj[d] = rp + o_adj;
            }
        }
        if (r != hipSuccess) return false;
        auto one_round = [&]() {
            r = launch_loop_round(packed, rowp,
                                  buf<uint32_t>(ctx, SB_LEFT), nl, buf<uint32_t>(ctx, SB_CSTART), buf<uint32_t>(ctx, SB_CNT),
                                  buf<GreedyCand>(ctx, SB_CAND), buf<uint8_t>(ctx, SB_STATUS), buf<uint32_t>(ctx, SB_CHOICE),
                                  buf<uint32_t>(ctx, SB_ACTIVE), buf<uint32_t>(ctx, SB_DIRTY), rounds, d_first, d_taken, d_clcursor,
                                  ncl, accept_passes, buf<uint32_t>(ctx, SB_ACCEPTED), d_jslot,
                                  buf<uint32_t>(ctx, SB_SUBSTART), buf<uint64_t>(ctx, SB_SUBS), buf<void>(ctx, SB_JOINED),
                                  ctx->has_sizes ? buf<int32_t>(ctx, SB_SEQSZ) : nullptr, buf<uint32_t>(ctx, SB_LCOUNT), ctx->h_loop, sw.loop_chain, S);
            rounds++;
        };
        if (nl == 0 || ncl == 0) {
            done = true;
        } else {
            const uint32_t LOOKAHEAD = 2;   // a round is 3-5 small dependent kernels; one round in the queue beside the running one keeps the device busy
                                            // (1 / 4 rounds in flight measured the same, 3.62 / 3.64 ms on the antibodies example) -- and every round enqueued
                                            // past the last one is 25 us the call waits for at its end
            volatile unsigned long long *word = ctx->h_loop;
            *word = 0;
            // never spin forever: the deadline runs from the last round the device was SEEN to finish (a long loop is fine, a
            // stalled device is not) and is looked at on every poll (a few thousand spins apart)
            auto t_progress = std::chrono::steady_clock::now();
            uint32_t last_seen = 0;
            bool stalled = false;
            while (r == hipSuccess && !done && rounds <= nl + 8) {
                one_round();
                for (uint32_t spins = 0;; spins++) {
                    const unsigned long long w = *word;
                    const uint32_t seen = (uint32_t)(w >> 32);      // rounds the device has finished
                    if (seen && (uint32_t)w == 0) { done = true; break; }
                    if (rounds - seen < LOOKAHEAD) break;
                    if (seen != last_seen) {
                        last_seen = seen;
                        t_progress = std::chrono::steady_clock::now();
                    }
                    else if ((spins & 1023u) == 1023u && ms_since(t_progress) > 60e3) { stalled = true; break; }
                    std::this_thread::yield();
                }
                if (stalled) break;
            }
            if (stalled) {
                // No k_loop_* kernel may still be writing cand[] or the progress word when the host path takes over -- but a
                // device that made no progress for a minute may never drain, and a blocking synchronise would spin forever
                // after all: poll for ten more seconds, then give the call up (HMK_ERR_DEVICE) instead of falling back.
                const auto t_drain = std::chrono::steady_clock::now();
                hipError_t q = hipStreamQuery(S);
                while (q == hipErrorNotReady && ms_since(t_drain) < 10e3) {
                    std::this_thread::sleep_for(std::chrono::milliseconds(5));
                    q = hipStreamQuery(S);
                }
                if (q == hipErrorNotReady) {
                    ctx->wedged = true;
                    status_inside = HMK_ERR_DEVICE;
                    hook_err = "the device made no progress for 70 s inside the second loop: call given up (the context is unusable)";
                }
                r = hipErrorNotReady;
            }
            if (r == hipSuccess && !done) {                         // (only when nl + 8 rounds were not enough: impossible)
                r = hipStreamSynchronize(S);
                done = r == hipSuccess && (uint32_t)*word == 0;
            }
            // (the round enqueued past the last one changes nothing -- a round without a join is the end -- and is not waited for here:
            // every join's slot was stored by a kernel that ended before the final round's began; cluster_on_device drains S before it returns)
        }
        if (r != hipSuccess || !done) return false;
        join_slot.resize(nl);
        if (nl) std::memcpy(join_slot.data(), ctx->h_stage, (size_t)nl * 4);   // (the stream is drained: every store has landed)
        if (r != hipSuccess) return false;
        ph.device_loop_ms = ms_since(tl);
        ph.loop_rounds = rounds;
        lap("device second loop (rounds)");
        return true;
    };

    hooks.precheck = [&](const int32_t *cluster_of, const std::vector<int32_t> &usize, const std::vector<uint32_t> &leftover,
                         std::vector<uint32_t> &cand_start, std::vector<GreedyCand> &cand) -> bool {
        if (!device_precheck(cluster_of, usize, leftover, false)) return false;
        if (!fetch_cand((uint32_t)leftover.size(), cand_start, cand)) return false;
        lap("device pre-check");
        return true;
    };

    hooks.adj_base = [&]() -> const void * { return (const char *)ctx->h_adj - adj_off * esz; };
    GreedyTimes times{};
    hooks.times = &times;
    std::string err;
    const int32_t *szs = ctx->has_sizes ? ctx->sizes.data() : nullptr;
    if (src.clink) {
        // clinkage mode: the chain needs every row; fetch the whole adjacency, then run it on the host
        int cst = HMK_OK;
        if (!src.format_known && !wait_full()) cst = -1;
        if (cst == HMK_OK && n && hooks.need_rows(n - 1, 0) < n) cst = -1;
        if (cst == HMK_OK)
            cst = packed ? clinkage_from_csr_packed(ctx->java_hashset, n, szs, h_start, (const NbrPacked *)ctx->h_adj, cluster_id, result_order, member_rank,
                                                    src.clink, &err)
                         : clinkage_from_csr(ctx->java_hashset, n, szs, h_start, (const Nbr *)ctx->h_adj, cluster_id, result_order, member_rank, src.clink,
                                             &err);
        (void)hipStreamSynchronize(S);
        (void)hipStreamSynchronize(C);
        if (status_inside == ST_RETRY_OVERFLOW) return ST_RETRY_OVERFLOW;
        if (status_inside != HMK_OK) return fail(ctx, status_inside, hook_err);
        if (cst) return fail(ctx, cst < 0 ? HMK_ERR_DEVICE : cst, err.empty() ? "clinkage: adjacency hand-over failed" : err);
        src.clink->n_edges = src.total_known ? src.total_known : h_start[n] / 2;
        return HMK_OK;
    }
    // the entry format is fixed before the merge starts unless it depends on the scores (then the first need_rows
    // call settles it through wait_full(), before any row is read): dispatch on a flag the row provider may update
    int st;
    if (!src.format_known) {
        if (!wait_full()) {
            (void)hipStreamSynchronize(S);
            (void)hipStreamSynchronize(C);
            return status_inside == ST_RETRY_OVERFLOW ? ST_RETRY_OVERFLOW : fail(ctx, status_inside, hook_err);
        }
    }
    const GreedyOptions gopt = greedy_options(ctx);
    st = packed ? greedy_from_csr_packed(n, szs, h_start, (const NbrPacked *)ctx->h_adj, symmetric ? h_up : nullptr, &hooks, symmetric,
                                         max_clusters, cluster_id, result_order, member_rank, stats, &err, gopt)
                : greedy_from_csr(n, szs, h_start, (const Nbr *)ctx->h_adj, symmetric ? h_up : nullptr, &hooks, symmetric, max_clusters,
                                  cluster_id, result_order, member_rank, stats, &err, gopt);
    // nothing of this call may still be running when the buffers are reused (a crash-parity exit leaves the pass in flight)
    if (!ctx->wedged) {
        (void)hipStreamSynchronize(S);
        (void)hipStreamSynchronize(C);
    }
    if (status_inside == ST_RETRY_OVERFLOW) return ST_RETRY_OVERFLOW;
    if (status_inside != HMK_OK) return fail(ctx, status_inside, hook_err);
    if (st == HMK_OK || st == HMK_ERR_REFERENCE_WOULD_CRASH) {
        // a crash-parity exit during phase 1 never looked at the final counts: an overflow must still be noticed
        if (src.check_overflow)
            for (int q = 0; q < HMK_EDGE_SHARDS; q++)
                if (ctx->h_counts[q] > src.seg_cap) return ST_RETRY_OVERFLOW;
    }
    ph.phase1_ms = times.phase1_ms;
    ph.sequential_ms = times.sequential_ms;
    ph.wait_rows_ms = t_rows;
    ph.host_precheck_ms = times.host_precheck_ms;
    stats->n_edges = src.total_known ? src.total_known : (symmetric ? h_start[n] / 2 : h_start[n]);
    if (st) return fail(ctx, st, err);
    return HMK_OK;
}

} }  // namespace hmk::impl
