// hmk_common.cpp -- what every part of the host side uses: error text, the device check, the 32-byte residue copy of the probes,
// the context's grow-only device and pinned buffers, the clustering calls' streams and events, buffer sizing.
#include "hmk_ctx.h"

namespace hmk { namespace impl {

thread_local std::string g_last_error;

// The environment switches, read once per entry-point call (hmk_ctx.h; INTEGRATION.md "Environment switches").
void Switches::read() {
    auto flag = [](const char *name) { return getenv(name) != nullptr; };
    auto num = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
    *this = Switches();
    greedy_timing = flag("HMK_GREEDY_TIMING");
    no_band = flag("HMK_NO_BAND");
    no_rows_kernel = flag("HMK_NO_ROWS_KERNEL");
    adj_8byte = flag("HMK_ADJ_8BYTE");
    local_literal = flag("HMK_LOCAL_LITERAL");
    local_signed = flag("HMK_LOCAL_SIGNED");
    local_no_pk = flag("HMK_LOCAL_NO_PK");
    multi_replicate = flag("HMK_MULTI_REPLICATE");
    if (const char *v = getenv("HMK_SECOND_LOOP")) second_loop = std::strcmp(v, "device") == 0 ? 1 : 2;
    phase1_threads = std::max(0, num("HMK_PHASE1_THREADS", 0));
    phase1_window = std::max(0, num("HMK_PHASE1_WINDOW", 0));
    if (const char *v = getenv("HMK_PHASE1_HOST_BAND")) {
        host_band_rows = std::max(0, atoi(v));
        if (const char *c = std::strchr(v, ',')) host_band_far_t = std::max(0, atoi(c + 1));
    }
    loop_chain = getenv("HMK_LOOP_CHAIN") ? (num("HMK_LOOP_CHAIN", 0) != 0 ? 1 : 0) : -1;
    loop_passes = std::min(8, std::max(0, num("HMK_LOOP_PASSES", 0)));
    if (const char *v = getenv("HMK_PRECHECK")) precheck = std::strcmp(v, "two_passes") == 0 ? 1 : std::strcmp(v, "one_stage") == 0 ? 2 : 0;
    late_buffers_delay_ms = std::max(0, num("HMK_LATE_BUFFERS_DELAY_MS", 0));
    csr_bucket_shift = num("HMK_CSR_BUCKET_SHIFT", 0);
    if (const char *v = getenv("HMK_EDGE_GUESS")) edge_guess = std::strtoull(v, nullptr, 10);
}
void refresh_switches(hmk_ctx *ctx) {
    ctx->sw.read();
    for (hmk_ctx *peer : ctx->peers) peer->sw = ctx->sw;
}
GreedyOptions greedy_options(const hmk_ctx *ctx) {
    GreedyOptions o;
    o.phase1_threads = ctx->sw.phase1_threads;
    o.phase1_window = ctx->sw.phase1_window;
    o.timing = ctx->sw.greedy_timing;
    o.host_band_rows = ctx->sw.host_band_rows;
    o.host_band_far_t = ctx->sw.host_band_far_t;
    return o;
}

int fail(hmk_ctx *ctx, int code, const std::string &msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}
int need_device(hmk_ctx *ctx) {
    if (!ctx->has_device)
        return fail(ctx, HMK_ERR_DEVICE,
                    "this context has no GPU (created with device = -1); scoring has no CPU fallback");
    if (ctx->wedged) return fail(ctx, HMK_ERR_DEVICE, "an earlier call on this context gave up on a device that had stopped making progress");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, HMK_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    // The launch wrappers report hipGetLastError(), which is the thread's last error from ANY earlier call -- e.g. one whose
    // result was deliberately ignored while another context was torn down.  What this context does starts with a clean slate.
    (void)hipGetLastError();
    return HMK_OK;
}

// The 32-byte-per-sequence copy of the residues (and the lengths) that the one-pair and block scorers index: built and
// uploaded at their first use -- the neighbour passes and the clustering calls never read it (they use the plan's sorted
// copy), and at 10^6 sequences it is 32 MB to build and send in every hmk_set_sequences.
int ensure_res32(hmk_ctx *ctx) {
    if (ctx->d_res32 || ctx->n == 0) return HMK_OK;
    const uint32_t n = ctx->n;
    std::vector<uint8_t> res32((size_t)n * 32, 0);
    for (uint32_t k = 0; k < n; k++) std::memcpy(&res32[(size_t)k * 32], ctx->res.data() + ctx->off[k], ctx->len[k]);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_res32, res32.size()));
    HIPCHK(ctx, hipMemcpy(ctx->d_res32, res32.data(), res32.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_len, n));
    HIPCHK(ctx, hipMemcpy(ctx->d_len, ctx->len.data(), n, hipMemcpyHostToDevice));
    return HMK_OK;
}
thread_local double g_alloc_ms = 0.0;
thread_local int g_allocs = 0;

hipError_t ensure_buf_now(hmk_ctx *ctx, int which, size_t bytes) {
    DevBuf &b = ctx->sb[which];
    if (b.cap >= bytes) return hipSuccess;
    AllocTimer at("hipMalloc", bytes);
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + (bytes < (1ull << 30) ? bytes / 8 : 0) + 256;   // (head room for the small ones only)
    const hipError_t e = hipMalloc(&b.p, want);
    if (e == hipSuccess) b.cap = want;
    return e;
}
bool late_buffers_pending(hmk_ctx *ctx) {
    return ctx->late_buffers.valid() && ctx->late_buffers.wait_for(std::chrono::seconds(0)) != std::future_status::ready;
}
hipError_t join_late_buffers(hmk_ctx *ctx) {
    if (!ctx->late_buffers.valid()) return hipSuccess;
    const hipError_t e = ctx->late_buffers.get();
    if (e != hipSuccess) (void)hipGetLastError();   // (the caller's own ensure_buf tries again and reports)
    return hipSuccess;
}
hipError_t ensure_buf(hmk_ctx *ctx, int which, size_t bytes) {
    if ((which == SB_ADJ || which == SB_PART) && ctx->late_buffers.valid()) (void)join_late_buffers(ctx);
    return ensure_buf_now(ctx, which, bytes);
}

// pinned host buffer, grow-only; the first `keep` bytes survive a reallocation
hipError_t ensure_pinned(void **p, size_t *cap, size_t bytes, size_t keep) {
    if (*cap >= bytes) return hipSuccess;
    AllocTimer at("hipHostMalloc", bytes + bytes / 4 + (1 << 20));
    void *q = nullptr;
    const size_t want = bytes + bytes / 4 + (1 << 20);
    const hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (*p) {
        if (keep) std::memcpy(q, *p, keep);
        (void)hipHostFree(*p);
    }
    *p = q;
    *cap = want;
    return hipSuccess;
}

int greedy_streams(hmk_ctx *ctx) {
    if (ctx->gstream) return HMK_OK;
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->gstream, hipStreamNonBlocking));
    // the band hand-over runs while the rest of the pair space is being scored: its small kernels must not queue behind
    // the thousands of workgroups of that launch, so its stream gets the highest priority
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio_hi));
    for (hipEvent_t *ev : {&ctx->ev_t0, &ctx->ev_band, &ctx->ev_edges, &ctx->ev_csr, &ctx->ev_bandcsr}) HIPCHK(ctx, hipEventCreate(ev));
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_counts, HC_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    // fine-grained, so that a system-scope store of a running kernel is seen by the polling host (no such block: batches + syncs)
    if (hipHostMalloc((void **)&ctx->h_loop, 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { ctx->h_loop = nullptr; (void)hipGetLastError(); }
    // The first device-to-host copy of more than a few KB on a stream sets up its DMA path: 8-9 ms, measured in the middle of
    // a first clustering call (the band's row starts).  One 64 KB copy through each stream now.
    HIPCHK(ctx, ensure_buf(ctx, SB_DEG, 1 << 20));
    HIPCHK(ctx, ensure_pinned(&ctx->h_start, &ctx->h_start_cap, 2 * 65536, 0));
    {   // ... and the first blocking upload from pageable memory its staging buffers (hmk_set_sequences: 8 of its 10 ms)
        std::vector<char> pageable(1 << 20, 0);
        HIPCHK(ctx, hipMemcpy(buf<void>(ctx, SB_DEG), pageable.data(), pageable.size(), hipMemcpyHostToDevice));
    }
    for (hipStream_t q : {ctx->gstream, ctx->copy_stream})
        HIPCHK(ctx, hipMemcpyAsync((char *)ctx->h_start + (q == ctx->gstream ? 0 : 65536), buf<void>(ctx, SB_DEG), 65536, hipMemcpyDeviceToHost, q));
    HIPCHK(ctx, hipStreamSynchronize(ctx->gstream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    return HMK_OK;
}
// The CSR scatter with its lower sections dealt by bucket (k_edges.hip, k_lower_*): for graphs whose scatter is bound by random
// writes.  Packed symmetric adjacency, at every size (10^5: CSR 0.44 -> 0.31 ms, 10^6: 58 -> 23 ms); asymmetric matrices and
// 8-byte entries scatter with atomics (k_edge_scatter).
bool csr_by_bucket(bool symmetric, bool packed) { return symmetric && packed; }
// The grow-only device and pinned buffers the tail of a clustering call on n sequences asks for (the edge buffer must have
// its size already): hmk_greedy_cluster before it enqueues the pass, hmk_reserve from a host that knows n early.
int reserve_tail_buffers(hmk_ctx *ctx, uint32_t n, bool packed, uint32_t r1, bool full, bool late_on_a_thread) {
    const size_t esz0 = packed ? sizeof(NbrPacked) : sizeof(Nbr);
    const size_t adj_bytes = std::max<uint64_t>((ctx->symmetric ? 2 : 1) * ctx->d_edges_cap, 1) * esz0;
    const size_t part_bytes = csr_by_bucket(ctx->symmetric, packed) ? (ctx->d_edges_cap + 1) * 8 : 0;
    const bool late = late_on_a_thread || late_buffers_pending(ctx);   // (pending: the call's CSR step joins the thread and checks the sizes)
    if (!late) HIPCHK(ctx, ensure_buf(ctx, SB_ADJ, adj_bytes));
    HIPCHK(ctx, ensure_buf(ctx, SB_DEG, (size_t)n * 8));   // (upper and lower counts of the fused pass)
    HIPCHK(ctx, ensure_buf(ctx, SB_CURSOR, (size_t)n * 8));
    HIPCHK(ctx, ensure_buf(ctx, SB_START, ((size_t)n + 1) * 8));
    HIPCHK(ctx, ensure_buf(ctx, SB_SCAN, scan_scratch_bytes(n)));
    HIPCHK(ctx, ensure_buf(ctx, SB_RANGE, 64));
    HIPCHK(ctx, ensure_pinned(&ctx->h_start, &ctx->h_start_cap, ((size_t)n + 1) * 8 + (size_t)n * 4 + 64, 0));
    if (r1) {
        HIPCHK(ctx, ensure_buf(ctx, SB_BDEG, (size_t)r1 * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_BCURSOR, (size_t)r1 * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_BSTART, ((size_t)r1 + 1) * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_BSCAN, scan_scratch_bytes(r1)));
        HIPCHK(ctx, ensure_buf(ctx, SB_BRANGE, 64));
    }
    if (part_bytes) {
        if (!late) HIPCHK(ctx, ensure_buf(ctx, SB_PART, part_bytes));
        HIPCHK(ctx, ensure_buf(ctx, SB_PARTSCR, csr_partition_scratch_bytes()));
    }
    HIPCHK(ctx, ensure_buf(ctx, SB_COF, (size_t)n * 4));
    HIPCHK(ctx, ensure_buf(ctx, SB_BITMAP, ((size_t)n + 31) / 32 * 4));
    HIPCHK(ctx, ensure_buf(ctx, SB_LEFT, (size_t)n * 4));
    HIPCHK(ctx, ensure_buf(ctx, SB_CNT, (size_t)n * 4));
    HIPCHK(ctx, ensure_buf(ctx, SB_CSTART, ((size_t)n + 1) * 4));
    HIPCHK(ctx, ensure_buf(ctx, SB_CAND, (size_t)n * 24 * sizeof(GreedyCand)));
    if (full) {
        // (hmk_reserve only: these are sized from data a call learns late -- estimates here, grown by the call if they fall short)
        const uint64_t avg_deg = n ? (ctx->symmetric ? 2 : 1) * ctx->d_edges_cap / n + 1 : 1;
        if (r1) {   // the band's adjacency: device + pinned host copy (0.8 GB at 10^6: the pinned allocation alone took 0.1 s of a first call)
            const uint64_t entries = (uint64_t)r1 * avg_deg;
            HIPCHK(ctx, ensure_buf(ctx, SB_BADJ, std::max<uint64_t>(entries, 1) * esz0));
            // (the prepared band: transposed far part on the device; near rows + travelling lists in the host's pinned block)
            HIPCHK(ctx, ensure_buf(ctx, SB_FADJ, std::max<uint64_t>(entries, 1) * 4));
            HIPCHK(ctx, ensure_buf(ctx, SB_FDEG, (size_t)n * 12 + 64));
            HIPCHK(ctx, ensure_buf(ctx, SB_FSTART, (size_t)n * 4));
            HIPCHK(ctx, ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, std::max<uint64_t>(entries, 1) * 2 * esz0 + (size_t)r1 * 64 + 4096, 0));
        }
        const size_t ncl = (size_t)(n * 0.025 + 2), nl = n, cands = (size_t)n * 16;   // second loop on the device
        HIPCHK(ctx, ensure_buf(ctx, SB_USIZE, ncl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_OVER, 64));
        HIPCHK(ctx, ensure_buf(ctx, SB_SCAN2, scan_scratch_bytes(std::max<uint32_t>(n, (uint32_t)std::min<size_t>(cands, 0xFFFFFFFFu)))));
        HIPCHK(ctx, ensure_buf(ctx, SB_PRECNT, HMK_PRE_REGIONS * sizeof(unsigned long long)));
        HIPCHK(ctx, ensure_buf(ctx, SB_RETRY, nl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_JOINED, ncl * 16));
        HIPCHK(ctx, ensure_buf(ctx, SB_SUBSTART, (ncl + 1) * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_SUBS, cands * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_SUBS2, cands * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_CSIZE, ncl * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_CID, ncl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_FIRST, ncl * 12));
        HIPCHK(ctx, ensure_buf(ctx, SB_STATUS, nl));
        HIPCHK(ctx, ensure_buf(ctx, SB_ACTIVE, nl * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_DIRTY, nl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_CHOICE, nl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_ACCEPTED, nl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_LCOUNT, 64));
        HIPCHK(ctx, ensure_buf(ctx, SB_SEQSZ, (size_t)n * 4));
        HIPCHK(ctx, ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, HMK_PRE_REGIONS * sizeof(unsigned long long) + (size_t)n * 12 + ncl * 4 + 64, 0));
    }
    if (late_on_a_thread) (void)join_late_buffers(ctx);   // (an earlier hmk_reserve's thread may still be writing the two sizes read next)
    if (late_on_a_thread && (ctx->sb[SB_ADJ].cap < adj_bytes || ctx->sb[SB_PART].cap < part_bytes)) {
        const int device = ctx->device;
        const int delay_ms = ctx->sw.late_buffers_delay_ms;   // tests: a host on which device memory is slow to get
        ctx->late_buffers = std::async(std::launch::async, [ctx, device, adj_bytes, part_bytes, delay_ms]() -> hipError_t {
            if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = ensure_buf_now(ctx, SB_ADJ, adj_bytes);
            if (e == hipSuccess && part_bytes) e = ensure_buf_now(ctx, SB_PART, part_bytes);
            return e;
        });
    }
    return HMK_OK;
}

uint64_t first_edge_capacity(const hmk_ctx *ctx, uint32_t n) {
    // first guess of the edge buffer: 0.3 % of the pair space (uniform random 12-mers at the default threshold give
    // 0.26 %); a segment that overflows makes the call size the buffer to the counts and score again
    uint64_t guess = (uint64_t)((double)n * (n - 1) / 2 * (ctx->symmetric ? 0.003 : 0.006)) + (1u << 20);
    if (ctx->sw.edge_guess) guess = ctx->sw.edge_guess;   // tests: force the overflow / retry path
    uint64_t cap = std::max<uint64_t>({std::min<uint64_t>(guess, 1ull << 31), (uint64_t)1 << 20, ctx->d_edges_cap});
    return (cap + HMK_EDGE_SHARDS - 1) / HMK_EDGE_SHARDS * HMK_EDGE_SHARDS;
}

int grow_edge_buffer(hmk_ctx *ctx, uint64_t cap) {
    if (ctx->d_edges_cap >= cap) return HMK_OK;
    if (ctx->d_edges) (void)hipFree(ctx->d_edges);
    ctx->d_edges = nullptr;
    ctx->d_edges_cap = 0;
    { AllocTimer at("hipMalloc (edges)", cap * sizeof(uint64_t)); HIPCHK(ctx, hipMalloc((void **)&ctx->d_edges, cap * sizeof(uint64_t))); }
    ctx->d_edges_cap = cap;
    return HMK_OK;
}

} }  // namespace hmk::impl
