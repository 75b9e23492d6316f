// hmk_api.cpp -- the C ABI of libhammock_hip.so (include/hammock_hip.h):
// context, sequence upload, neighbour-kernel planning, launches, host buffers.
// Host code only; the kernels live in k_*.hip (launchers declared in hmk_kernels.h).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <thread>
#include <vector>

#include "hmk_internal.h"
#include "hmk_kernels.h"

using namespace hmk;

namespace {

thread_local std::string g_last_error;

struct Group {
    int path;
    int nw;
    int lbk;  // column-length capacity of the kernel instantiation
    uint32_t base, count;
    uint32_t band;  // the first `band` tiles of the group touch a "band" row (caller index < Plan::band_rows)
};

// grow-only device scratch of the greedy tail (one hipMalloc per buffer and context, not per call)
struct DevBuf { void *p = nullptr; size_t cap = 0; };
enum {
    SB_DEG, SB_CURSOR, SB_START, SB_SCAN, SB_RANGE, SB_ADJ, SB_PART, SB_PARTSCR,   // full CSR (+ the bucketed lower sections)
    SB_BDEG, SB_BCURSOR, SB_BSTART, SB_BSCAN, SB_BRANGE, SB_BADJ, SB_BCOUNTS,     // band CSR (first rows only)
    SB_COF, SB_BITMAP, SB_USIZE, SB_LEFT, SB_CNT, SB_CSTART, SB_OVER, SB_SCAN2, SB_CAND,             // pre-check of the second loop
    SB_LIDX, SB_PCNT, SB_PSTART, SB_PROP,
    SB_JOINED, SB_CSIZE, SB_CID, SB_SEQSZ, SB_STATUS, SB_CHOICE, SB_FIRST, SB_ACTIVE, SB_DIRTY, SB_SUBS2, SB_RANK, SB_RETRY, SB_PRECNT, SB_ACCEPTED, SB_JSLOT, SB_LCOUNT, SB_SUBSTART, SB_SUBS,   // device-side second loop                                                 // join-propagation lists
    SB_BANDCTR,   // [0] band tiles done (band_tile_done), [1] k_wait_counter gave up
    SB_PEER, SB_PEERCNT, SB_PEERBAND, SB_PEERDEG,                                                     // edge blocks gathered from other devices (root) / compacted for the root (peers)
    SB_N
};

struct Plan {
    bool valid = false;
    int X = 0, p = 0, thr = 0;
    uint32_t part = 0, n_parts = 1;
    uint32_t band_rows = 0;   // tiles touching caller indices below this come first in every group (0: no band)
    int64_t band_req = 0;     // what the caller asked for (the plan may have had to drop the band)
    int lbmax = 12, lpad = 16;
    bool exact = false;       // the shift-packed length-12 kernel (k_neighbors_swar; only with HMK_NO_ROWS_KERNEL)
    bool rows_exact = false;  // one length for all and a row-packed instantiation for exactly that length
    int hot_variant = 7;
    uint32_t cols_per_tile = 16384;
    uint8_t *d_res_sorted = nullptr;
    uint32_t *d_perm = nullptr;
    bool perm_identity = false;
    uint8_t *d_mb = nullptr;
    TileClass *d_classes = nullptr;
    Tile *d_tiles = nullptr;
    std::vector<Group> groups;
    hmk_neighbor_stats stats{};
    uint64_t band_pairs = 0;   // pairs inside the band tiles (of stats.pairs_scored)
};

struct PlanLocal {
    bool valid = false;
    uint32_t part = 0, n_parts = 1;
    uint8_t *d_res_sorted = nullptr;
    uint32_t *d_perm = nullptr;
    bool perm_identity = false;
    TileClass *d_classes = nullptr;
    Tile *d_tiles = nullptr;
    uint32_t n_tiles = 0;
    uint64_t pairs = 0;
};

}  // namespace

struct hmk_ctx {
    int32_t M[HMK_ALPHABET * HMK_ALPHABET];
    bool symmetric = true;
    int min_m = 0, max_m = 0;
    int device = -1;
    bool has_device = false;
    int java_hashset = 8;   // hmk_set_java_hashset: whose HashSet iteration order clinkage emulates

    uint32_t n = 0;
    std::vector<uint8_t> res;
    std::vector<uint32_t> off;
    std::vector<int32_t> sizes;
    bool has_sizes = false;
    std::vector<uint8_t> len;
    int min_len = 0, max_len = 0;

    uint8_t *d_res32 = nullptr;
    uint8_t *d_len = nullptr;
    int32_t *d_M = nullptr;

    Plan plan;
    PlanLocal plan_local;
    uint64_t *d_edges = nullptr;  // internal buffer of the host-buffer entry points
    uint64_t d_edges_cap = 0;
    unsigned long long *d_counts = nullptr;
    // side streams of the neighbour pass: the per-class launches of a mixed-length plan overlap their tails
    static constexpr int N_SIDE = 8;     // created; HMK_SIDE_STREAMS (default 3) of them are used
    hipStream_t side[N_SIDE] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[N_SIDE] = {nullptr};
    hipStream_t copy_stream = nullptr;   // band CSR + device-to-host copies of adjacency rows (hmk_greedy_cluster)
    uint32_t *d_rows_scratch = nullptr;  // deg[n], cursor[n], misfit of hmk_pack_rows_dev
    uint32_t d_rows_scratch_n = 0;

    double last_kernel_ms = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // greedy tail: own stream + events, grow-only device scratch, pinned host staging (all made once per context)
    hipStream_t gstream = nullptr;
    hipStream_t rest_stream = nullptr;   // lowest priority: the tiles outside the band, scored beside the band tiles (hmk_greedy_cluster)
    hipEvent_t ev_rest = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_band = nullptr, ev_edges = nullptr, ev_csr = nullptr, ev_bandcsr = nullptr;
    DevBuf sb[SB_N];
    void *h_start = nullptr;  // pinned: uint64 start[n + 1], then uint32 up[n]
    size_t h_start_cap = 0;
    void *h_stage = nullptr;  // pinned: what the merge uploads after phase 1 (cluster_of, sizes, leftovers, ...)
    size_t h_stage_cap = 0;
    void *h_adj = nullptr;    // pinned: adjacency rows fetched so far
    size_t h_adj_cap = 0;
    unsigned long long *h_loop = nullptr;    // pinned, coherent: progress word of the device-side second loop (written by k_loop_apply)
    unsigned long long *h_counts = nullptr;  // pinned: final segment counts [HMK_EDGE_SHARDS], band snapshot [HMK_EDGE_SHARDS], misc (HC_* below)
    hmk_greedy_phases phases{};

    // hmk_create_multi: this context is the root (devices[0]); one sub-context per further device, each with its own
    // copy of the sequences, its plan (shard d of n) and its edge buffer.  Empty for a single-device context.
    std::vector<hmk_ctx *> peers;
    // (in a peer's sub-context, created on the ROOT device:) the stream its blocks travel to the root on and the events the
    // root's streams wait for
    hipStream_t gather_stream = nullptr;
    hipEvent_t ev_bandgather = nullptr, ev_gather = nullptr;

    // hmk_reserve sizes the two buffers a clustering call needs LAST (adjacency, bucket records: 2 x 11 GB at 10^6) on its own
    // thread: on some hosts a fresh 11 GB of device memory takes 0.3-1.5 s to get, and a call has 0.27 s of scoring to do
    // before it writes to them.  Whoever touches SB_ADJ / SB_PART joins this first (ensure_buf does).
    std::future<hipError_t> late_buffers;

    bool wedged = false;   // a clustering call gave a stalled device up: nothing waits for it any more (calls fail with HMK_ERR_DEVICE)
    std::string err;
    mutable std::mutex mu;
};

namespace {

int fail(hmk_ctx *ctx, int code, const std::string &msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, e_ == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE,          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

int greedy_streams(hmk_ctx *ctx);   // streams, events and pinned blocks of the clustering calls (defined with them below)
hipError_t join_late_buffers(hmk_ctx *ctx);

int need_device(hmk_ctx *ctx) {
    if (!ctx->has_device)
        return fail(ctx, HMK_ERR_DEVICE,
                    "this context has no GPU (created with device = -1); scoring has no CPU fallback");
    if (ctx->wedged) return fail(ctx, HMK_ERR_DEVICE, "an earlier call on this context gave up on a device that had stopped making progress");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, HMK_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
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
void classify(const hmk_ctx *ctx, int la, int lb, int X, int p, int thr, TileClass *out, long long row_bound = -1,
              long long *u8_row_limit = nullptr) {
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
int build_plan(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, int64_t band_rows = -1) {
    Plan &pl = ctx->plan;
    if (pl.valid && pl.X == X && pl.p == p && pl.thr == thr && pl.part == part && pl.n_parts == n_parts &&
        (band_rows < 0 || pl.band_req == band_rows))
        return HMK_OK;
    free_plan(pl);
    if (band_rows < 0) band_rows = 0;
    const int64_t band_req = band_rows;
    const bool plan_timing = getenv("HMK_PLAN_TIMING") != nullptr;
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
    if (getenv("HMK_NO_ROW_BOUNDS")) refine = false;
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
            all_rows_fit = getenv("HMK_ALWAYS_SORT_BOUNDS") == nullptr;
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
    const bool use_rows = getenv("HMK_NO_ROWS_KERNEL") == nullptr;
    pl.exact = false;
    pl.rows_exact = false;
    if (use_rows && ctx->min_len == ctx->max_len) {
        TileClass t1;
        classify(ctx, ctx->min_len, ctx->min_len, X, p, thr, &t1);
        // (8-bit lanes for any pair of the class, or -- by their score bounds -- for every row the set has)
        pl.rows_exact = (t1.path == PATH_U8 || (refine && all_rows_fit)) && rows_kernel_available(X, ctx->min_len, ctx->min_len, true) &&
                        getenv("HMK_NO_ROWS_EXACT") == nullptr;
    }
    if (!use_rows && ctx->min_len == 12 && ctx->max_len == 12 && X == 3) {
        TileClass t12;
        classify(ctx, 12, 12, X, p, thr, &t12);
        pl.exact = t12.path == PATH_U8 && t12.nw == 2;
    }
    // Tiling (measured on MI355X, tools/tune_hot.py): 6 rows x 2 columns per lane and long
    // column runs win (7 workgroups/CU, table build amortised); shrink the runs for small
    // inputs so the grid still has a few thousand workgroups.
    pl.hot_variant = 7;
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
    if (const char *v = getenv("HMK_HOT_VARIANT")) pl.hot_variant = atoi(v);   // tuning knobs
    if (const char *v = getenv("HMK_COLS_PER_TILE")) pl.cols_per_tile = (uint32_t)std::min(65536, std::max(256, atoi(v)));   // hit records hold a 16-bit column offset

    // ---- classes and tiles --------------------------------------------------------
    std::vector<TileClass> classes;
    std::map<int, int> class_of;  // la * 64 + lb
    std::map<std::tuple<int, int, int>, std::vector<Tile>> grouped;  // (path, nw, column capacity)
    const uint32_t COLS = pl.cols_per_tile;
    const bool equal_runs = getenv("HMK_NO_EQUAL_RUNS") == nullptr;
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
                                        : tc.path == PATH_DIRECT ? 16u : (uint32_t)swar_rows_per_tile(lbk, tc.nw, pl.exact, pl.hot_variant);
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
        if (getenv("HMK_NO_LPT") == nullptr)
            parallel_stable_sort(kv.second, [](const Tile &a, const Tile &b) {
                if (a.pad0 != b.pad0) return a.pad0 > b.pad0;
                return (uint64_t)a.nrows * a.ncols > (uint64_t)b.nrows * b.ncols;
            });
        else
            parallel_stable_sort(kv.second, [](const Tile &a, const Tile &b) { return a.pad0 > b.pad0; });
        uint32_t n_band = 0;
        for (const Tile &t : kv.second) n_band += t.pad0;
        pl.groups.push_back(Group{std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first),
                                  (uint32_t)tiles.size(), (uint32_t)kv.second.size(), n_band});
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

// which: LAUNCH_ALL, or only the band tiles of the plan (LAUNCH_BAND: also zeroes the counts) / only the others
// (LAUNCH_REST: appends to the counts of the band launch)
enum { LAUNCH_ALL = 0, LAUNCH_BAND = 1, LAUNCH_REST = 2, LAUNCH_BAND_NOZERO = 3 };   // (NOZERO: the caller has zeroed the counts)
int neighbors_dev_locked(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, void *d_edges,
                         uint64_t capacity, void *d_counts, hipStream_t stream, int which = LAUNCH_ALL,
                         int64_t band_rows = -1, uint32_t *d_deg = nullptr, uint32_t *d_deg_lo = nullptr, uint32_t *d_rank = nullptr,
                         uint32_t shard_base = 0, uint32_t shard_mod = HMK_EDGE_SHARDS, uint32_t band_mod = 0, uint32_t *band_counter = nullptr) {
    int st = need_device(ctx);
    if (st) return st;
    if (!d_edges || !d_counts || capacity < HMK_EDGE_SHARDS)
        return fail(ctx, HMK_ERR_BAD_ARG, "d_edges/d_counts must be device buffers, capacity >= HMK_EDGE_SHARDS");
    st = build_plan(ctx, X, p, thr, part, n_parts, band_rows);
    if (st) return st;
    Plan &pl = ctx->plan;
    if (which != LAUNCH_REST && which != LAUNCH_BAND_NOZERO)
        HIPCHK(ctx, hipMemsetAsync(d_counts, 0, HMK_EDGE_SHARDS * sizeof(unsigned long long), stream));
    if (which == LAUNCH_BAND_NOZERO) which = LAUNCH_BAND;
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
    P.deg = d_rank ? nullptr : d_deg;
    P.deg_up = d_rank ? d_deg : nullptr;
    P.deg_lo = d_rank ? d_deg_lo : nullptr;
    P.deg_m_offset = (!d_rank && d_deg && d_deg_lo) ? (uint32_t)(d_deg_lo - d_deg) : 0u;   // counting mode with split counters
    P.shard_base = shard_base;
    P.shard_mod = shard_mod;
    P.band_mod = band_mod;
    P.band_counter = band_counter;
    P.rank = d_rank;
    // one launch per (lane path, entry width, column capacity) group.  A mixed-length plan has a dozen of
    // them: fork them round-robin onto side streams so that one group's tail overlaps the next group's
    // start, and join back into `stream`.
    const bool fork = pl.groups.size() > 2 && getenv("HMK_NO_SIDE_STREAMS") == nullptr;
    int n_side = 3;
    if (const char *v = getenv("HMK_SIDE_STREAMS")) n_side = std::max(1, std::min((int)hmk_ctx::N_SIDE, atoi(v)));
    // The streams the launches are dealt to: the pass's own stream and n_side - 1 others.  A process gets few hardware queues
    // (4 by default), and streams beyond them share one and serialise: with the clustering calls' two streams created first
    // (hmk_create), three more side streams cost this pass 5 % (5.36 -> 5.65 ms on BASELINE config 4a).  So a pass that does
    // not run on the clustering stream borrows those two (idle: calls on a context are serialised); a clustering call without
    // a band borrows the copy stream and creates one side stream; one with a band, whose hand-over needs the copy stream for
    // itself, creates two.
    hipStream_t sides[hmk_ctx::N_SIDE] = {nullptr};
    if (fork) {
        if (!ctx->ev_fork) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        std::vector<hipStream_t> lend;
        if (getenv("HMK_OWN_SIDE_STREAMS") == nullptr && ctx->gstream && ctx->copy_stream) {
            if (stream != ctx->gstream && stream != ctx->copy_stream && stream != ctx->rest_stream) lend = {ctx->gstream, ctx->copy_stream};   // (rest_stream: a clustering call's second launch -- both are busy)
            else if (stream == ctx->gstream && which == LAUNCH_ALL) lend = {ctx->copy_stream};
        }
        int own = 0;
        sides[0] = stream;
        for (int k = 1; k < n_side; k++) {
            if ((size_t)(k - 1) < lend.size()) { sides[k] = lend[k - 1]; continue; }
            if (!ctx->side[own]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side[own], hipStreamNonBlocking));
            sides[k] = ctx->side[own++];
        }
        for (int k = 1; k < n_side; k++)
            if (!ctx->ev_join[k]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, stream));
        for (int k = 1; k < n_side; k++) HIPCHK(ctx, hipStreamWaitEvent(sides[k], ctx->ev_fork, 0));
    }
    // biggest groups first
    std::vector<const Group *> order;
    for (const Group &g : pl.groups) order.push_back(&g);
    std::stable_sort(order.begin(), order.end(), [](const Group *a, const Group *b) { return a->count > b->count; });
    size_t q = 0;
    for (const Group *gp : order) {
        const Group &g = *gp;
        hipStream_t s = fork ? sides[q++ % n_side] : stream;
        const uint32_t t0 = which == LAUNCH_REST ? g.base + g.band : g.base;
        const uint32_t cnt = which == LAUNCH_ALL ? g.count : which == LAUNCH_BAND ? g.band : g.count - g.band;
        if (g.path == PATH_DIRECT)
            HIPCHK(ctx, launch_neighbors_direct(P, t0, cnt, ctx->d_M, X, p, thr, s));
        else if (g.path == PATH_ROWS)
            HIPCHK(ctx, launch_neighbors_rows(X, g.nw, g.lbk, pl.rows_exact, P, t0, cnt, s));
        else
            HIPCHK(ctx, launch_neighbors_swar(g.lbk, g.nw, pl.exact, pl.hot_variant, P, t0, cnt, s));
    }
    if (fork)
        for (int k = 1; k < n_side; k++) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_join[k], sides[k]));
            HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[k], 0));
        }
    return HMK_OK;
}

// Runs the neighbour pass into the context's own device buffer, growing it until
// every segment fits, and returns the per-segment counts.
// the tagged-max SW kernels carry 4 * value + direction in int8 table bytes
bool local_enc(const hmk_ctx *ctx, int gap_open, int gap_extend) {
    return ctx->min_m >= -31 && ctx->max_m <= 31 && gap_open >= -31 && gap_extend >= -31 && gap_open <= 0 &&
           gap_extend <= 0 && getenv("HMK_LOCAL_PLAIN") == nullptr;
}

template <typename LaunchFn>
int neighbors_grow(hmk_ctx *ctx, uint64_t want_cap, unsigned long long counts[HMK_EDGE_SHARDS], double *kernel_ms,
                   LaunchFn launch) {
    int st = need_device(ctx);
    if (st) return st;
    if (!ctx->d_counts) HIPCHK(ctx, hipMalloc((void **)&ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long)));
    uint64_t cap = std::max<uint64_t>(want_cap, (uint64_t)1 << 20);
    cap = (cap + HMK_EDGE_SHARDS - 1) / HMK_EDGE_SHARDS * HMK_EDGE_SHARDS;
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    for (int attempt = 0; attempt < 4; attempt++) {
        if (ctx->d_edges_cap < cap) {
            if (ctx->d_edges) (void)hipFree(ctx->d_edges);
            ctx->d_edges = nullptr;
            ctx->d_edges_cap = 0;
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_edges, cap * sizeof(uint64_t)));
            ctx->d_edges_cap = cap;
        }
        HIPCHK(ctx, hipEventRecord(e0, nullptr));
        st = launch(ctx->d_edges, ctx->d_edges_cap, ctx->d_counts);
        if (st) break;
        HIPCHK(ctx, hipEventRecord(e1, nullptr));
        HIPCHK(ctx, hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms) *kernel_ms = ms;
        HIPCHK(ctx, hipMemcpy(counts, ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (int s = 0; s < HMK_EDGE_SHARDS; s++) mx = std::max(mx, counts[s]);
        if (mx <= ctx->d_edges_cap / HMK_EDGE_SHARDS) {
            st = HMK_OK;
            break;
        }
        cap = (uint64_t)HMK_EDGE_SHARDS * (mx + mx / 8 + 1024);  // a segment overflowed: grow and rescore
        st = HMK_ERR_CAPACITY;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (st == HMK_ERR_CAPACITY) return fail(ctx, HMK_ERR_DEVICE, "internal edge buffer kept overflowing");
    return st;
}

int neighbors_internal(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, uint64_t want_cap,
                       unsigned long long counts[HMK_EDGE_SHARDS], double *kernel_ms) {
    return neighbors_grow(ctx, want_cap, counts, kernel_ms, [&](uint64_t *d_edges, uint64_t cap, unsigned long long *d_counts) {
        return neighbors_dev_locked(ctx, X, p, thr, part, n_parts, d_edges, cap, d_counts, nullptr);
    });
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

int neighbors_local_dev_locked(hmk_ctx *ctx, int gap_open, int gap_extend, int thr, uint32_t part, uint32_t n_parts,
                               uint64_t *d_edges, uint64_t capacity, unsigned long long *d_counts, hipStream_t stream) {
    int st = need_device(ctx);
    if (st) return st;
    // the striped register kernels take gap penalties <= 0 and int8 matrix entries; anything else (the reference imposes
    // neither, LocalAlignmentScorer.java:43-55) runs the literal DP on the same tiles
    const bool literal = gap_open > 0 || gap_extend > 0 || ctx->min_m < -127 || ctx->max_m > 127 || getenv("HMK_LOCAL_LITERAL") != nullptr;
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
    P.shard_base = 0;
    P.shard_mod = HMK_EDGE_SHARDS;
    P.band_mod = 0;
    P.band_counter = nullptr;
    P.n_tiles = pl.n_tiles;
    P.lpad = 32;
    P.symmetric = 0;
    P.row_is_m = 1;
    if (literal)
        HIPCHK(ctx, launch_neighbors_local_literal(P, 0, pl.n_tiles, ctx->d_M, gap_open, gap_extend, thr, stream));
    else
        HIPCHK(ctx, launch_neighbors_local(ctx->max_len, local_enc(ctx, gap_open, gap_extend), P, 0, pl.n_tiles, ctx->d_M, gap_open, gap_extend, thr, stream));
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
                int32_t *out, int32_t *out_shift = nullptr) {
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    const bool fast_local = scorer == 1 && a <= 0 && b <= 0 && ctx->min_m >= -127 && ctx->max_m <= 127 &&
                            getenv("HMK_LOCAL_LITERAL") == nullptr;
    if (fast_local)
        e = launch_local_block(ctx->max_len, local_enc(ctx, a, b), ctx->d_res32, ctx->d_len, ctx->d_M, r0, r1, c0, c1, a, b, d_out, nullptr);
    else
        e = launch_pairs(scorer, ctx->d_res32, ctx->d_len, ctx->d_M, nullptr, nullptr, n_pairs, r0, c0, c1 - c0, a, b,
                         d_out, nullptr, nullptr);
    timer_stop(ctx);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n_pairs * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, HMK_ERR_DEVICE, std::string("score_block: ") + hipGetErrorString(e));
    return HMK_OK;
}

}  // namespace

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
    const bool timing = getenv("HMK_CLI_TIMING") != nullptr || getenv("HMK_GREEDY_TIMING") != nullptr;
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
        if (getenv("HMK_LAZY_CONTEXT") == nullptr) {
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

void hmk_destroy(hmk_ctx *ctx) {
    if (!ctx) return;
    for (hmk_ctx *peer : ctx->peers) hmk_destroy(peer);
    ctx->peers.clear();
    if (ctx->has_device) {
        (void)hipSetDevice(ctx->device);
        (void)join_late_buffers(ctx);
        free_plan(ctx->plan);
        free_plan_local(ctx->plan_local);
        if (ctx->d_res32) (void)hipFree(ctx->d_res32);
        if (ctx->d_len) (void)hipFree(ctx->d_len);
        if (ctx->d_M) (void)hipFree(ctx->d_M);
        if (ctx->d_edges) (void)hipFree(ctx->d_edges);
        if (ctx->d_counts) (void)hipFree(ctx->d_counts);
        if (ctx->d_rows_scratch) (void)hipFree(ctx->d_rows_scratch);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->rest_stream) (void)hipStreamDestroy(ctx->rest_stream);
        if (ctx->ev_rest) (void)hipEventDestroy(ctx->ev_rest);
        for (int k = 0; k < hmk_ctx::N_SIDE; k++) {
            if (ctx->side[k]) (void)hipStreamDestroy(ctx->side[k]);
            if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
        }
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        if (ctx->h_start) (void)hipHostFree(ctx->h_start);
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        if (ctx->h_adj) (void)hipHostFree(ctx->h_adj);
        if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
        if (ctx->h_loop) (void)hipHostFree(ctx->h_loop);
        for (int b = 0; b < SB_N; b++)
            if (ctx->sb[b].p) (void)hipFree(ctx->sb[b].p);
        if (ctx->gstream) (void)hipStreamDestroy(ctx->gstream);
        for (hipEvent_t ev : {ctx->ev_t0, ctx->ev_band, ctx->ev_edges, ctx->ev_csr, ctx->ev_bandcsr})
            if (ev) (void)hipEventDestroy(ev);
        if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
        if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    }
    delete ctx;
}

int hmk_set_sequences(hmk_ctx *ctx, const uint8_t *residues, const uint32_t *offsets, const int32_t *sizes,
                      uint32_t n) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    return neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, part, n_parts, d_edges, capacity, d_counts,
                                (hipStream_t)stream);
}

int hmk_compact_edges_dev(hmk_ctx *ctx, const void *d_edges, uint64_t capacity, const void *d_counts, void *d_out,
                          uint64_t out_capacity, void *d_total, void *stream) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    if (!ctx->plan.valid) return fail(ctx, HMK_ERR_BAD_ARG, "no neighbour pass has been planned yet");
    *stats = ctx->plan.stats;
    return HMK_OK;
}

int hmk_neighbors_shifted(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, uint32_t part,
                          uint32_t n_parts, uint64_t *edges, uint64_t capacity, uint64_t *n_edges,
                          hmk_neighbor_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    if (ctx->n && !cluster_id) return fail(ctx, HMK_ERR_BAD_ARG, "null cluster_id");
    if (n_edges && !edges) return fail(ctx, HMK_ERR_BAD_ARG, "null edge list");
    std::string err;
    int st = greedy_from_edges(ctx->n, ctx->has_sizes ? ctx->sizes.data() : nullptr, edges, n_edges, symmetric != 0,
                               threshold, max_clusters, cluster_id, result_order, member_rank, stats, &err);
    if (st) return fail(ctx, st, err);
    return HMK_OK;
}

// =============================================================================
// greedy clustering on a device-resident neighbour graph
// =============================================================================
}  // extern "C"

namespace {

constexpr int ST_RETRY_OVERFLOW = 1000;   // internal: an edge segment overflowed, grow the buffer and score again
// layout of the small pinned block hmk_ctx::h_counts (64-bit words)
enum { HC_COUNTS = 0, HC_BAND = HMK_EDGE_SHARDS, HC_PEER = 2 * HMK_EDGE_SHARDS, HC_RANGE = HC_PEER + 32, HC_MISC = HC_RANGE + 8, HC_TOTAL = HC_MISC + 8, HC_WORDS = HC_TOTAL + 16 };

// (HMK_GREEDY_TIMING: what the grow-only buffers cost a call, i.e. the first call of a context)
static thread_local double g_alloc_ms = 0.0;
static thread_local int g_allocs = 0;
struct AllocTimer {
    const char *what;
    size_t bytes;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    AllocTimer(const char *w, size_t b) : what(w), bytes(b) {}
    ~AllocTimer() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        g_alloc_ms += ms;
        g_allocs++;
        static const bool timing = getenv("HMK_CLI_TIMING") != nullptr || getenv("HMK_GREEDY_TIMING") != nullptr;
        if (timing && ms > 5.0) std::fprintf(stderr, "[hmk] %s of %.1f MB took %.1f ms\n", what, (double)bytes / 1048576.0, ms);
    }
};

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
template <class T> T *buf(hmk_ctx *ctx, int which) { return (T *)ctx->sb[which].p; }

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
    // HMK_CU_RESERVE=k: the clustering stream may not use k of the device's CUs (a CU mask), so that the small kernels of the
    // band hand-over, on their own stream, find a free CU at once instead of waiting for a workgroup of the scoring pass to end
    int reserve = 0;
    if (const char *v = getenv("HMK_CU_RESERVE")) reserve = std::max(0, std::min(64, atoi(v)));
    if (reserve > 0) {
        hipDeviceProp_t prop;
        HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
        const int cus = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xFFFFFFFFu);
        if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
        for (int k = 0; k < reserve && k < cus; k++) mask[(size_t)k / 32] &= ~(1u << (k % 32));
        HIPCHK(ctx, hipExtStreamCreateWithCUMask(&ctx->gstream, (uint32_t)mask.size(), mask.data()));
    } else
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->gstream, hipStreamNonBlocking));
    // the band hand-over runs while the rest of the pair space is being scored: its small kernels must not queue behind
    // the thousands of workgroups of that launch, so its stream gets the highest priority
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->rest_stream, hipStreamNonBlocking, prio_lo));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_rest, hipEventDisableTiming));
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

// Where the edges of one greedy call are, and what is already known about them.  Everything the caller enqueued
// (scoring, snapshots) is on ctx->gstream; ev_edges has been recorded there after the last edge was written.
struct EdgeSource {
    EdgeSegs segs{};
    bool symmetric = true;
    bool check_overflow = false;       // segs are the HMK_EDGE_SHARDS segments of a neighbour pass: h_counts[0..16) receives
    uint64_t seg_cap = 0;              // their counts (copied on gstream before ev_edges); a count above seg_cap = overflow
    bool format_known = false;         // adjacency entry format decided without looking at the edges
    bool packed = false;
    int base = 0;
    uint64_t adj_bound = 0;            // upper bound of the adjacency entries (format_known only)
    uint64_t total_known = 0;          // exact number of edges, if known (else 0)
    uint32_t band_rows = 0;            // rows [0, band_rows) are complete in band_segs once ev_band has passed
    const uint32_t *band_gave_up = nullptr;   // device word: 1 = the wait for the band tiles timed out (band_segs are NOT complete)
    EdgeSegs band_segs{};
    bool deg_fused = false;            // the neighbour kernel placed the edges itself: SB_CURSOR holds the rows' upper | lower counters
                                       // (zeroed before the pass), SB_RANK every edge's ranks (parallel to the buffer at edges0)
    const uint64_t *edges0 = nullptr;
    bool deg_split = false;            // deg_fused without ranks: SB_DEG holds upper counts [0, n) and lower counts [n, 2n) instead of totals
    bool placed = false;               // deg_fused with ranks (else deg_fused = SB_DEG holds the rows' total degrees, counted by the pass)
    hmk_clinkage_stats *clink = nullptr;   // non-null: run the clinkage nearest-neighbour chain instead of the greedy merge
    // multi-device calls: the peers' blocks arrive while the calling thread is already inside cluster_on_device.
    //   before_band  blocks until every peer's band block is on its way to the root and makes the copy stream wait for them;
    //                non-zero: no band hand-over in this call (phase 1 then waits for the full graph)
    //   before_full  blocks until every peer's edges are on their way, makes gstream wait for them and records ev_edges;
    //                HMK_OK, ST_RETRY_OVERFLOW or an error code (the text is in the context)
    std::function<int()> before_band, before_full;
};

// The CSR scatter with its lower sections dealt by bucket (k_edges.hip, k_lower_*): for graphs whose scatter is bound by random
// writes.  Packed symmetric adjacency only; edges that were placed while they were written have their own atomic-free scatter.
// The default at every size (10^5: CSR 0.44 -> 0.31 ms, 10^6: 58 -> 23 ms); HMK_CSR_BY_BUCKET=0 scatters with atomics.
static bool csr_by_bucket(uint32_t n, bool symmetric, bool packed, bool placed) {
    if (!symmetric || !packed || placed) return false;
    if (const char *v = getenv("HMK_CSR_BY_BUCKET")) return atoi(v) != 0;
    (void)n;
    return true;
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
    const bool timing = getenv("HMK_GREEDY_TIMING") != nullptr;
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[hmk greedy] %s at %.2f ms\n", what, ms_since(t0));
    };
    bool packed = src.packed;
    int base = src.base;
    size_t esz = packed ? sizeof(NbrPacked) : sizeof(Nbr);
    const bool symmetric = src.symmetric;

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
        if (csr_by_bucket(n, symmetric, packed, src.deg_fused && src.placed)) {   // large graphs: lower sections dealt by bucket
            uint64_t records = 1;   // one per edge: at most what the segments hold
            for (uint32_t q = 0; q < src.segs.n; q++) records += src.segs.s[q].cap;
            hipError_t e = ensure_buf(ctx, SB_PART, records * 8);   // in place already when hmk_greedy_cluster scored the edges itself
            if (e == hipSuccess) e = ensure_buf(ctx, SB_PARTSCR, csr_partition_scratch_bytes());
            if (e == hipSuccess)
                e = launch_csr_scatter_partitioned(src.segs, buf<uint64_t>(ctx, SB_START), buf<uint32_t>(ctx, SB_CURSOR), buf<void>(ctx, SB_ADJ),
                                                   base, n, buf<uint64_t>(ctx, SB_PART), buf<void>(ctx, SB_PARTSCR),
                                                   src.deg_fused && src.deg_split ? buf<uint32_t>(ctx, SB_DEG) + n : nullptr, S);
            if (e == hipSuccess) e = hipEventRecord(ctx->ev_csr, S);
            return e;
        }
        hipError_t e = src.deg_fused && src.placed
                           ? launch_csr_scatter_ranked(src.segs, src.edges0, buf<uint32_t>(ctx, SB_RANK), symmetric, buf<uint64_t>(ctx, SB_START),
                                                       buf<void>(ctx, SB_ADJ), packed, base, S)
                           : launch_csr_scatter(src.segs, symmetric, buf<uint64_t>(ctx, SB_START), buf<uint32_t>(ctx, SB_CURSOR),
                                                buf<void>(ctx, SB_ADJ), packed, base, n, S);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_csr, S);
        return e;
    };
    auto enqueue_counts = [&]() -> hipError_t {
        hipError_t e_;
        if (src.deg_fused && src.placed) {
            if ((e_ = (launch_csr_scan_only(buf<uint32_t>(ctx, SB_CURSOR), symmetric ? buf<uint32_t>(ctx, SB_CURSOR) + n : nullptr,
                                             buf<uint64_t>(ctx, SB_START), n, buf<uint64_t>(ctx, SB_SCAN), buf<int>(ctx, SB_RANGE), S))) != hipSuccess) return e_;
        } else if (src.deg_fused) {
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
    if (!src.before_full && !late_buffers) HIPCHK(ctx, enqueue_full());

    // ---- band: the first rows' adjacency from the edges of the band launch, on the copy stream --------------
    uint32_t rows_here = 0;          // rows [0, rows_here) are valid in h_start / h_adj
    bool band_pending = false, band_used = false;
    uint32_t R1 = src.band_rows;
    if (R1 > 0 && src.before_band && src.before_band() != HMK_OK) R1 = 0;   // (the peers' band blocks did not make it: no band)
    if (R1 > 0 && src.format_known) {
        HIPCHK(ctx, ensure_buf(ctx, SB_BDEG, (size_t)R1 * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_BCURSOR, (size_t)R1 * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_BSTART, ((size_t)R1 + 1) * 8));
        HIPCHK(ctx, ensure_buf(ctx, SB_BSCAN, scan_scratch_bytes(R1)));
        HIPCHK(ctx, ensure_buf(ctx, SB_BRANGE, 64));
        HIPCHK(ctx, hipStreamWaitEvent(C, ctx->ev_band, 0));
        HIPCHK(ctx, hipMemsetAsync(buf<void>(ctx, SB_BDEG), 0, (size_t)R1 * 4, C));
        HIPCHK(ctx, hipMemsetAsync(buf<void>(ctx, SB_BCURSOR), 0, (size_t)R1 * 8, C));
        HIPCHK(ctx, launch_csr_degree_scan(src.band_segs, n, R1, symmetric, buf<uint32_t>(ctx, SB_BDEG), buf<uint64_t>(ctx, SB_BSTART),
                                           buf<uint64_t>(ctx, SB_BSCAN), buf<int>(ctx, SB_BRANGE), C));
        HIPCHK(ctx, hipMemcpyAsync(h_start, buf<uint64_t>(ctx, SB_BSTART), ((size_t)R1 + 1) * 8, hipMemcpyDeviceToHost, C));

        // (the band's own segments only: beside a pass that runs at the same time the other cursors are in motion)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts + HC_BAND, src.band_segs.s[0].count,
                                   std::min<uint32_t>(src.band_segs.n, HMK_EDGE_SHARDS) * sizeof(unsigned long long), hipMemcpyDeviceToHost, C));
        ((uint32_t *)(ctx->h_counts + HC_MISC))[7] = 0;
        if (src.band_gave_up)
            HIPCHK(ctx, hipMemcpyAsync((uint32_t *)(ctx->h_counts + HC_MISC) + 7, src.band_gave_up, 4, hipMemcpyDeviceToHost, C));
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
        if (!full_enqueued) {   // multi-device: the peers' edges first; (or: the late buffers are ready only now)
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
                packed = h_start[n] == 0 || ((long long)h_range[1] - h_range[0] <= 255 && getenv("HMK_ADJ_8BYTE") == nullptr);
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
    hooks.need_rows = [&](uint32_t k) -> uint32_t {
        if (ctx->wedged) return 0;   // (the second loop gave the device up: the merge stops here instead of waiting for it again)
        if (k < rows_here) return rows_here;
        const auto tw = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        if (band_pending) {
            band_pending = false;
            e = hipEventSynchronize(ctx->ev_bandcsr);
            bool ok = e == hipSuccess && ((const uint32_t *)(ctx->h_counts + HC_MISC))[7] == 0;   // ([7]: the wait for the band tiles gave up)
            for (uint32_t q = 0; q < std::min<uint32_t>(src.band_segs.n, HMK_EDGE_SHARDS) && ok; q++) ok = ctx->h_counts[HC_BAND + q] <= src.seg_cap;
            if (ok) {
                const uint64_t entries = h_start[R1];
                e = ensure_buf(ctx, SB_BADJ, std::max<uint64_t>(entries, 1) * esz);
                if (e == hipSuccess) e = ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, std::max<uint64_t>(entries, 1) * esz, 0);
                if (e == hipSuccess)
                    e = launch_csr_scatter(src.band_segs, symmetric, buf<uint64_t>(ctx, SB_BSTART), buf<uint32_t>(ctx, SB_BCURSOR),
                                           buf<void>(ctx, SB_BADJ), packed, base, R1, C);
                if (e == hipSuccess && entries)
                    e = hipMemcpyAsync(ctx->h_adj, buf<void>(ctx, SB_BADJ), entries * esz, hipMemcpyDeviceToHost, C);
                // the band rows' upper-section sizes travel with them: upper[] must never hold a previous call's values for rows
                // the merge may read (today every reader refetches from the full CSR first; this keeps it true by construction)
                if (e == hipSuccess && symmetric)
                    e = hipMemcpyAsync(h_up, buf<uint32_t>(ctx, SB_BCURSOR), (size_t)R1 * 4, hipMemcpyDeviceToHost, C);
                if (e == hipSuccess) e = hipStreamSynchronize(C);
                if (e == hipSuccess) {
                    rows_here = R1;
                    band_used = true;
                    lap("band rows on the host");
                }
            }
            if (e != hipSuccess) { hook_fail(HMK_ERR_DEVICE, std::string("band hand-over: ") + hipGetErrorString(e)); return 0; }
            if (k < rows_here) { t_rows += ms_since(tw); return rows_here; }
        }
        // more rows from the full CSR (which must be complete by now)
        if (!wait_full()) return 0;
        if (band_used) { rows_here = 0; band_used = false; }   // the band rows come again, in the full CSR's layout
        uint32_t r_end = n;
        if (k + 1 < n) r_end = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>({(uint64_t)k + 1, 2ull * rows_here, 8192ull}));
        const uint64_t *d_start = buf<uint64_t>(ctx, SB_START);
        e = hipMemcpyAsync(h_start + rows_here, d_start + rows_here, ((size_t)(r_end - rows_here) + 1) * 8, hipMemcpyDeviceToHost, C);
        if (e == hipSuccess && symmetric)
            e = hipMemcpyAsync(h_up + rows_here, buf<uint32_t>(ctx, SB_CURSOR) + rows_here, (size_t)(r_end - rows_here) * 4, hipMemcpyDeviceToHost, C);
        if (e == hipSuccess) e = hipStreamSynchronize(C);
        if (e == hipSuccess) {
            const uint64_t a0 = h_start[rows_here], a1 = h_start[r_end];
            e = ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, std::max<uint64_t>(a1, 1) * esz, a0 * esz);
            if (e == hipSuccess && a1 > a0)
                e = hipMemcpyAsync((char *)ctx->h_adj + a0 * esz, (const char *)buf<void>(ctx, SB_ADJ) + a0 * esz, (a1 - a0) * esz,
                                   hipMemcpyDeviceToHost, C);
            if (e == hipSuccess) e = hipStreamSynchronize(C);
        }
        if (e != hipSuccess) { hook_fail(e == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE, std::string("adjacency copy: ") + hipGetErrorString(e)); return 0; }
        rows_here = r_end;
        t_rows += ms_since(tw);
        return rows_here;
    };

    // ---- second loop on the device-resident CSR ---------------------------------------------------------------
    // (1) pre-check (k_greedy_precheck): per leftover the clusters that are feasible after phase 1 -> cand CSR on the device.
    // Then either (2a) small / medium inputs: join-propagation lists (k_greedy_prop), the sequential loop runs on the
    // host over those lists; or (2b) large inputs: the loop itself runs on the device level by level (k_greedy_level).
    // pre_mode: 0 nothing yet, 1 = two passes done (cand_start[] are prefix sums: what the host-side consumers read),
    // 2 = one pass done (every leftover's block lies where the global counter put it: the device loop takes either)
    int pre_mode = 0;
    uint32_t pre_total_c = 0;
    auto device_precheck = [&](const int32_t *cluster_of, const std::vector<int32_t> &usize, const std::vector<uint32_t> &leftover,
                               bool single_pass) -> bool {
        if (pre_mode == 1 || (pre_mode == 2 && single_pass)) return true;
        if (getenv("HMK_HOST_PRECHECK")) return false;
        if (getenv("HMK_PRECHECK_TWO_PASSES")) single_pass = false;
        if (!wait_full()) return false;
        const auto tp = std::chrono::steady_clock::now();
        const uint32_t nl = (uint32_t)leftover.size();
        hipError_t r = ensure_buf(ctx, SB_COF, (size_t)n * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_BITMAP, ((size_t)n + 31) / 32 * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_USIZE, std::max<size_t>(usize.size(), 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_LEFT, std::max<size_t>(nl, 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_CNT, std::max<size_t>(nl, 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_CSTART, ((size_t)nl + 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_OVER, 64);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SCAN2, scan_scratch_bytes(std::max<uint32_t>(nl, n)));
        if (r != hipSuccess) return false;
        int32_t *d_cof = buf<int32_t>(ctx, SB_COF), *d_usize = buf<int32_t>(ctx, SB_USIZE);
        uint32_t *d_left = buf<uint32_t>(ctx, SB_LEFT), *d_cnt = buf<uint32_t>(ctx, SB_CNT), *d_cstart = buf<uint32_t>(ctx, SB_CSTART);
        uint32_t *d_over = buf<uint32_t>(ctx, SB_OVER);                      // [0] table overflows, [2..3] the single pass's entry counter
        unsigned long long *d_total = (unsigned long long *)(d_over + 2);
        uint64_t *d_scan = buf<uint64_t>(ctx, SB_SCAN2);
        const uint64_t *d_start = buf<uint64_t>(ctx, SB_START);
        const void *d_adj = buf<void>(ctx, SB_ADJ);
        uint32_t *h_misc = (uint32_t *)(ctx->h_counts + HC_MISC);
        if (pre_mode == 0) {
            // through a pinned block: an "async" upload from pageable memory is staged by the runtime chunk by chunk and the
            // stream waits for it (0.3 ms for these 0.8 MB at 10^5, seen as the pre-check kernel starting late)
            const size_t b_cof = (size_t)n * 4, b_us = usize.size() * 4, b_left = (size_t)nl * 4;
            r = ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, HMK_PRE_REGIONS * sizeof(unsigned long long) + b_cof + b_us + b_left + 64, 0);
            if (r != hipSuccess) return false;
            char *hs = (char *)ctx->h_stage + HMK_PRE_REGIONS * sizeof(unsigned long long);   // (the block starts with the single pass's region counters)
            std::memcpy(hs, cluster_of, b_cof);
            std::memcpy(hs + b_cof, usize.data(), b_us);
            std::memcpy(hs + b_cof + b_us, leftover.data(), b_left);
            r = hipMemcpyAsync(d_cof, hs, b_cof, hipMemcpyHostToDevice, S);
            if (r == hipSuccess) r = launch_cluster_bitmap(d_cof, n, buf<uint32_t>(ctx, SB_BITMAP), S);
            if (r == hipSuccess && b_us) r = hipMemcpyAsync(d_usize, hs + b_cof, b_us, hipMemcpyHostToDevice, S);
            if (r == hipSuccess && b_left) r = hipMemcpyAsync(d_left, hs + b_cof + b_us, b_left, hipMemcpyHostToDevice, S);
        }
        if (r == hipSuccess) r = hipMemsetAsync(d_over, 0, 16, S);
        if (r == hipSuccess && single_pass) {
            // One pass: every wave takes its block of entries from the counter of its workgroup's region of the buffer.  The
            // buffer is sized from what previous calls needed (or 24 entries per leftover); a call that overruns a region falls
            // back to the two passes below.
            const size_t want = std::max<size_t>({ctx->sb[SB_CAND].cap / sizeof(GreedyCand), (size_t)nl * 24, (size_t)HMK_PRE_REGIONS * 64});
            const unsigned long long region_cap = std::min<unsigned long long>(want, 0xFFFFFFFFull) / HMK_PRE_REGIONS;
            r = ensure_buf(ctx, SB_CAND, (size_t)region_cap * HMK_PRE_REGIONS * sizeof(GreedyCand));
            if (r == hipSuccess) r = ensure_buf(ctx, SB_PRECNT, HMK_PRE_REGIONS * sizeof(unsigned long long));
            unsigned long long *d_regions = buf<unsigned long long>(ctx, SB_PRECNT);
            if (r == hipSuccess) r = hipMemsetAsync(d_regions, 0, HMK_PRE_REGIONS * sizeof(unsigned long long), S);
            // rows with few neighbours inside clusters (the estimate: average degree x the clustered share of the sequences) go
            // through small tables first
            size_t in_clusters = 0;
            for (int32_t u : usize) in_clusters += (size_t)u;
            const double est = (double)h_start[n] / std::max<uint32_t>(n, 1) * (double)in_clusters / std::max<uint32_t>(n, 1);
            uint32_t *d_retry = nullptr;
            const int first_slots = est <= 24.0 ? 128 : 512;   // ~5 x the expected number of distinct clusters in a row
            // (10^6 default-threshold 12-mers give an estimate of 130; small tables first for them too -- 512 slots, five workgroups
            // per CU instead of two -- was measured and loses: 22.0 against 18.7 ms, 38.9 against 25.8 ms in the reference's
            // default order, where many rows see far more clusters than the average and are scanned twice)
            double two_stage_limit = 100.0;
            if (const char *v = getenv("HMK_PRECHECK_TWO_STAGE_LIMIT")) two_stage_limit = atof(v);
            if (r == hipSuccess && est <= two_stage_limit && getenv("HMK_PRECHECK_ONE_STAGE") == nullptr) {
                r = ensure_buf(ctx, SB_RETRY, std::max<size_t>(nl, 1) * 4);
                d_retry = buf<uint32_t>(ctx, SB_RETRY);
            }
            if (r == hipSuccess) r = launch_greedy_precheck(2, packed, d_start, d_adj, d_cof, buf<uint32_t>(ctx, SB_BITMAP), d_usize, d_left, nl,
                                                            d_cnt, d_cstart, buf<GreedyCand>(ctx, SB_CAND), d_over, d_regions, region_cap,
                                                            d_retry, d_over + 1, first_slots, S);
            unsigned long long *h_regions = (unsigned long long *)ctx->h_stage;   // (sized with the uploads above: pre_mode is 0 here)
            if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[0], d_over, 4, hipMemcpyDeviceToHost, S);
            if (r == hipSuccess) r = hipMemcpyAsync(h_regions, d_regions, HMK_PRE_REGIONS * sizeof(unsigned long long), hipMemcpyDeviceToHost, S);
            if (r == hipSuccess) r = hipStreamSynchronize(S);
            if (r != hipSuccess || h_misc[0] != 0) return false;   // a row overflowed its hash table: host pre-check
            unsigned long long total = 0;
            bool fits = true;
            for (uint32_t g = 0; g < HMK_PRE_REGIONS; g++) { total += h_regions[g]; fits = fits && h_regions[g] <= region_cap; }
            if (fits) {
                pre_total_c = (uint32_t)total;
                pre_mode = 2;
                ph.cand_entries = pre_total_c;
                ph.precheck_ms = ms_since(tp);
                return true;
            }
            if (total > 0xFFFFFFFFull) return false;
            r = hipMemsetAsync(d_over, 0, 16, S);   // more entries than a region holds: count, size, fill
        }
        if (r == hipSuccess) r = launch_greedy_precheck(0, packed, d_start, d_adj, d_cof, buf<uint32_t>(ctx, SB_BITMAP), d_usize, d_left, nl,
                                                        d_cnt, nullptr, nullptr, d_over, d_total, 0, nullptr, nullptr, 0, S);
        if (r == hipSuccess) r = launch_scan_u32(d_cnt, d_cstart, nl, d_scan, S);
        if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[0], d_over, 4, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[1], d_cstart + nl, 4, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess) r = hipStreamSynchronize(S);
        if (r != hipSuccess || h_misc[0] != 0) return false;   // a row overflowed its hash table: host pre-check
        pre_total_c = h_misc[1];
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
    const char *loop_mode = getenv("HMK_SECOND_LOOP");   // "device" / "lists" / "host": force one implementation (tests)
    const bool force_device = loop_mode && std::strcmp(loop_mode, "device") == 0;
    const bool forbid_device = loop_mode && !force_device;
    const bool forbid_lists = loop_mode && std::strcmp(loop_mode, "lists") != 0;

    hooks.device_loop = [&](const int32_t *cluster_of, const std::vector<int32_t> &usize, const std::vector<int64_t> &csize,
                            const std::vector<int32_t> &cids, const std::vector<uint32_t> &leftover,
                            std::vector<int32_t> &join_slot) -> bool {
        if (forbid_device || !symmetric) return false;
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
        if (r == hipSuccess) r = ensure_buf(ctx, SB_JSLOT, std::max<size_t>(nl, 1) * 4);
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
        if (r == hipSuccess) r = hipMemsetAsync(d_taken, 0, (size_t)ncl * 4, S);
        if (r == hipSuccess) r = hipMemsetAsync(buf<void>(ctx, SB_STATUS), 0, nl, S);
        if (r == hipSuccess) r = hipMemsetAsync(buf<void>(ctx, SB_JSLOT), 0xFF, (size_t)nl * 4, S);
        if (r == hipSuccess) r = hipMemsetAsync(buf<void>(ctx, SB_LCOUNT), 0, 64, S);
        if (r == hipSuccess) r = hipMemcpyAsync(buf<void>(ctx, SB_CSIZE), csize.data(), (size_t)ncl * 8, hipMemcpyHostToDevice, S);
        if (r == hipSuccess) r = hipMemcpyAsync(buf<void>(ctx, SB_CID), cids.data(), (size_t)ncl * 4, hipMemcpyHostToDevice, S);
        if (r == hipSuccess) r = launch_loop_init(ncl, buf<long long>(ctx, SB_CSIZE), buf<int32_t>(ctx, SB_CID), buf<void>(ctx, SB_JOINED),
                                                  buf<uint32_t>(ctx, SB_SUBSTART), d_clcursor, nl, buf<uint32_t>(ctx, SB_ACTIVE),
                                                  buf<uint32_t>(ctx, SB_DIRTY), buf<uint32_t>(ctx, SB_LCOUNT), S);
        if (r == hipSuccess && ctx->has_sizes)
            r = hipMemcpyAsync(buf<void>(ctx, SB_SEQSZ), ctx->sizes.data(), (size_t)n * 4, hipMemcpyHostToDevice, S);
        uint32_t *h_misc = (uint32_t *)(ctx->h_counts + HC_MISC);
        uint32_t rounds = 0;
        bool done = false;
        // a second first/accept pass per round saves a third of the rounds; it pays once a round's apply and eval are big enough
        int accept_passes = ncl >= 8192 ? 2 : 1;
        if (const char *v = getenv("HMK_LOOP_PASSES")) accept_passes = std::min(8, std::max(1, atoi(v)));
        // Every round accepts at least the earliest open leftover that has a feasible cluster, so nl + 1 rounds always suffice
        // and a round without a join is the end.  The host keeps enqueuing rounds while it watches the progress word that
        // k_loop_apply stores into pinned host memory (round << 32 | joins of that round), at most LOOKAHEAD rounds ahead of
        // the device; rounds enqueued after the end find nothing to do.  Without the word: batches of rounds and a sync each.
        auto one_round = [&]() {
            r = launch_loop_round(packed, buf<uint64_t>(ctx, SB_START), buf<uint32_t>(ctx, SB_CURSOR), buf<void>(ctx, SB_ADJ),
                                  buf<uint32_t>(ctx, SB_LEFT), nl, buf<uint32_t>(ctx, SB_CSTART), buf<uint32_t>(ctx, SB_CNT),
                                  buf<GreedyCand>(ctx, SB_CAND), buf<uint8_t>(ctx, SB_STATUS), buf<uint32_t>(ctx, SB_CHOICE),
                                  buf<uint32_t>(ctx, SB_ACTIVE), buf<uint32_t>(ctx, SB_DIRTY), rounds, d_first, d_taken, d_clcursor,
                                  ncl, accept_passes, buf<uint32_t>(ctx, SB_ACCEPTED), buf<int32_t>(ctx, SB_JSLOT),
                                  buf<uint32_t>(ctx, SB_SUBSTART), buf<uint64_t>(ctx, SB_SUBS), buf<void>(ctx, SB_JOINED),
                                  ctx->has_sizes ? buf<int32_t>(ctx, SB_SEQSZ) : nullptr, buf<uint32_t>(ctx, SB_LCOUNT), ctx->h_loop, S);
            rounds++;
        };
        if (nl == 0 || ncl == 0) {
            done = true;
        } else if (ctx->h_loop && getenv("HMK_LOOP_BATCHES") == nullptr) {
            uint32_t LOOKAHEAD = 4;   // a round is 4-6 small dependent kernels: a few rounds in the queue keep the device busy
            if (const char *v = getenv("HMK_LOOP_LOOKAHEAD")) LOOKAHEAD = (uint32_t)std::max(1, atoi(v));
            volatile unsigned long long *word = ctx->h_loop;
            *word = 0;
            const bool loop_trace = getenv("HMK_LOOP_TRACE") != nullptr;   // (with HMK_LOOP_LOOKAHEAD=1 every round is seen)
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
                        if (loop_trace) std::fprintf(stderr, "[hmk greedy] loop round %u: %u joins, %.3f ms since the loop began\n", seen, (uint32_t)w, ms_since(tl));
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
            if (r == hipSuccess) r = hipStreamSynchronize(S);       // drain the rounds enqueued past the end
            if (loop_trace && r == hipSuccess) {   // (a build with -DHMK_APPLY_STATS=1 fills these)
                uint32_t hc[16] = {0};
                if (hipMemcpy(hc, buf<uint32_t>(ctx, SB_LCOUNT), 64, hipMemcpyDeviceToHost) == hipSuccess && (hc[8] | hc[10]))
                    std::fprintf(stderr, "[hmk greedy] apply walked %u subscriber entries (longest list %u) and %u row entries (longest row %u); "
                                         "joins took %.2f ms in all (longest %.1f us), of which table build %.2f ms, subscribers %.2f ms\n",
                                 hc[8], hc[9], hc[10], hc[11], hc[12] * 1e-5, hc[13] * 1e-2, hc[14] * 1e-5, hc[15] * 1e-5);
            }
        } else {
            for (uint32_t batch = 8; r == hipSuccess && !done && rounds <= nl + 8; batch = std::min<uint32_t>(batch * 2, 64)) {
                for (uint32_t b = 0; b < batch && r == hipSuccess; b++) one_round();
                if (r == hipSuccess) r = hipMemcpyAsync(&h_misc[3], buf<uint32_t>(ctx, SB_LCOUNT) + 3, 4, hipMemcpyDeviceToHost, S);
                if (r == hipSuccess) r = hipStreamSynchronize(S);
                done = r == hipSuccess && h_misc[3] == 0;
            }
        }
        if (r != hipSuccess || !done) return false;
        join_slot.resize(nl);
        if (nl) {   // through the pinned block (a copy into pageable memory is staged chunk by chunk)
            r = ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, (size_t)nl * 4 + 64, 0);
            if (r == hipSuccess) r = hipMemcpyAsync(ctx->h_stage, buf<void>(ctx, SB_JSLOT), (size_t)nl * 4, hipMemcpyDeviceToHost, S);
            if (r == hipSuccess) r = hipStreamSynchronize(S);
            if (r == hipSuccess) std::memcpy(join_slot.data(), ctx->h_stage, (size_t)nl * 4);
        }
        if (r != hipSuccess) return false;
        ph.device_loop_ms = ms_since(tl);
        ph.loop_rounds = rounds;
        lap("device second loop (rounds)");
        return true;
    };

    hooks.precheck = [&](const int32_t *cluster_of, const std::vector<int32_t> &usize, const std::vector<uint32_t> &leftover,
                         bool want_prop, std::vector<uint32_t> &cand_start, std::vector<GreedyCand> &cand,
                         std::vector<uint32_t> &prop_start, std::vector<GreedyProp> &prop, bool *have_prop) -> bool {
        *have_prop = false;
        if (!device_precheck(cluster_of, usize, leftover, false)) return false;
        const uint32_t nl = (uint32_t)leftover.size();
        const uint32_t total_c = pre_total_c;
        if (!fetch_cand(nl, cand_start, cand)) return false;
        lap("device pre-check");
        if (!want_prop || !symmetric || forbid_lists || getenv("HMK_HOST_PROPAGATION")) return true;
        // ---- join-propagation lists -------------------------------------------------------------------------
        const auto tq = std::chrono::steady_clock::now();
        prop_start.assign((size_t)total_c + 1, 0);
        prop.clear();
        if (total_c == 0) { *have_prop = true; return true; }
        hipError_t r = ensure_buf(ctx, SB_LIDX, (size_t)n * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_PCNT, (size_t)total_c * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_PSTART, ((size_t)total_c + 1) * 4);
        if (r == hipSuccess) r = ensure_buf(ctx, SB_SCAN2, scan_scratch_bytes(std::max<uint32_t>({nl, n, total_c})));
        if (r != hipSuccess) return true;   // candidates are fine; the merge falls back to stamping rows
        uint64_t *d_scan = buf<uint64_t>(ctx, SB_SCAN2);
        int32_t *d_lidx = buf<int32_t>(ctx, SB_LIDX);
        uint32_t *d_pcnt = buf<uint32_t>(ctx, SB_PCNT), *d_pstart = buf<uint32_t>(ctx, SB_PSTART);
        const uint32_t *d_up = buf<uint32_t>(ctx, SB_CURSOR);
        const uint64_t *d_start = buf<uint64_t>(ctx, SB_START);
        const void *d_adj = buf<void>(ctx, SB_ADJ);
        const uint32_t *d_left = buf<uint32_t>(ctx, SB_LEFT), *d_cstart = buf<uint32_t>(ctx, SB_CSTART);
        r = launch_fill_lidx(d_left, nl, d_lidx, n, S);
        if (r == hipSuccess) r = hipMemsetAsync(d_pcnt, 0, (size_t)total_c * 4, S);
        if (r == hipSuccess) r = launch_greedy_prop(false, packed, d_start, d_up, d_adj, d_lidx, d_left, nl, d_cstart,
                                                    buf<GreedyCand>(ctx, SB_CAND), d_pcnt, nullptr, nullptr, S);
        if (r == hipSuccess) r = launch_scan_u32(d_pcnt, d_pstart, total_c, d_scan, S);
        unsigned long long *h_total = ctx->h_counts + HC_TOTAL;   // the scan's 64-bit grand total (its uint32 start[] may wrap)
        if (r == hipSuccess) r = hipMemcpyAsync(h_total, d_scan + scan_total_index(total_c), 8, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess) r = hipStreamSynchronize(S);
        if (r != hipSuccess) return true;
        if (*h_total > (1ull << 28)) return true;   // very dense families: let the host stamp rows instead (2 GB of lists)
        const uint32_t total_p = (uint32_t)*h_total;
        prop.resize(total_p);
        r = hipMemcpyAsync(prop_start.data(), d_pstart, ((size_t)total_c + 1) * 4, hipMemcpyDeviceToHost, S);
        if (r == hipSuccess && total_p) {
            r = ensure_buf(ctx, SB_PROP, (size_t)total_p * sizeof(GreedyProp));
            if (r == hipSuccess) r = hipMemsetAsync(d_pcnt, 0, (size_t)total_c * 4, S);
            if (r == hipSuccess) r = launch_greedy_prop(true, packed, d_start, d_up, d_adj, d_lidx, d_left, nl, d_cstart,
                                                        buf<GreedyCand>(ctx, SB_CAND), d_pcnt, d_pstart, buf<GreedyProp>(ctx, SB_PROP), S);
            if (r == hipSuccess) r = hipMemcpyAsync(prop.data(), buf<GreedyProp>(ctx, SB_PROP), (size_t)total_p * sizeof(GreedyProp),
                                                    hipMemcpyDeviceToHost, S);
        }
        if (r == hipSuccess) r = hipStreamSynchronize(S);
        if (r != hipSuccess) { prop.clear(); return true; }
        *have_prop = true;
        ph.prop_ms = ms_since(tq);
        ph.prop_entries = total_p;
        lap("device join-propagation lists");
        return true;
    };

    hooks.adj_base = [&]() -> const void * { return ctx->h_adj; };
    GreedyTimes times{};
    hooks.times = &times;
    std::string err;
    const int32_t *szs = ctx->has_sizes ? ctx->sizes.data() : nullptr;
    if (src.clink) {
        // clinkage mode: the chain needs every row; fetch the whole adjacency, then run it on the host
        int cst = HMK_OK;
        if (!src.format_known && !wait_full()) cst = -1;
        if (cst == HMK_OK && n && hooks.need_rows(n - 1) < n) cst = -1;
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
    st = packed ? greedy_from_csr_packed(n, szs, h_start, (const NbrPacked *)ctx->h_adj, symmetric ? h_up : nullptr, &hooks, symmetric,
                                         max_clusters, cluster_id, result_order, member_rank, stats, &err)
                : greedy_from_csr(n, szs, h_start, (const Nbr *)ctx->h_adj, symmetric ? h_up : nullptr, &hooks, symmetric, max_clusters,
                                  cluster_id, result_order, member_rank, stats, &err);
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

// The grow-only device and pinned buffers the tail of a clustering call on n sequences asks for (the edge buffer must have
// its size already): hmk_greedy_cluster before it enqueues the pass, hmk_reserve from a host that knows n early.
int reserve_tail_buffers(hmk_ctx *ctx, uint32_t n, bool packed, uint32_t r1, bool full = false, bool late_on_a_thread = false) {
    const size_t esz0 = packed ? sizeof(NbrPacked) : sizeof(Nbr);
    const size_t adj_bytes = std::max<uint64_t>((ctx->symmetric ? 2 : 1) * ctx->d_edges_cap, 1) * esz0;
    size_t part_bytes = 0;
    {
        bool place0 = false;
        if (const char *v = getenv("HMK_PLACE_EDGES")) place0 = getenv("HMK_NO_FUSED_DEGREE") == nullptr && atoi(v) != 0;
        if (csr_by_bucket(n, ctx->symmetric, packed, place0)) part_bytes = (ctx->d_edges_cap + 1) * 8;
    }
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
            HIPCHK(ctx, ensure_pinned(&ctx->h_adj, &ctx->h_adj_cap, std::max<uint64_t>(entries, 1) * esz0, 0));
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
        HIPCHK(ctx, ensure_buf(ctx, SB_JSLOT, nl * 4));
        HIPCHK(ctx, ensure_buf(ctx, SB_LCOUNT, 64));
        HIPCHK(ctx, ensure_buf(ctx, SB_SEQSZ, (size_t)n * 4));
        HIPCHK(ctx, ensure_pinned(&ctx->h_stage, &ctx->h_stage_cap, HMK_PRE_REGIONS * sizeof(unsigned long long) + (size_t)n * 12 + ncl * 4 + 64, 0));
    }
    if (late_on_a_thread) (void)join_late_buffers(ctx);   // (an earlier hmk_reserve's thread may still be writing the two sizes read next)
    if (late_on_a_thread && (ctx->sb[SB_ADJ].cap < adj_bytes || ctx->sb[SB_PART].cap < part_bytes)) {
        const int device = ctx->device;
        ctx->late_buffers = std::async(std::launch::async, [ctx, device, adj_bytes, part_bytes]() -> hipError_t {
            if (const char *v = getenv("HMK_LATE_BUFFERS_DELAY_MS"))   // tests: a host on which device memory is slow to get
                std::this_thread::sleep_for(std::chrono::milliseconds(std::max(0, atoi(v))));
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
    if (const char *v = getenv("HMK_EDGE_GUESS")) guess = std::strtoull(v, nullptr, 10);   // tests: force the overflow / retry path
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

int greedy_cluster_multi(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int max_clusters, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats, hmk_clinkage_stats *clink = nullptr);

}  // namespace

extern "C" {

int hmk_greedy_cluster(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int max_clusters,
                       int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    const bool call_timing = getenv("HMK_GREEDY_TIMING") != nullptr;
    auto call_lap = [&](const char *what) {
        if (call_timing) fprintf(stderr, "[hmk greedy] %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    const uint32_t n = ctx->n;
    hipStream_t S = ctx->gstream;
    // Band: phase 1 of the merge (LimitedGreedySequenceClusterer.java:77-120) reads the adjacency rows in order and
    // stops once maxClusters clusters exist, normally a little after row maxClusters.  The tiles that complete the first
    // band_rows rows are launched first, their rows are handed to the host while the rest of the pair space is being scored.
    int64_t band_rows = 0;
    if (max_clusters > 0 && n >= 16384 && getenv("HMK_NO_BAND") == nullptr)
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
    src.packed = top - threshold <= 255 && getenv("HMK_ADJ_8BYTE") == nullptr;
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
        // HMK_BAND_CONCURRENT=1 (measured, rejected, kept as a switch): the band tiles and the others scored AT THE SAME TIME, two
        // streams, the others' at the lowest priority, each launch into output segments of its own (the band's share of the
        // HMK_EDGE_SHARDS segments = its share of the pairs; the snapshot of the band's cursors then names complete segments only
        // -- in shared segments the other launch's waves would have reserved places they have not written yet).  The idea: one
        // launch after the other leaves the GPU half empty twice, at the band's tail and at the rest's start (10^5: 3.4 ms of
        // scoring against 3.0 ms in one launch).  What happened: the scoring took 3.33 ms, and the band's launch, sharing every
        // CU with the other one whatever the priorities say, finished with it -- band rows on the host at 3.4 instead of 1.7 ms,
        // the call 6.3 instead of 5.0 ms.  Results identical (the GPU test suite passes either way).
        // HMK_BAND_ONE_LAUNCH=1 (measured, rejected, kept as a switch): ONE launch, band tiles first in dispatch order, segments of
        // their own for them (the same share rule), and a counter every band tile's workgroup bumps when its edges are out
        // (band_tile_done): a one-wave kernel on the hand-over stream waits for the counter (launch_wait_counter), and what is
        // enqueued behind it -- the snapshot of the band's cursors, the band CSR, the copies -- starts when the band is complete.
        // What happened at 10^5: the waiting kernel (and everything behind it) got no workgroup slot before the pass was over --
        // band rows on the host at 3.8 ms instead of 1.7, the call 6.4 instead of 4.9 ms; with CUs kept free for it
        // (HMK_CU_RESERVE=8: the clustering stream under a CU mask) the rows came at 1.4 ms, but the masked pass took 3.55 ms and
        // the call 5.0 ms.  A kernel of another stream does not get in while a launch has workgroups waiting; the band launch's
        // END is what lets the hand-over in.  So: two launches, one after the other, as in rounds 2-3.
        uint32_t band_shards = 0;
        const bool band_one_launch = band_rows > 0 && getenv("HMK_BAND_ONE_LAUNCH") != nullptr && getenv("HMK_BAND_CONCURRENT") == nullptr &&
                                     ctx->plan.stats.pairs_scored > 0;
        if (band_rows > 0 && (band_one_launch || getenv("HMK_BAND_CONCURRENT") != nullptr) && ctx->plan.stats.pairs_scored > 0) {
            const double share = (double)ctx->plan.band_pairs / (double)ctx->plan.stats.pairs_scored;
            band_shards = (uint32_t)std::min<double>(HMK_EDGE_SHARDS / 2, std::max<double>(2.0, std::ceil(share * HMK_EDGE_SHARDS)));
        }
        src.band_segs = band_shards ? shard_segments(ctx->d_edges, seg, buf<unsigned long long>(ctx, SB_BCOUNTS), 0, band_shards)
                                    : shard_segments(ctx->d_edges, seg, buf<unsigned long long>(ctx, SB_BCOUNTS));
        // the neighbour kernel places every edge in the CSR as it writes it: per row an upper and a lower counter (they end
        // up as the sizes of the row's two sections; the upper ones ARE up[]) and, beside the edge, its two ranks
        const bool fuse = getenv("HMK_NO_FUSED_DEGREE") == nullptr;
        // (placing beat the atomic scatter up to 5 x 10^5 sequences -- 10^5 CSR 0.89 -> 0.42 ms, 3 x 10^5 6.0 -> 3.7 ms -- and lost to
        // the bucketed one at every size: 10^5 0.31 ms, 3 x 10^5 2.1 ms, 5 x 10^5 5.7 against 12.1 ms, and the pass itself is 1-4 %
        // faster without the returning atomics.  It stays as HMK_PLACE_EDGES=1.)
        bool place = false;
        if (const char *v = getenv("HMK_PLACE_EDGES")) place = fuse && atoi(v) != 0;
        uint32_t *d_deg = nullptr, *d_deg_lo = nullptr, *d_rank = nullptr;
        if (place) {
            HIPCHK(ctx, ensure_buf(ctx, SB_CURSOR, (size_t)n * 8));
            HIPCHK(ctx, ensure_buf(ctx, SB_RANK, ctx->d_edges_cap * 8));
            d_deg = buf<uint32_t>(ctx, SB_CURSOR);
            d_deg_lo = ctx->symmetric ? d_deg + n : nullptr;
            d_rank = buf<uint32_t>(ctx, SB_RANK);
            HIPCHK(ctx, hipMemsetAsync(d_deg, 0, (size_t)n * 8, S));
        } else if (fuse) {
            // symmetric: the smaller end counts into up[], the larger into lo[] -- the same number of atomics as one total per
            // row, and the lower counts give the bucket sizes of the CSR's dealing pass without a pass over the edges
            // (k_lower_count, 2 ms at 10^6).  HMK_NO_SPLIT_DEGREE=1: one counter per row.
            const bool split = ctx->symmetric && getenv("HMK_NO_SPLIT_DEGREE") == nullptr;
            HIPCHK(ctx, ensure_buf(ctx, SB_DEG, (size_t)n * (split ? 8 : 4)));
            d_deg = buf<uint32_t>(ctx, SB_DEG);
            d_deg_lo = split ? d_deg + n : nullptr;
            HIPCHK(ctx, hipMemsetAsync(d_deg, 0, (size_t)n * (split ? 8 : 4), S));
            src.deg_split = split;
        }
        src.deg_fused = fuse;
        src.placed = place;
        src.edges0 = ctx->d_edges;
        if (band_one_launch) {
            uint32_t n_band_tiles = 0;
            for (const Group &g : ctx->plan.groups) n_band_tiles += g.band;
            HIPCHK(ctx, ensure_buf(ctx, SB_BANDCTR, 64));
            HIPCHK(ctx, hipMemsetAsync(buf<void>(ctx, SB_BANDCTR), 0, 64, S));
            HIPCHK(ctx, hipEventRecord(ctx->ev_t0, S));
            st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, 1, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S,
                                      LAUNCH_ALL, band_req, d_deg, d_deg_lo, d_rank, band_shards, HMK_EDGE_SHARDS - band_shards, band_shards,
                                      buf<uint32_t>(ctx, SB_BANDCTR));
            call_lap("all tiles enqueued");
            if (st) { (void)hipStreamSynchronize(S); return st; }
            // the hand-over stream: behind the counter's memset, wait for the band tiles, then the snapshot of the band's cursors
            hipStream_t C = ctx->copy_stream;
            HIPCHK(ctx, hipStreamWaitEvent(C, ctx->ev_t0, 0));
            HIPCHK(ctx, launch_wait_counter(buf<uint32_t>(ctx, SB_BANDCTR), n_band_tiles, buf<uint32_t>(ctx, SB_BANDCTR) + 1, C));
            HIPCHK(ctx, hipMemcpyAsync(buf<void>(ctx, SB_BCOUNTS), ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long),
                                       hipMemcpyDeviceToDevice, C));
            HIPCHK(ctx, hipEventRecord(ctx->ev_band, C));
            src.band_gave_up = buf<uint32_t>(ctx, SB_BANDCTR) + 1;
        } else {
        HIPCHK(ctx, hipEventRecord(ctx->ev_t0, S));
        if (band_rows > 0) {
            if (band_shards) {   // the cursors are zeroed HERE, ahead of the event the other launch's stream waits for
                HIPCHK(ctx, hipMemsetAsync(ctx->d_counts, 0, HMK_EDGE_SHARDS * sizeof(unsigned long long), S));
                HIPCHK(ctx, hipEventRecord(ctx->ev_rest, S));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->rest_stream, ctx->ev_rest, 0));
            }
            st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, 1, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S,
                                      band_shards ? LAUNCH_BAND_NOZERO : LAUNCH_BAND, band_req, d_deg, d_deg_lo, d_rank, 0,
                                      band_shards ? band_shards : HMK_EDGE_SHARDS);
            if (st) return st;
            HIPCHK(ctx, hipMemcpyAsync(buf<void>(ctx, SB_BCOUNTS), ctx->d_counts, HMK_EDGE_SHARDS * sizeof(unsigned long long),
                                       hipMemcpyDeviceToDevice, S));
            HIPCHK(ctx, hipEventRecord(ctx->ev_band, S));
            call_lap("band tiles enqueued");
        }
        if (band_rows > 0 && band_shards) {
            st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, 1, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, ctx->rest_stream,
                                      LAUNCH_REST, band_req, d_deg, d_deg_lo, d_rank, band_shards, HMK_EDGE_SHARDS - band_shards);
            if (st == HMK_OK) {
                HIPCHK(ctx, hipEventRecord(ctx->ev_rest, ctx->rest_stream));
                HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_rest, 0));
            }
        } else {
            st = neighbors_dev_locked(ctx, max_shift, shift_penalty, threshold, 0, 1, ctx->d_edges, ctx->d_edges_cap, ctx->d_counts, S,
                                      band_rows > 0 ? LAUNCH_REST : LAUNCH_ALL, band_req, d_deg, d_deg_lo, d_rank);
        }
        call_lap("all tiles enqueued");
        if (st) { (void)hipStreamSynchronize(ctx->rest_stream); (void)hipStreamSynchronize(S); return st; }
        }
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
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stats->neighbors_ms = ctx->phases.score_ms;
    if (getenv("HMK_GREEDY_TIMING"))
        fprintf(stderr, "[hmk greedy] call %.2f ms: streams/events/pinned block %.2f, plan %.2f, %d buffer (re)allocations %.2f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(),
                std::chrono::duration<double, std::milli>(t0 - t_entry).count(), ctx->phases.plan_ms, g_allocs, g_alloc_ms);
    return st;
}

}  // extern "C"

namespace {

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
    int64_t band_req = 0;
    if (!clink && max_clusters > 0 && n >= 16384 && getenv("HMK_NO_BAND") == nullptr) band_req = std::min<int64_t>(n, 2LL * max_clusters + 1024);
    if (band_req * 2 > (int64_t)n) band_req = 0;
    uint64_t guess = (uint64_t)((double)n * (n - 1) / 2 * (ctx->symmetric ? 0.003 : 0.006) / G * 1.25) + (1u << 20);
    if (const char *v = getenv("HMK_EDGE_GUESS")) guess = std::strtoull(v, nullptr, 10);   // tests: force the overflow / retry path
    const long long top = (long long)ctx->max_len * std::max(0, ctx->max_m) +
                          (long long)std::max(0, shift_penalty) * ((ctx->max_len - ctx->min_len) + 2LL * max_shift);
    // every device counts the row degrees of the edges it writes (the CSR's first pass, fused into the scoring as in the
    // single-device call); the peers' counters travel with their blocks and are added to the root's
    const bool fuse = ctx->symmetric && getenv("HMK_NO_FUSED_DEGREE") == nullptr;
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
        const bool packed = top - threshold <= 255 && getenv("HMK_ADJ_8BYTE") == nullptr;
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
            if (csr_by_bucket(n, ctx->symmetric, packed, false)) {
                HIPCHK(ctx, ensure_buf(ctx, SB_PART, (all_cap + 1) * 8));
                HIPCHK(ctx, ensure_buf(ctx, SB_PARTSCR, csr_partition_scratch_bytes()));
            }
        }
        const int root_dev = ctx->device;
        uint64_t *root_peer = buf<uint64_t>(ctx, SB_PEER), *root_band = buf<uint64_t>(ctx, SB_PEERBAND);
        unsigned long long *root_cnt = buf<unsigned long long>(ctx, SB_PEERCNT);

        // ---- a peer's whole share: plan, band tiles, band block, the rest, the whole block; each hand-over as soon as it can go ----
        auto peer_main = [&](PeerJob &J) {
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
        for (auto &jp : jobs) { PeerJob *J = jp.get(); J->th = std::thread([&peer_main, J]() { peer_main(*J); }); }
        struct Joiner {   // on every way out: the workers are done before their state goes away
            std::vector<std::unique_ptr<PeerJob>> &jobs;
            ~Joiner() { for (auto &jp : jobs) if (jp->th.joinable()) jp->th.join(); }
        } joiner{jobs};

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
    ctx->phases.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) stats->neighbors_ms = ctx->phases.score_ms;
    if (clink) clink->neighbors_ms = ctx->phases.score_ms;
    return st;
}

}  // namespace

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
        if (G > 1 && getenv("HMK_EDGE_GUESS") == nullptr)
            cap = std::max<uint64_t>({(uint64_t)((double)cap / G * 1.25), (uint64_t)1 << 20, ctx->d_edges_cap});
        st = grow_edge_buffer(ctx, (cap + HMK_EDGE_SHARDS - 1) / HMK_EDGE_SHARDS * HMK_EDGE_SHARDS);
    }
    if (st) return st;
    const int64_t maxc = (int64_t)(n_sequences * 0.025 + 0.5);      // Hammock.java:398-401, the default cluster limit
    int64_t band = n_sequences >= 16384 ? std::min<int64_t>(n_sequences, 2 * maxc + 1024) : 0;
    if (band * 2 > (int64_t)n_sequences) band = 0;
    st = reserve_tail_buffers(ctx, n_sequences, true, (uint32_t)band, true, getenv("HMK_NO_LATE_BUFFERS") == nullptr);
    if (getenv("HMK_GREEDY_TIMING") || getenv("HMK_CLI_TIMING")) {
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
    ctx->java_hashset = version;
    for (hmk_ctx *peer : ctx->peers) peer->java_hashset = version;
    return HMK_OK;
}

int hmk_device_count(const hmk_ctx *ctx) { return ctx ? (ctx->has_device ? 1 + (int)ctx->peers.size() : 0) : 0; }

int hmk_greedy_from_edges_dev(hmk_ctx *ctx, const void *d_edges, uint64_t n_edges, int symmetric, int max_clusters,
                              int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats) {
    if (!ctx) return fail(nullptr, HMK_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    src.packed = top - threshold <= 255 && getenv("HMK_ADJ_8BYTE") == nullptr;
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
