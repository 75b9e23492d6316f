// hmk_ctx.h -- the context behind the C ABI and what the translation units of libhammock_hip.so's host side share:
// hmk_api.cpp (the extern "C" entry points), hmk_common.cpp (errors, device, grow-only buffers, streams), hmk_plan.cpp (the
// neighbour passes' planner), hmk_pass.cpp (launching the passes; the pair and block probes), hmk_cluster.cpp (CSR pipeline, row
// hand-over, second-loop driver), hmk_multi.cpp (one process, several devices).  Not part of the public ABI.
#ifndef HMK_CTX_H
#define HMK_CTX_H
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

namespace hmk { namespace impl {

struct Group {
    int path;
    int nw;
    int lbk;  // column-length capacity of the kernel instantiation
    uint32_t base, count;
    uint32_t band;  // the first `band` tiles of the group touch a "band" row (caller index < Plan::band_rows)
    uint64_t work;  // cells its tiles add (pairs x cells per pair): what the launches of a forked pass are balanced by (hmk_pass.cpp)
};

// grow-only device scratch of the greedy tail (one hipMalloc per buffer and context, not per call)
struct DevBuf { void *p = nullptr; size_t cap = 0; };
enum {
    SB_DEG, SB_CURSOR, SB_START, SB_SCAN, SB_RANGE, SB_ADJ, SB_PART, SB_PARTSCR,   // full CSR (+ the bucketed lower sections)
    SB_BDEG, SB_BCURSOR, SB_BSTART, SB_BSCAN, SB_BRANGE, SB_BADJ, SB_BCOUNTS,     // band CSR (first rows only)
    SB_BNCNT, SB_BNUP, SB_BNSTART, SB_BFTOP, SB_BFMORE, SB_FDEG, SB_FSTART, SB_FADJ, SB_TRCNT, SB_TRSTART,   // the band prepared for phase 1 (BandPack)
    SB_COF, SB_BITMAP, SB_USIZE, SB_LEFT, SB_CNT, SB_CSTART, SB_OVER, SB_SCAN2, SB_CAND,             // pre-check of the second loop
    SB_JOINED, SB_CSIZE, SB_CID, SB_SEQSZ, SB_STATUS, SB_CHOICE, SB_FIRST, SB_ACTIVE, SB_DIRTY, SB_SUBS2, SB_RETRY, SB_PRECNT, SB_ACCEPTED, SB_JSLOT, SB_LCOUNT, SB_SUBSTART, SB_SUBS,   // device-side second loop
    SB_PEER, SB_PEERCNT, SB_PEERBAND, SB_PEERDEG,   // multi-device calls: the inbox of edge blocks from the other devices + their counts, band blocks (peers: compacted; root: gathered), degree slices
    SB_ROUTE, SB_ROUTECNT,                          // ... this device's edges dealt into one block per owning device; counts / offsets / cursors
    SB_REPL,                                        // ... (in a peer's context, on the ROOT's device) the peer's piece copied to the root: no peer access, or HMK_MULTI_REPLICATE
    SB_N
};

// The library's environment switches (INTEGRATION.md lists them: every one is a test or diagnostic aid, none is needed for a
// result or for speed).  Read ONCE per entry-point call (refresh_switches, under the context's lock) -- never from a launch loop
// or a worker thread.
struct Switches {
    bool greedy_timing = false;   // HMK_GREEDY_TIMING: a clustering call's timeline, the plan's and the merge's phases on stderr
    bool no_band = false;         // HMK_NO_BAND: no band hand-over, phase 1 waits for the whole pass
    bool no_rows_kernel = false;  // HMK_NO_ROWS_KERNEL: the shift-packed tier (k_neighbors.hip) for every class
    bool adj_8byte = false;       // HMK_ADJ_8BYTE: 8-byte adjacency entries even where 4 bytes hold every score
    bool local_literal = false;   // HMK_LOCAL_LITERAL: LocalAlignmentScorer through the literal DP
    bool local_signed = false;    // HMK_LOCAL_SIGNED: the signed tagged-max form (what gap_open = 0 runs) for every penalty
    bool local_no_pk = false;     // HMK_LOCAL_NO_PK: one column sequence per lane in the tagged-max DP
    bool multi_replicate = false; // HMK_MULTI_REPLICATE: a multi-device root copies the peers' finished adjacency pieces to itself instead of
                                  // reading them in place (what a machine without peer access between two of its devices runs)
    int second_loop = 0;          // HMK_SECOND_LOOP=device|host: 1 / 2 force that implementation of the second loop (0: the default choice)
    int phase1_threads = 0, phase1_window = 0;        // HMK_PHASE1_THREADS, HMK_PHASE1_WINDOW (GreedyOptions)
    int host_band_rows = 0, host_band_far_t = 0;      // HMK_PHASE1_HOST_BAND=rows[,far_t] (GreedyOptions)
    int loop_chain = -1;          // HMK_LOOP_CHAIN=0|1: chained joins never / from the first round (-1: from round 256 on)
    int loop_passes = 0;          // HMK_LOOP_PASSES: first/accept passes per round of the device-side loop (0: by cluster count)
    int precheck = 0;             // HMK_PRECHECK=two_passes|one_stage: 1 = count + fill passes, 2 = full-size tables for every row at once
    int late_buffers_delay_ms = 0;   // HMK_LATE_BUFFERS_DELAY_MS: hmk_reserve's buffer thread sleeps first (a host where device memory is slow to get)
    int csr_bucket_shift = 0;     // HMK_CSR_BUCKET_SHIFT: rows per bucket = 2^shift in the CSR's dealing pass (the wide buckets of n > 2^21)
    uint64_t edge_guess = 0;      // HMK_EDGE_GUESS: first edge-buffer capacity (forces the overflow / retry path)
    void read();
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
    bool no_rows_kernel = false;   // the switch this plan was built under (part of the cache key)
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

} }  // namespace hmk::impl
using namespace hmk::impl;

struct hmk_ctx {
    int32_t M[HMK_ALPHABET * HMK_ALPHABET];
    bool symmetric = true;
    int min_m = 0, max_m = 0;
    int device = -1;
    bool has_device = false;
    int java_hashset = 8;   // hmk_set_java_hashset: whose HashSet iteration order clinkage emulates
    Switches sw;            // this call's environment switches (refresh_switches)

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
    static constexpr int N_SIDE = 3;     // streams a mixed-length pass's launches are dealt to (hmk_pass.cpp)
    hipStream_t side[N_SIDE] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[N_SIDE] = {nullptr};
    hipStream_t copy_stream = nullptr;   // band CSR + device-to-host copies of adjacency rows (hmk_greedy_cluster)
    uint32_t *d_rows_scratch = nullptr;  // deg[n], cursor[n], misfit of hmk_pack_rows_dev
    uint32_t d_rows_scratch_n = 0;

    double last_kernel_ms = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // greedy tail: own stream + events, grow-only device scratch, pinned host staging (all made once per context)
    hipStream_t gstream = nullptr;
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
    // the stream this device's blocks travel to the other devices on (multi-device calls; a stream of this device)
    hipStream_t xfer_stream = nullptr;
    bool peer_loads_ok = true;   // (a peer:) the root's kernels may read this device's memory in place

    // hmk_reserve sizes the two buffers a clustering call needs LAST (adjacency, bucket records: 2 x 11 GB at 10^6) on its own
    // thread: on some hosts a fresh 11 GB of device memory takes 0.3-1.5 s to get, and a call has 0.27 s of scoring to do
    // before it writes to them.  Whoever touches SB_ADJ / SB_PART joins this first (ensure_buf does).
    std::future<hipError_t> late_buffers;

    bool wedged = false;   // a clustering call gave a stalled device up: nothing waits for it any more (calls fail with HMK_ERR_DEVICE)
    std::string err;
    mutable std::mutex mu;
};

namespace hmk { namespace impl {

extern thread_local std::string g_last_error;
int fail(hmk_ctx *ctx, int code, const std::string &msg);

#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, e_ == hipErrorOutOfMemory ? HMK_ERR_OOM : HMK_ERR_DEVICE,          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

// a HIP call whose failure is not this call's business (teardown): the error must not stay behind as the thread's last one
#define HMK_QUIET(call) do { if ((call) != hipSuccess) (void)hipGetLastError(); } while (0)

// which: LAUNCH_ALL, or only the band tiles of the plan (LAUNCH_BAND: also zeroes the counts) / only the others
// (LAUNCH_REST: appends to the counts of the band launch)
enum { LAUNCH_ALL = 0, LAUNCH_BAND = 1, LAUNCH_REST = 2 };

constexpr int ST_RETRY_OVERFLOW = 1000;   // internal: an edge segment overflowed, grow the buffer and score again
// layout of the small pinned block hmk_ctx::h_counts (64-bit words)
enum { HC_COUNTS = 0, HC_BAND = HMK_EDGE_SHARDS, HC_PEER = 2 * HMK_EDGE_SHARDS, HC_RANGE = HC_PEER + 32, HC_MISC = HC_RANGE + 8, HC_TOTAL = HC_MISC + 8, HC_WORDS = HC_TOTAL + 16 };

// (HMK_GREEDY_TIMING: what the grow-only buffers cost a call, i.e. the first call of a context)
extern thread_local double g_alloc_ms;
extern thread_local int g_allocs;
struct AllocTimer {
    const char *what;
    size_t bytes;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    AllocTimer(const char *w, size_t b) : what(w), bytes(b) {}
    ~AllocTimer() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        g_alloc_ms += ms;
        g_allocs++;
        static const bool timing = getenv("HMK_GREEDY_TIMING") != nullptr;   // (a destructor without a context: read once per process)
        if (timing && ms > 5.0) std::fprintf(stderr, "[hmk] %s of %.1f MB took %.1f ms\n", what, (double)bytes / 1048576.0, ms);
    }
};

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
    EdgeSegs band_segs{};
    bool deg_fused = false;            // the neighbour kernel counted the rows' degrees while it wrote the edges (SB_DEG, zeroed before the pass)
    bool deg_split = false;            // ... as upper counts [0, n) and lower counts [n, 2n) instead of totals
    // Multi-device calls: the adjacency is built in PIECES, piece d = the rows [d * rows_per, (d + 1) * rows_per) on device d, by that
    // device's worker (hmk_multi.cpp): `segs` is not used; before_full says when every piece's CSR has been built, precheck_pieces runs
    // the second loop's pre-check on every device for the leftovers whose rows it holds.
    struct Piece { hmk_ctx *c; uint32_t r0, r1; };
    std::vector<Piece> pieces;         // empty: one piece, the whole graph on this context
    uint32_t rows_per = 0;
    uint64_t total_edges = 0;          // (pieces: set by before_full) edges of all devices: the pieces' entries must add up to twice that
    std::function<bool(const struct PreIn &, unsigned long long *total)> precheck_pieces;   // -> every piece fits; *total = candidate entries
    hmk_clinkage_stats *clink = nullptr;   // non-null: run the clinkage nearest-neighbour chain instead of the greedy merge
    // multi-device calls: the peers' blocks arrive while the calling thread is already inside cluster_on_device.
    //   before_band  blocks until every peer's band block is on its way to the root and makes the copy stream wait for them;
    //                non-zero: no band hand-over in this call (phase 1 then waits for the full graph)
    //   before_full  blocks until every device's piece of the adjacency is complete (its size in the device's h_counts[HC_TOTAL]);
    //                HMK_OK, ST_RETRY_OVERFLOW or an error code (the text is in the context)
    std::function<int()> before_band, before_full;
};

// ---- hmk_common.cpp
void refresh_switches(hmk_ctx *ctx);              // the root's and its peers' (under the context's lock, once per entry-point call)
GreedyOptions greedy_options(const hmk_ctx *ctx);
int need_device(hmk_ctx *ctx);
int ensure_res32(hmk_ctx *ctx);
hipError_t ensure_buf_now(hmk_ctx *ctx, int which, size_t bytes);
bool late_buffers_pending(hmk_ctx *ctx);
hipError_t join_late_buffers(hmk_ctx *ctx);
hipError_t ensure_buf(hmk_ctx *ctx, int which, size_t bytes);
template <class T> T *buf(hmk_ctx *ctx, int which) { return (T *)ctx->sb[which].p; }
hipError_t ensure_pinned(void **p, size_t *cap, size_t bytes, size_t keep);
int greedy_streams(hmk_ctx *ctx);   // streams, events and pinned blocks of the clustering calls
bool csr_by_bucket(bool symmetric, bool packed);
int reserve_tail_buffers(hmk_ctx *ctx, uint32_t n, bool packed, uint32_t r1, bool full = false, bool late_on_a_thread = false);
uint64_t first_edge_capacity(const hmk_ctx *ctx, uint32_t n);
int grow_edge_buffer(hmk_ctx *ctx, uint64_t cap);
// ---- hmk_plan.cpp
void free_plan(Plan &pl);
void classify(const hmk_ctx *ctx, int la, int lb, int X, int p, int thr, TileClass *out, long long row_bound = -1,
              long long *u8_row_limit = nullptr);
int build_plan(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, int64_t band_rows = -1);
void free_plan_local(PlanLocal &pl);
int build_plan_local(hmk_ctx *ctx, uint32_t part, uint32_t n_parts);
// ---- hmk_pass.cpp
int neighbors_dev_locked(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, void *d_edges,
                         uint64_t capacity, void *d_counts, hipStream_t stream, int which = LAUNCH_ALL,
                         int64_t band_rows = -1, uint32_t *d_deg = nullptr, uint32_t *d_deg_lo = nullptr);
bool local_enc(const hmk_ctx *ctx, int gap_open, int gap_extend);
int neighbors_internal(hmk_ctx *ctx, int X, int p, int thr, uint32_t part, uint32_t n_parts, uint64_t want_cap,
                       unsigned long long counts[HMK_EDGE_SHARDS], double *kernel_ms);
int neighbors_local_dev_locked(hmk_ctx *ctx, int gap_open, int gap_extend, int thr, uint32_t part, uint32_t n_parts,
                               uint64_t *d_edges, uint64_t capacity, unsigned long long *d_counts, hipStream_t stream);
void timer_start(hmk_ctx *ctx);
void timer_stop(hmk_ctx *ctx);
int check_pairs(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, bool shifted, int X);
int score_pairs(hmk_ctx *ctx, int scorer, const uint32_t *i, const uint32_t *j, uint64_t n_pairs, int a, int b,
                int32_t *out, int32_t *out_shift = nullptr);
int score_block(hmk_ctx *ctx, int scorer, uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int a, int b,
                int32_t *out);
// ---- hmk_cluster.cpp
// What the second loop's pre-check needs of the host (pinned, read-only while the pieces run): cluster_of | usize | leftover in one block
struct PreIn {
    uint32_t n = 0, nl = 0, ncl = 0;
    bool packed = true;
    const char *h_block = nullptr;       // pinned: cluster_of int32[n], usize int32[ncl], leftover uint32[nl]
    size_t b_cof = 0, b_us = 0, b_left = 0;
    unsigned long long region_cap = 0;   // entries per region of the candidate buffer (HMK_PRE_REGIONS regions)
    int first_slots = 128;               // table size of the first stage
    bool two_stage = true;
};
// CSR of the rows [r0, r1) from `segs` on c's device, enqueued on q (counts or the fused counters, scan, scatter); records c->ev_csr;
// the piece's entry count lands in c->h_counts[HC_TOTAL], the score range / invalid edges in c->h_counts + HC_RANGE
hipError_t piece_enqueue_csr(hmk_ctx *c, const EdgeSegs &segs, bool symmetric, bool packed, int base, uint32_t n, uint32_t r0, uint32_t r1,
                             bool deg_fused, bool deg_split, hipStream_t q);
// the pre-check of one piece (k_greedy_precheck, single pass with region counters): upload on c's copy stream, kernels on q behind the piece's
// CSR, counters back, q synchronised.  -> 0: fits (entries in *total), 1: a region overran (count + fill passes needed), -1: failed / a row
// overflowed its tables (the host's pre-check)
int piece_precheck(hmk_ctx *c, const PreIn &in, uint32_t r0, uint32_t r1, uint32_t region_base, uint32_t region_count, hipStream_t q,
                   bool upload, unsigned long long *total);
hipError_t piece_precheck_upload(hmk_ctx *c, const PreIn &in);   // (the upload alone, on c's copy stream: records c->ev_bandcsr)
int cluster_on_device(hmk_ctx *ctx, const EdgeSource &src, int max_clusters, int32_t *cluster_id, int32_t *result_order,
                      int32_t *member_rank, hmk_greedy_stats *stats, std::chrono::steady_clock::time_point t0);
// ---- hmk_multi.cpp
int greedy_cluster_multi(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int max_clusters, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats, hmk_clinkage_stats *clink = nullptr);

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

} }  // namespace hmk::impl
#endif
