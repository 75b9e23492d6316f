// hmk_greedy.cpp -- host-side exact greedy merge on the GPU's thresholded
// neighbour graph.  Pure C++ (no HIP): part of the product path, it is what
// north_star calls "the host-side greedy merge".
//
// It reproduces LimitedGreedySequenceClusterer.cluster
// (LimitedGreedySequenceClusterer.java:39-120) exactly, but instead of calling
// the scorer pair by pair it reads the edge list the neighbour kernel produced:
//
//   ClinkageClusterScorer.clusterScore(c, x) (ClinkageClusterScorer.java:30-49)
//     = min over members m of score(m, x), or MIN_VALUE+1 as soon as one
//       score is below the threshold
//   => c is "feasible" for x  <=>  every member of c is a neighbour of x in the
//      >= threshold graph, and then the cluster score is the min of the stored
//      edge scores.  One pass over adj[x] with a per-cluster counter decides
//      all clusters at once.
//
//   findNearestClusterParallel (ClinkageSequenceClusterer.java:137-177,258-293)
//     = arg-max over feasible clusters of (score, Cluster.size(), -id);
//       null when nothing is feasible; a non-null dummy (cluster == null,
//       score == MIN_VALUE) when the candidate list is empty.
//
// The three branches where the reference dereferences the dummy's null cluster
// (:97, :104, :108) are reported as HMK_ERR_REFERENCE_WOULD_CRASH.
#include "hmk_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include <sched.h>

namespace hmk {

unsigned usable_cpus() {
    unsigned n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (n == 0) n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    std::ifstream f("/sys/fs/cgroup/cpu.max");   // cgroup v2: "<quota> <period>" or "max <period>"
    std::string quota;
    long long period = 0;
    if (f && (f >> quota >> period) && quota != "max" && period > 0) {
        const long long q = std::atoll(quota.c_str()) / period;
        if (q >= 1 && (unsigned)q < n) n = (unsigned)q;
    }
    return n;
}

namespace {

enum : uint8_t { ST_FREE = 0, ST_IN_CLUSTER = 1, ST_ORPHAN = 2 };
enum { NEAR_NULL = 0, NEAR_DUMMY = 1, NEAR_REAL = 2 };

struct ClusterRec {
    int32_t id;      // seed index (LimitedGreedySequenceClusterer.java:82)
    int32_t usize;   // Cluster.getUniqueSize()
    int64_t size;    // Cluster.size(): sum of member counts (Cluster.java:156)
};

struct Found {
    int kind;
    int32_t slot;   // cluster slot (A / phase 2) or sequence index (B)
    int32_t score;
};

// (score, size, -id) total order of NearestClusterRunner.call :265-289
inline bool better(int32_t s, int64_t size, int32_t id, int32_t bs, int64_t bsize, int32_t bid) {
    if (s != bs) return s > bs;
    if (size != bsize) return size > bsize;
    return id < bid;
}

}  // namespace

int greedy_from_edges(uint32_t n, const int32_t *sizes, const uint64_t *edges, uint64_t n_edges,
                      bool symmetric, int threshold, int max_clusters, int32_t *cluster_id,
                      int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st, std::string *err,
                      const GreedyOptions &opt) {
    auto t0 = std::chrono::steady_clock::now();
    (void)threshold;  // every stored edge already satisfies score >= threshold
    // ---- CSR adjacency: adj[x] = {(m, sequenceScore(m, x))} -----------------
    std::vector<uint64_t> start((size_t)n + 1, 0);
    for (uint64_t e = 0; e < n_edges; e++) {
        uint32_t x = HMK_EDGE_X(edges[e]), m = HMK_EDGE_M(edges[e]);
        if (x >= n || m >= n || x == m) {
            if (err) *err = "edge list references a sequence outside [0, n) or a self pair";
            return HMK_ERR_BAD_ARG;
        }
        start[x + 1]++;
        if (symmetric) start[m + 1]++;
    }
    for (uint32_t k = 0; k < n; k++) start[k + 1] += start[k];
    std::vector<Nbr> adj(start[n]);
    {
        std::vector<uint64_t> fill(start.begin(), start.end() - 1);
        for (uint64_t e = 0; e < n_edges; e++) {
            uint32_t x = HMK_EDGE_X(edges[e]), m = HMK_EDGE_M(edges[e]);
            int32_t s = HMK_EDGE_SCORE(edges[e]);
            adj[fill[x]++] = Nbr{m, s};
            if (symmetric) adj[fill[m]++] = Nbr{x, s};
        }
    }
    // HMK_PHASE1_HOST_BAND: the prepared band (BandPack) built here, literally, from the whole graph
    GreedyHooks hooks;
    std::vector<uint32_t> b_near_start, b_near_up, b_near, b_far_top, b_near_top, b_tr_cnt, b_tr_start, b_tr;
    std::vector<uint8_t> b_far_more;
    BandPack pack;
    bool use_pack = opt.host_band_rows > 0 && symmetric && n > 0;
    if (use_pack)
        for (const Nbr &a : adj)
            if (a.s - threshold < 0 || a.s - threshold > 255) { use_pack = false; break; }
    // (what the hooks below call lives as long as they do: declared here, not inside the block that fills the pack)
    const uint32_t R1 = std::min<uint32_t>((uint32_t)std::max(0, opt.host_band_rows), n), FT = (uint32_t)std::max(1, opt.host_band_far_t > 0 ? opt.host_band_far_t : 8);
    auto ent = [&](uint32_t id, int32_t s) { return id << 8 | (uint32_t)(s - threshold); };
    auto far_key = [&](const Nbr &a) {   // (score, Cluster.size(), smaller id), larger = better
        return std::make_tuple(a.s, (int64_t)(sizes ? sizes[a.m] : 1), -(int64_t)a.m);
    };
    auto band_nbrs = [&](uint32_t id, std::vector<uint32_t> &out) {   // the rows below R1 that have `id` as a neighbour
        for (uint64_t q = start[id]; q < start[id + 1]; q++)
            if (adj[q].m < R1) out.push_back(ent(adj[q].m, adj[q].s));
    };
    if (use_pack) {
        b_near_start.assign((size_t)R1 + 1, 0);
        b_near_up.assign(R1, 0);
        b_far_top.assign((size_t)R1 * FT, ~0u);
        b_near_top.assign((size_t)R1 * BandPack::NEAR_T, ~0u);
        b_far_more.assign(R1, 0);
        b_tr_start.assign((size_t)BandPack::TR_PER_ROW * R1 + 1, 0);
        b_tr_cnt.assign((size_t)BandPack::TR_PER_ROW * R1, 0);
        std::vector<int32_t> of_x(R1, INT_MIN);   // score(x, row) for the rows above x that are its neighbours
        for (uint32_t x = 0; x < R1; x++) {
            std::vector<Nbr> far, near_above;
            for (int pass = 0; pass < 2; pass++)   // the neighbours above x first
                for (uint64_t q = start[x]; q < start[x + 1]; q++) {
                    const Nbr &a = adj[q];
                    if (a.m >= R1) { if (pass == 0) far.push_back(a); continue; }
                    if ((pass == 0) == (a.m > x)) { b_near.push_back(ent(a.m, a.s)); if (pass == 0) { b_near_up[x]++; near_above.push_back(a); } }
                }
            std::sort(near_above.begin(), near_above.end(), [&](const Nbr &p, const Nbr &q) { return far_key(p) > far_key(q); });
            for (uint32_t t = 0; t < BandPack::NEAR_T && t < near_above.size(); t++) b_near_top[(size_t)x * BandPack::NEAR_T + t] = ent(near_above[t].m, near_above[t].s);
            b_near_start[x + 1] = (uint32_t)b_near.size();
            std::sort(far.begin(), far.end(), [&](const Nbr &p, const Nbr &q) { return far_key(p) > far_key(q); });
            for (uint32_t t = 0; t < FT && t < far.size(); t++) b_far_top[(size_t)x * FT + t] = ent(far[t].m, far[t].s);
            b_far_more[x] = far.size() > FT;
            // the later band rows that have both x and the candidate as neighbours, with the smaller of the two scores; every fifth slot
            // is left unprepared (~0u), as the device leaves a row beyond its table: the loop then filters the whole list itself
            for (uint64_t q = start[x]; q < start[x + 1]; q++)
                if (adj[q].m > x && adj[q].m < R1) of_x[adj[q].m] = adj[q].s;
            for (uint32_t t = 0; t < BandPack::TR_PER_ROW; t++) {
                const size_t u = (size_t)BandPack::TR_PER_ROW * x + t;
                if (t < FT && t < far.size()) {
                    if ((x + t) % 5 == 4) b_tr_cnt[u] = ~0u;
                    else {
                        const uint32_t c = far[t].m;
                        for (uint64_t q = start[c]; q < start[c + 1]; q++) {
                            const uint32_t y = adj[q].m;
                            if (y > x && y < R1 && of_x[y] != INT_MIN) { b_tr.push_back(ent(y, std::min(of_x[y], adj[q].s))); b_tr_cnt[u]++; }
                        }
                    }
                }
                b_tr_start[u + 1] = (uint32_t)b_tr.size();
            }
            for (uint64_t q = start[x]; q < start[x + 1]; q++)
                if (adj[q].m > x && adj[q].m < R1) of_x[adj[q].m] = INT_MIN;
        }
        pack.rows = R1; pack.far_t = FT;
        pack.near_start = b_near_start.data(); pack.near_up = b_near_up.data(); pack.near = b_near.data();
        pack.far_top = b_far_top.data(); pack.far_more = b_far_more.data();
        if (opt.host_band_far_t % 2 == 0) pack.near_top = b_near_top.data();   // (odd far_t: without the list, the plain scan)
        pack.tr_cnt = b_tr_cnt.data(); pack.tr_start = b_tr_start.data(); pack.tr = b_tr.data();
        hooks.band_pack = [&]() -> const BandPack * { return &pack; };
        hooks.far_row = [&](uint32_t id, std::vector<uint32_t> &out) -> bool { out.clear(); band_nbrs(id, out); return true; };
        hooks.band_far = [&](uint32_t x, std::vector<uint32_t> &out) -> bool {
            out.clear();
            for (uint64_t q = start[x]; q < start[x + 1]; q++)
                if (adj[q].m >= R1) out.push_back(ent(adj[q].m, adj[q].s));
            return true;
        };
        hooks.need_rows = [&](uint32_t, uint32_t) -> uint32_t { return n; };   // everything is here
    }
    const int rc = greedy_from_csr(n, sizes, start.data(), adj.data(), nullptr, use_pack ? &hooks : nullptr, symmetric, max_clusters, cluster_id,
                                   result_order, member_rank, st, err, opt);
    if (st) {
        st->n_edges = n_edges;
        st->greedy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

// The merge proper, on a CSR adjacency: start[n + 1], adj[start[x] .. start[x + 1]) = neighbours of x.
// NbrT: Nbr (8 bytes) or NbrPacked (4 bytes); only id() and score() are used, scores only in comparisons.
template <class NbrT>
static int greedy_from_csr_impl(uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrT *adj_in,
                                const uint32_t *upper, const GreedyHooks *hooks, bool symmetric_scores, int max_clusters,
                                int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st,
                                std::string *err, const GreedyOptions &opt) {
    auto t0 = std::chrono::steady_clock::now();
    hmk_greedy_stats local;
    if (!st) st = &local;
    std::memset(st, 0, sizeof(*st));
    const bool partial = hooks && hooks->need_rows;   // only a prefix of the rows is on the host so far
    const NbrT *adj = adj_in;                         // re-read after every need_rows(): the host buffer may have moved
    if (!partial) st->n_edges = start[n];
    double t_phase1 = 0;

    std::vector<uint8_t> state(n, ST_FREE);
    std::vector<int32_t> cluster_of(n, -1);
    std::vector<ClusterRec> clusters;
    std::vector<uint32_t> orphans;
    auto seq_size = [&](uint32_t k) -> int64_t { return sizes ? sizes[k] : 1; };

    // nearest(clusters, x): findNearestClusterParallel over actualClusters (the second loop's plain path, :59-66)
    std::vector<int32_t> ncnt, nmn, ntouched;          // per-cluster scratch of the feasibility pass
    auto nearest_cluster = [&](uint32_t x) -> Found {
        if (clusters.empty()) return Found{NEAR_DUMMY, -1, INT_MIN};  // :138-140
        if (ncnt.size() < clusters.size()) { ncnt.resize(clusters.size(), 0); nmn.resize(clusters.size(), 0); }
        ntouched.clear();
        for (uint64_t q = start[x]; q < start[x + 1]; q++) {
            int32_t c = cluster_of[adj[q].id()];
            if (c < 0) continue;
            if (ncnt[c] == 0) { ntouched.push_back(c); nmn[c] = adj[q].score(); }
            else if (adj[q].score() < nmn[c]) nmn[c] = adj[q].score();
            ncnt[c]++;
        }
        Found best{NEAR_NULL, -1, 0};
        for (int32_t c : ntouched) {
            if (ncnt[c] == clusters[c].usize) {  // complete linkage: all members >= threshold
                if (best.kind == NEAR_NULL ||
                    better(nmn[c], clusters[c].size, clusters[c].id, best.score,
                           clusters[best.slot].size, clusters[best.slot].id))
                    best = Found{NEAR_REAL, c, nmn[c]};
            }
            ncnt[c] = 0;
        }
        return best;
    };

    if (member_rank)
        for (uint32_t q = 0; q < n; q++) member_rank[q] = 0;
    auto insert_into = [&](int32_t c, uint32_t k) {  // Cluster.insertAll of a singleton
        if (member_rank) member_rank[k] = clusters[c].usize;
        cluster_of[k] = c;
        state[k] = ST_IN_CLUSTER;
        clusters[c].usize++;
        clusters[c].size += seq_size(k);
    };

    // ---- firstPhase, LimitedGreedySequenceClusterer.java:77-120 ------------
    // One pass over the row of x answers both searches of a step: nearest(clusters, x) (:92, findNearestClusterParallel over
    // actualClusters: per-cluster counters over the neighbours that are members) and nearest(initial[idx+1:], x) (:93: the best
    // neighbour that is still an untouched singleton; every earlier position has been clustered, orphaned or absorbed).
    //
    // The steps are sequential in the reference, but a step reads little: the states of x's neighbours and the clusters they
    // are in.  So the rows of a WINDOW of upcoming positions are scanned by several threads against the state at the window's
    // start, and the steps are then committed in order; a commit invalidates exactly the scans it can have changed:
    //   * k itself changes state (member, seed or orphan): every later row of the window that has k as a neighbour -- known
    //     from k's own row (symmetric scores) -- is scanned again at its turn;
    //   * k absorbs B (:99-101, :108-110): B stops being a candidate -- rows whose best candidate was B are scanned again; the
    //     new cluster {k, B} can only be feasible for rows that have k as a neighbour (above);
    //   * k joins cluster c (:97, :104): for rows that do not have k as a neighbour c stops being feasible (complete linkage:
    //     every member must be a neighbour) -- c is struck from their feasible lists.
    // A row's scan leaves out the window's earlier positions as candidates (they are decided before its turn).  Results are
    // those of the sequential loop, step by step; with one thread the window is one row.
    struct RowScan {
        std::vector<std::pair<int32_t, int32_t>> feas;   // (cluster slot, min score) of the clusters feasible at the scan
        std::vector<std::pair<uint32_t, int32_t>> ahead;    // (id, score) of the FREE neighbours at later positions of the window
        std::vector<std::pair<uint32_t, int32_t>> behind;   // ... and at the window's earlier positions (decided before x's turn)
        Found B{NEAR_NULL, -1, 0};
        bool no_clusters = false;                        // the scan saw an empty cluster list (:138-140: the dummy)
        bool dirty = false;
    };
    struct ScanScratch { std::vector<int32_t> cnt, mn, touched; };
    const size_t slots_max = std::min<size_t>((size_t)std::max(max_clusters, 0), (size_t)n) + 2;   // cluster slots a scan can meet
    auto scan_row = [&](uint32_t x, uint32_t win_lo, uint32_t win_hi, ScanScratch &sc, RowScan &out) {
        out.feas.clear();
        out.ahead.clear();
        out.behind.clear();
        out.dirty = false;
        out.no_clusters = clusters.empty();
        Found B{NEAR_NULL, -1, 0};
        sc.touched.clear();
        const uint64_t q_end = start[x + 1];
        for (uint64_t q = start[x]; q < q_end; q++) {
            // the states are random reads from an n-byte array: ask for the one needed 32 entries from now
            if (q + 32 < q_end) __builtin_prefetch(&state[adj[q + 32].id()], 0, 1);
            const uint32_t m = adj[q].id();
            const int32_t s = adj[q].score();
            const uint8_t stt = state[m];
            if (stt == ST_FREE) {                           // only untouched singletons follow x
                if (m < x) { if (m >= win_lo) { out.behind.emplace_back(m, s); continue; } }   // an earlier position of this window: decided before x's turn
                else if (m < win_hi) out.ahead.emplace_back(m, s);
                if (B.kind == NEAR_NULL || s > B.score ||
                    (s == B.score && better(s, seq_size(m), (int32_t)m, B.score, seq_size((uint32_t)B.slot), B.slot)))
                    B = Found{NEAR_REAL, (int32_t)m, s};
            } else if (stt == ST_IN_CLUSTER) {
                const int32_t c = cluster_of[m];
                if (sc.cnt[c] == 0) { sc.touched.push_back(c); sc.mn[c] = s; }
                else if (s < sc.mn[c]) sc.mn[c] = s;
                sc.cnt[c]++;
            }
        }
        for (int32_t c : sc.touched) {
            if (sc.cnt[c] == clusters[c].usize) out.feas.emplace_back(c, sc.mn[c]);   // complete linkage: all members >= threshold
            sc.cnt[c] = 0;
        }
        out.B = B;
    };

    unsigned T = 1;
    if (symmetric_scores && n >= 65536 && max_clusters >= 256) {
        const unsigned hw = usable_cpus();   // (the affinity mask cut by the cgroup quota: a container's share, not the host's 256)
        T = std::max(1u, std::min(8u, hw / 2));
    }
    bool pool_decided = false;                        // the threads start with the first window, if the rows are long enough (below)
    if (opt.phase1_threads > 0 && symmetric_scores) { T = (unsigned)std::min(32, opt.phase1_threads); pool_decided = true; }   // tests: any input, any thread count
    uint32_t W = T > 1 ? 4 * T : 1;                   // positions per window (more positions: more scans a commit invalidates)
    if (opt.phase1_window > 0) W = (uint32_t)std::min(4096, opt.phase1_window);
    if (!symmetric_scores) W = 1;   // (a commit patches the later rows' scans with its OWN row's scores: exact for symmetric scores only)
    std::vector<RowScan> res(W);
    std::vector<int32_t> ahead_score(W, INT_MIN);     // commit of k: score(k, x) for the window's later rows x that have k as a neighbour
    std::vector<ScanScratch> scratch(T);
    for (ScanScratch &sc : scratch) { sc.cnt.assign(slots_max, 0); sc.mn.assign(slots_max, 0); }
    // a small pool for the window scans: generation counter + work cursor, spinning workers (a window is tens of microseconds)
    std::vector<uint32_t> win_rows;                    // the window's FREE positions
    uint32_t win_lo = 0, win_hi = 0;
    std::atomic<uint32_t> gen{0}, cursor{0}, finished{0};
    // A worker that notices a generation late must not walk into the NEXT window's set-up (win_rows, win_lo / win_hi, the
    // cursor are plain data rewritten by this thread): a generation is open while its scans run; the main thread closes it and
    // waits until every worker that has entered has left (entered == left) before it touches the window again; a worker that
    // enters after the close sees that and leaves at once.  (Dekker-style: both sides use sequentially consistent operations,
    // so either the worker's entry is seen here or the close is seen there.)
    std::atomic<uint32_t> open_gen{0};
    std::atomic<uint64_t> entered{0}, left{0};
    std::atomic<bool> quit{false};
    auto run_window = [&](unsigned t) {
        for (;;) {
            const uint32_t i = cursor.fetch_add(1, std::memory_order_relaxed);
            if (i >= win_rows.size()) break;
            const uint32_t x = win_rows[i];
            scan_row(x, win_lo, win_hi, scratch[t], res[x - win_lo]);
            finished.fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    auto spawn_pool = [&]() {
    for (unsigned t = 1; t < T; t++)
        pool.emplace_back([&, t]() {
            uint32_t seen = 0;
            for (;;) {
                uint32_t g;
                unsigned spins = 0;
                while ((g = gen.load(std::memory_order_acquire)) == seen) {
                    if (quit.load(std::memory_order_acquire)) return;
                    if (++spins > 2000) { std::this_thread::yield(); spins = 0; }
                }
                seen = g;
                entered.fetch_add(1, std::memory_order_seq_cst);
                if (open_gen.load(std::memory_order_seq_cst) == g) run_window(t);
                left.fetch_add(1, std::memory_order_seq_cst);
            }
        });
    };
    struct PoolGuard {   // joins on every way out of the function (crash parity returns early; a thread that could not be started throws)
        std::vector<std::thread> &pool; std::atomic<bool> &quit;
        ~PoolGuard() { quit.store(true, std::memory_order_release); for (std::thread &th : pool) th.join(); }
    } pool_guard{pool, quit};
    if (pool_decided) spawn_pool();

    int64_t remaining = n;  // elements of initialList at positions >= index
    int64_t index = 0;
    uint32_t k = 0;         // sequence behind initialList.get(index)
    uint32_t rows_here = partial ? 0 : n;   // rows [0, rows_here) of the adjacency are on the host
    const bool p1_timing = opt.timing;
    double p1_scan = 0, p1_commit = 0, p1_rows = 0;
    uint64_t p1_windows = 0, p1_scanned = 0, p1_rescans = 0;
    auto p1_now = []() { return std::chrono::steady_clock::now(); };
    uint32_t rows_lo = 0;   // ... or rows [rows_lo, rows_here), once the prepared band has served the rows below rows_lo

    // ---- firstPhase on the PREPARED band (BandPack, hmk_internal.h) --------------------------------------------------------
    // The loop above reads whole adjacency rows: 2,560 entries per step at 10^6 sequences, of which it needs little.  What a
    // step needs, by where a neighbour lies (R = the band's row limit, a little above 2 * maxClusters):
    //   * :93, the best later singleton: among the neighbours x < id < R (the near row's leading section: their states change in
    //     this loop) and the FAR neighbours, whose state changes only by being absorbed -- the device lists each row's best few
    //     far candidates in the reference's order, and the first one still free is the answer;
    //   * :92, the clusters whose EVERY member is a neighbour (ClinkageClusterScorer.java:36-48): kept INCREMENTALLY instead of
    //     being recounted.  feas[x] lists the clusters that have been feasible for the later band row x, each with its minimum score
    //     and `covered`, the number of its members known to be neighbours of x; an entry is feasible iff covered == the cluster's
    //     member count.  Seeding {k, B} (:99-101, :108-110) creates entries for the later rows that have BOTH as neighbours (k's
    //     near row, stamped, against B's band neighbours: B's own near row, or the transposed list the device sent for a far B);
    //     k joining c (:97, :104) counts one more covered member in the entries of c's later rows that have k as a neighbour and
    //     leaves the others behind for good (complete linkage is monotone).  Scores are symmetric (checked by the caller).
    // A step costs the near row's leading section plus a few short lists: a few hundred entries instead of the whole row, no
    // random access outside the band's own state.  The steps, their order and every comparison are those of the loop below;
    // rows at and beyond R (a loop that runs past the band) are served by that loop.
    if (partial && symmetric_scores && hooks->band_pack && max_clusters > 0) {
        const auto tb0 = p1_now();
        const BandPack *bp = hooks->band_pack();
        p1_rows += std::chrono::duration<double, std::milli>(p1_now() - tb0).count();
        if (bp && bp->rows > 0) {
            const uint32_t R1 = std::min<uint32_t>(bp->rows, n), FT = bp->far_t;
            // One pool of entries in CHUNKS of 8 (two cache lines): a band row's entries are the slots of its chain of chunks, walked at
            // the row's turn -- made one by one over thousands of steps, they would otherwise lie a cache miss apart each (measured at
            // 10^6 in the default order: 60 ns per entry visited, more than the near rows' scans) --; a cluster's entries are a singly
            // linked list through the slots (walked when a member joins; dead entries are unlinked on the way).  No allocation per row
            // or cluster; a slot's index never changes.
            constexpr uint32_t NIL = 0xFFFFFFFFu, CHUNK = 8;
            struct FeasEnt { int32_t c, covered; uint32_t next_cl, x_mn; };   // x_mn: row << 8 | minimum score - threshold
            std::vector<FeasEnt> pool;
            std::vector<uint32_t> chunk_next;                         // per chunk: the row's next chunk
            pool.reserve((size_t)R1 * 16);
            chunk_next.reserve((size_t)R1 * 2);
            std::vector<uint32_t> row_first(R1, NIL), row_last(R1, NIL), cl_head;   // a row's first chunk / its last entry made
            cl_head.reserve(slots_max);
            std::vector<uint32_t> stamp(R1, 0);   // (step & 0xFFFFFF) << 8 | score(k, x) - base for the rows x of step k's leading near section
            uint32_t step = 0;                    // (steps <= rows of the band < 2^24)
            std::vector<uint32_t> fetched, picked;
            uint64_t n_far_row = 0, n_band_far = 0;   // on-demand fetches (timing output)
            double t_fetch = 0;
            // timing output: ticks and entries visited per section (row's entries, near candidates, stamps, join walk, seed walk)
            uint64_t sec_t[5] = {0, 0, 0, 0, 0}, sec_n[5] = {0, 0, 0, 0, 0}, sec_mark = 0, n_near_b = 0;
            auto sec_begin = [&]() { if (p1_timing) sec_mark = __builtin_ia32_rdtsc(); };
            auto sec_end = [&](int which, uint64_t visited) { if (p1_timing) { sec_t[which] += __builtin_ia32_rdtsc() - sec_mark; sec_n[which] += visited; } };
            auto consider = [&](Found &B, uint32_t m, int32_t s) {   // NearestClusterRunner's order over singletons (scan_row above)
                if (B.kind == NEAR_NULL || s > B.score ||
                    (s == B.score && better(s, seq_size(m), (int32_t)m, B.score, seq_size((uint32_t)B.slot), B.slot)))
                    B = Found{NEAR_REAL, (int32_t)m, s};
            };
            while (k < R1 && remaining > 0 && (int64_t)clusters.size() < max_clusters) {
                if (state[k] != ST_FREE) { k++; continue; }   // removed from initialList (:101, :110)
                step++;
                // The pack was written by the device: every line of it comes from memory, and the lists are short (no hardware stream
                // builds up).  Ask for the next row's near section now.
                if (k + 2 < R1) {
                    const uint32_t *nr = bp->near + bp->near_start[k + 1];
                    for (uint32_t l = 0, nl = std::min<uint32_t>(bp->near_up[k + 1], 48 * 16); l < nl; l += 16) __builtin_prefetch(nr + l);
                }
                const uint32_t *row = bp->near + bp->near_start[k];
                const uint32_t n_up = bp->near_up[k];
                Found A{NEAR_NULL, -1, 0};                      // :92
                sec_begin();
                uint64_t visited = 0;
                if (clusters.empty()) A = Found{NEAR_DUMMY, -1, INT_MIN};   // :138-140
                else
                    for (uint32_t ch = row_first[k]; ch != NIL; ch = chunk_next[ch]) {
                        if (chunk_next[ch] != NIL) __builtin_prefetch(&pool[(size_t)chunk_next[ch] * CHUNK]);
                        const uint32_t cnt = chunk_next[ch] != NIL ? CHUNK : row_last[k] - ch * CHUNK + 1;
                        for (uint32_t q = 0; q < cnt; q++) {
                            const FeasEnt &f = pool[(size_t)ch * CHUNK + q];
                            const int32_t mn = (int32_t)(f.x_mn & 0xFFu);
                            visited++;
                            if (f.covered == clusters[f.c].usize &&
                                (A.kind == NEAR_NULL || better(mn, clusters[f.c].size, clusters[f.c].id, A.score, clusters[A.slot].size, clusters[A.slot].id)))
                                A = Found{NEAR_REAL, f.c, mn};
                        }
                    }
                sec_end(0, visited);
                sec_begin();
                Found B{NEAR_NULL, -1, 0};                      // :93
                const uint32_t *ft = bp->far_top + (size_t)k * FT;
                if (remaining - 1 == 0) B = Found{NEAR_DUMMY, -1, INT_MIN};
                else {
                    // the near neighbours above k: the device listed the best few in the reference's order -- the first one still free is
                    // the best; the whole leading section only when every listed one has been taken and there are more
                    bool scan = true;
                    if (bp->near_top) {
                        const uint32_t *nt = bp->near_top + (size_t)k * BandPack::NEAR_T;
                        uint32_t t = 0;
                        for (; t < BandPack::NEAR_T && nt[t] != ~0u; t++)
                            if (state[nt[t] >> 8] == ST_FREE) { consider(B, nt[t] >> 8, (int32_t)(nt[t] & 0xFFu)); break; }
                        scan = t == BandPack::NEAR_T && n_up > BandPack::NEAR_T;   // (ran off the list's end without a free one)
                    }
                    // (whether a neighbour is still free is a coin toss: no branch on it -- a taken one only where a candidate can
                    // still win, which is rare after the first few)
                    int32_t bar = 0;
                    for (uint32_t q = 0; scan && q < n_up; q++) {
                        const uint32_t m = row[q] >> 8;
                        const int32_t s = (int32_t)(row[q] & 0xFFu), s_free = state[m] == ST_FREE ? s : -1;
                        if (s_free >= bar) { consider(B, m, s); bar = B.score; }
                    }
                    visited = scan ? n_up : 0;
                    uint32_t t = 0;
                    bool far_found = false;
                    for (; t < FT && ft[t] != ~0u; t++)
                        if (state[ft[t] >> 8] == ST_FREE) { consider(B, ft[t] >> 8, (int32_t)(ft[t] & 0xFFu)); far_found = true; break; }
                    if (!far_found && t == FT && bp->far_more[k]) {   // every listed candidate has been absorbed and the row has more: ask for them
                        const auto tf = p1_now();
                        if (!hooks->band_far || !hooks->band_far(k, fetched)) return HMK_INTERNAL_ROWS_FAILED;
                        n_band_far++;
                        t_fetch += std::chrono::duration<double, std::milli>(p1_now() - tf).count();
                        for (uint32_t e : fetched)
                            if (state[e >> 8] == ST_FREE) consider(B, e >> 8, (int32_t)(e & 0xFFu));
                    }
                }
                sec_end(1, visited);
                bool absorb = false;
                int32_t joined = -1;
                if (A.kind != NEAR_NULL) {                          // :94
                    if (B.kind != NEAR_NULL) {                      // :95
                        if (A.score >= B.score) {                   // :96
                            if (A.kind == NEAR_DUMMY) { st->crash_case = 2; st->crash_index = (int32_t)index; goto crash; }
                            joined = A.slot;                        // :97
                        } else absorb = true;                       // :99-101
                    } else {
                        if (A.kind == NEAR_DUMMY) { st->crash_case = 1; st->crash_index = (int32_t)index; goto crash; }
                        joined = A.slot;                            // :104
                    }
                } else if (B.kind != NEAR_NULL) {                   // :107
                    if (B.kind == NEAR_DUMMY) { st->crash_case = 3; st->crash_index = (int32_t)index; goto crash; }
                    absorb = true;                                  // :108-110
                } else {
                    state[k] = ST_ORPHAN;                           // :112
                    orphans.push_back(k);
                }
                // (seeding) the later band rows the new cluster {k, B} is feasible for.  A far B that is one of k's first far candidates:
                // the device sent them ready (`ready`: rows with the cluster's score).  Otherwise B's band neighbours -- its own near row,
                // or its whole list fetched from the device -- are filtered against k's stamped row below.  Found and asked for HERE,
                // before the stamps are written: the list's lines come from memory (the device wrote them).
                const uint32_t *bl = nullptr;
                uint32_t bn = 0;
                bool ready = false;
                if (absorb) {
                    const uint32_t b = (uint32_t)B.slot;
                    if (b < R1) { bl = bp->near + bp->near_start[b]; bn = bp->near_start[b + 1] - bp->near_start[b]; n_near_b++; }
                    else {
                        for (uint32_t t = 0; t < BandPack::TR_PER_ROW && t < FT && !bl; t++)
                            if (ft[t] != ~0u && (ft[t] >> 8) == b) {
                                const size_t u = (size_t)BandPack::TR_PER_ROW * k + t;
                                if (bp->tr_cnt[u] == ~0u) break;            // not prepared: the whole list, below
                                bl = bp->tr + bp->tr_start[u];
                                bn = bp->tr_cnt[u];
                                ready = true;
                            }
                        if (!bl) {
                            const auto tf = p1_now();
                            if (!hooks->far_row || !hooks->far_row(b, fetched)) return HMK_INTERNAL_ROWS_FAILED;
                            n_far_row++;
                            t_fetch += std::chrono::duration<double, std::milli>(p1_now() - tf).count();
                            bl = fetched.data();
                            bn = (uint32_t)fetched.size();
                        }
                    }
                    for (uint32_t l = 0; l < bn && l < 64 * 16; l += 16) __builtin_prefetch(bl + l);
                }
                sec_begin();
                if (joined >= 0 || (absorb && !ready))   // (k's row, for the walks that ask "is k a neighbour of this row, and with what score")
                    for (uint32_t q = 0; q < n_up; q++) stamp[row[q] >> 8] = step << 8 | (row[q] & 0xFFu);
                sec_end(2, joined >= 0 || (absorb && !ready) ? n_up : 0);
                sec_begin();
                visited = 0;
                if (joined >= 0) {
                    const int32_t before = clusters[joined].usize;
                    insert_into(joined, k);
                    uint32_t *link = &cl_head[joined];
                    for (uint32_t e = *link; e != NIL; e = *link) {
                        FeasEnt &f = pool[e];
                        const uint32_t fx = f.x_mn >> 8, sw = stamp[fx];
                        visited++;
                        // its turn is over / lost earlier, for good / k is not its neighbour (lost now): out of the cluster's list
                        if (fx <= k || f.covered != before || (sw >> 8) != step) { *link = f.next_cl; continue; }
                        f.covered++;
                        f.x_mn = fx << 8 | std::min(f.x_mn & 0xFFu, sw & 0xFFu);
                        link = &f.next_cl;
                    }
                } else if (absorb) {
                    const int32_t c = (int32_t)clusters.size();
                    clusters.push_back(ClusterRec{(int32_t)k, 1, seq_size(k)});
                    cluster_of[k] = c;
                    state[k] = ST_IN_CLUSTER;
                    insert_into(c, (uint32_t)B.slot);
                    remaining--;  // initialList.remove(B)
                    cl_head.push_back(NIL);
                    // two passes: the entries of B's list that are later band rows AND neighbours of k (stamped), picked out without a
                    // branch on the data (the three tests fail two times in three, unpredictably: 10.7 ticks per entry with branches),
                    // then the few that are left
                    picked.resize(bn);
                    uint32_t np = 0;
                    if (ready) np = bn;   // (every entry is a later band row with both as neighbours, its score already the minimum)
                    else
                        for (uint32_t q = 0; q < bn; q++) {
                            const uint32_t x = bl[q] >> 8, xs = x < R1 ? x : 0;   // (row 0 is never later than k)
                            picked[np] = q;
                            np += (uint32_t)((xs > k) & ((stamp[xs] >> 8) == step));
                        }
                    for (uint32_t i = 0; i < np; i++) {
                        const uint32_t q = ready ? i : picked[i], x = bl[q] >> 8, sw = ready ? 0xFFu : stamp[x];
                        if (state[x] != ST_FREE) continue;
                        uint32_t e = row_last[x] + 1;
                        if (row_last[x] == NIL || e % CHUNK == 0) {     // the row's first entry, or its last chunk is full: a new chunk
                            const uint32_t ch = (uint32_t)chunk_next.size();
                            chunk_next.push_back(NIL);
                            pool.resize(pool.size() + CHUNK);
                            if (row_last[x] == NIL) row_first[x] = ch; else chunk_next[row_last[x] / CHUNK] = ch;
                            e = ch * CHUNK;
                        }
                        pool[e] = FeasEnt{c, 2, cl_head[c], x << 8 | std::min(sw & 0xFFu, bl[q] & 0xFFu)};
                        row_last[x] = e;
                        cl_head[c] = e;
                    }
                    visited = bn;
                }
                sec_end(joined >= 0 ? 3 : 4, visited);
                remaining--;
                index++;          // :115
                k++;
            }
            rows_lo = rows_here = k < R1 ? 0 : R1;   // (the loop below takes over at row R1, if at all)
            if (p1_timing)
                fprintf(stderr, "[hmk greedy] phase 1 on the prepared band: %.2f ms for %llu steps (%.2f ms of it waiting for the band; %llu far lists and %llu rows' far parts "
                                "fetched on demand in %.2f ms), stopped at row %u of %u\n",
                        std::chrono::duration<double, std::milli>(p1_now() - tb0).count(), (unsigned long long)step, p1_rows, (unsigned long long)n_far_row,
                        (unsigned long long)n_band_far, t_fetch, k, R1);
            if (p1_timing)
                fprintf(stderr, "[hmk greedy] phase 1 sections, Mticks / entries visited: the row's entries %.1f / %llu, near candidates %.1f / %llu, stamps %.1f / %llu, "
                                "join walks %.1f / %llu, seed walks %.1f / %llu; %zu entry slots, %llu seeds with a near B\n",
                        sec_t[0] / 1e6, (unsigned long long)sec_n[0], sec_t[1] / 1e6, (unsigned long long)sec_n[1], sec_t[2] / 1e6, (unsigned long long)sec_n[2],
                        sec_t[3] / 1e6, (unsigned long long)sec_n[3], sec_t[4] / 1e6, (unsigned long long)sec_n[4], pool.size(), (unsigned long long)n_near_b);
        }
    }
    while (k < n && remaining > 0 && (int64_t)clusters.size() < max_clusters) {
        if (state[k] != ST_FREE) { k++; continue; }  // removed from initialList (:101, :110)
        if (k >= rows_here) {                                   // row k must have landed
            const auto tw = p1_now();
            rows_here = hooks->need_rows(k, rows_lo);
            if (rows_here <= k) return HMK_INTERNAL_ROWS_FAILED;
            if (hooks->adj_base) adj = (const NbrT *)hooks->adj_base();
            p1_rows += std::chrono::duration<double, std::milli>(p1_now() - tw).count();
        }
        if (!pool_decided) {
            // Several threads pay off when a row is long: a window of 32 short rows is less work than handing it out.  Measured
            // (8 threads against 1, phase 1 in the default / generator order): 10^5 sequences (256 entries per row) 5.3 / 3.2
            // against 3.2 / 2.7 ms, 3 x 10^5 (770) 21.8 / 14.5 against 17.0 / 14.4, 5 x 10^5 (1,280) 47.6 / 35.1 against
            // 46.1 / 38.2, 10^6 (2,560) 95 / 60 against 137 / 105.
            pool_decided = true;
            const double min_row = 1600.0;
            const double avg_row = rows_here > rows_lo ? (double)(start[rows_here] - start[rows_lo]) / (double)(rows_here - rows_lo) : 0.0;
            if (T > 1 && avg_row >= min_row) spawn_pool();
            else { T = 1; if (opt.phase1_window <= 0) W = 1; }
        }
        const auto ts = p1_now();
        // ---- scan a window of positions against the current state ----
        win_lo = k;
        win_hi = (uint32_t)std::min<uint64_t>({(uint64_t)k + W, (uint64_t)rows_here, (uint64_t)n});
        win_rows.clear();
        for (uint32_t x = win_lo; x < win_hi; x++)
            if (state[x] == ST_FREE) win_rows.push_back(x);
        cursor.store(0, std::memory_order_relaxed);
        finished.store(0, std::memory_order_relaxed);
        const bool shared = T > 1 && win_rows.size() > 1;
        if (shared) {
            const uint32_t g = gen.load(std::memory_order_relaxed) + 1;
            open_gen.store(g, std::memory_order_seq_cst);
            gen.store(g, std::memory_order_release);
        }
        run_window(0);
        // (a window is tens of microseconds, so the waits spin -- but not without end: on a host with fewer free cores than
        // threads a worker may have been descheduled, and a spinning main thread only keeps it off the core longer)
        for (unsigned spins = 0; finished.load(std::memory_order_acquire) < win_rows.size();)
            if (++spins > 4000) { std::this_thread::yield(); spins = 0; }
        if (shared) {   // close the generation and let the workers that are inside leave (their cursor is exhausted)
            open_gen.store(0, std::memory_order_seq_cst);
            for (unsigned spins = 0; left.load(std::memory_order_seq_cst) != entered.load(std::memory_order_seq_cst);)
                if (++spins > 4000) { std::this_thread::yield(); spins = 0; }
        }
        const auto tc = p1_now();
        p1_scan += std::chrono::duration<double, std::milli>(tc - ts).count();
        p1_windows++;
        p1_scanned += win_rows.size();
        // ---- commit the window's steps in order ----
        for (; k < win_hi && remaining > 0 && (int64_t)clusters.size() < max_clusters; k++) {
            if (state[k] != ST_FREE) continue;
            RowScan &R = res[k - win_lo];
            if (R.dirty || R.no_clusters != clusters.empty()) p1_rescans++;
            if (R.dirty || R.no_clusters != clusters.empty()) scan_row(k, k, win_hi, scratch[0], R);   // as the sequential loop sees it now (its later neighbours inside the window stay listed)
            Found A{NEAR_NULL, -1, 0};                      // :92
            if (clusters.empty()) A = Found{NEAR_DUMMY, -1, INT_MIN};  // :138-140
            else
                for (const std::pair<int32_t, int32_t> &f : R.feas)
                    if (A.kind == NEAR_NULL || better(f.second, clusters[f.first].size, clusters[f.first].id, A.score,
                                                      clusters[A.slot].size, clusters[A.slot].id))
                        A = Found{NEAR_REAL, f.first, f.second};
            Found B = remaining - 1 == 0 ? Found{NEAR_DUMMY, -1, INT_MIN} : R.B;   // :93
            bool absorb = false;
            int32_t joined = -1;
            if (A.kind != NEAR_NULL) {                          // :94
                if (B.kind != NEAR_NULL) {                      // :95
                    if (A.score >= B.score) {                   // :96
                        if (A.kind == NEAR_DUMMY) { st->crash_case = 2; st->crash_index = (int32_t)index; goto crash; }
                        insert_into(A.slot, k);                 // :97
                        joined = A.slot;
                    } else {
                        absorb = true;                          // :99-101
                    }
                } else {
                    if (A.kind == NEAR_DUMMY) { st->crash_case = 1; st->crash_index = (int32_t)index; goto crash; }
                    insert_into(A.slot, k);                     // :104
                    joined = A.slot;
                }
            } else {
                if (B.kind != NEAR_NULL) {                      // :107
                    if (B.kind == NEAR_DUMMY) { st->crash_case = 3; st->crash_index = (int32_t)index; goto crash; }
                    absorb = true;                              // :108-110
                } else {
                    state[k] = ST_ORPHAN;                       // :112
                    orphans.push_back(k);
                }
            }
            if (absorb) {
                int32_t c = (int32_t)clusters.size();
                clusters.push_back(ClusterRec{(int32_t)k, 1, seq_size(k)});
                cluster_of[k] = c;
                state[k] = ST_IN_CLUSTER;
                insert_into(c, (uint32_t)B.slot);
                remaining--;  // initialList.remove(B)
            }
            remaining--;
            index++;          // :115
            // ---- what this step can have changed for the window's later rows ----
            // Rows that have k as a neighbour (known from k's own row: symmetric scores) are PATCHED where that is exact:
            //   * k joined cluster c: for such a row c stays feasible if it was (k is a neighbour too) and its minimum takes the
            //     pair's score in; for every other row c stops being feasible (struck below);
            //   * k seeded a cluster with B: the new cluster {k, B} is feasible for such a row iff B is its neighbour as well --
            //     known from the row's own scan when B lies inside the window (the lists of its free neighbours at window
            //     positions), else from one sequential search of the row for B's id (no state lookups: a tenth of a scan);
            //   * k became an orphan: nothing changes for them.
            // (In the reference's default order the rows of a window are each other's neighbours: scanning every such row again
            // -- what this did until the end of round 3 -- took as long as the sequential loop.)
            const bool b_in_window = absorb && (uint32_t)B.slot >= win_lo && (uint32_t)B.slot < win_hi;
            for (uint32_t x = k + 1; x < win_hi; x++) ahead_score[x - win_lo] = INT_MIN;
            for (const std::pair<uint32_t, int32_t> &a : R.ahead) ahead_score[a.first - win_lo] = a.second;
            if (absorb || joined >= 0)
                for (uint32_t x = k + 1; x < win_hi; x++) {
                    RowScan &L = res[x - win_lo];
                    if (L.dirty || state[x] != ST_FREE) continue;
                    const int32_t s_kx = ahead_score[x - win_lo];      // INT_MIN: k is not a neighbour of x
                    if (absorb) {
                        if (L.B.kind == NEAR_REAL && L.B.slot == B.slot) { L.dirty = true; continue; }   // its best candidate is gone
                        if (s_kx == INT_MIN) continue;                  // the new cluster holds a non-neighbour (k): never feasible
                        if (b_in_window) {                              // is B a neighbour of x too?  its scan knows its window neighbours
                            const std::vector<std::pair<uint32_t, int32_t>> &side = (uint32_t)B.slot > x ? L.ahead : L.behind;
                            for (const std::pair<uint32_t, int32_t> &nb : side)
                                if (nb.first == (uint32_t)B.slot) { L.feas.emplace_back((int32_t)clusters.size() - 1, std::min(s_kx, nb.second)); break; }
                        } else {                                        // ... else its row does: one sequential pass over ~10 KB, no state lookups
                            for (uint64_t q = start[x], qe = start[x + 1]; q < qe; q++)
                                if (adj[q].id() == (uint32_t)B.slot) {
                                    L.feas.emplace_back((int32_t)clusters.size() - 1, std::min(s_kx, (int32_t)adj[q].score()));
                                    break;
                                }
                        }
                    } else if (s_kx != INT_MIN) {
                        for (std::pair<int32_t, int32_t> &f : L.feas)
                            if (f.first == joined) { f.second = std::min(f.second, s_kx); break; }
                    } else {
                        for (size_t f = 0; f < L.feas.size(); f++)
                            if (L.feas[f].first == joined) { L.feas[f] = L.feas.back(); L.feas.pop_back(); break; }
                    }
                }
        }
        p1_commit += std::chrono::duration<double, std::milli>(p1_now() - tc).count();
    }
    t_phase1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (p1_timing)
        fprintf(stderr, "[hmk greedy] phase 1: %.2f ms = rows %.2f + window scans %.2f + commits %.2f; %llu windows of %u, %llu rows scanned, %llu scanned again, %u threads\n",
                t_phase1, p1_rows, p1_scan, p1_commit, (unsigned long long)p1_windows, W, (unsigned long long)p1_scanned,
                (unsigned long long)p1_rescans, T);
    st->phase1_stop_index = (int32_t)index;
    st->phase1_clusters = (int32_t)clusters.size();
    st->phase1_orphans = (int32_t)orphans.size();

    {
        // ---- cluster() second loop, :59-66 ---------------------------------
        auto need_all_rows = [&]() -> bool {
            if (n && (rows_here < n || rows_lo > 0)) {
                rows_here = hooks->need_rows(n - 1, 0);
                rows_lo = 0;
                if (hooks->adj_base) adj = (const NbrT *)hooks->adj_base();
            }
            return rows_here >= n;
        };
        std::vector<uint32_t> leftover(orphans);
        for (uint32_t q = k; q < n; q++)
            if (state[q] == ST_FREE) leftover.push_back(q);
        // Complete linkage is monotone: clusters only GROW in this loop, so a sequence whose neighbours
        // do not cover some cluster's members at the START of the loop can never join that cluster later.
        // (1) Pre-check, independent per sequence and read-only, on all host threads: the clusters that
        //     are feasible against the start-of-loop membership, with their min score (a CSR list per
        //     sequence, filled by the threads in order).
        // (2) Sequential, order-dependent part (:60-62) only for the survivors; with symmetric scores a
        //     survivor does not rescan its neighbours (see "Subscribers" below).
        using Cand = GreedyCand;   // cluster, min score so far, members joined in this loop that are neighbours
        const size_t nl = leftover.size();
        std::vector<uint32_t> cand_start(nl + 1, 0);     // CSR of candidate clusters per leftover
        std::vector<Cand> cand;
        const bool fast = !clusters.empty() && nl > 512;
        bool have_cand = false;
        bool device_done = false, prefilled = false;
        std::vector<int32_t> join_slot;
        if (fast && symmetric_scores && hooks && hooks->device_loop) {   // large inputs: the whole loop on the GPU, in optimistic rounds
            std::vector<int32_t> usize(clusters.size()), cids(clusters.size());
            std::vector<int64_t> csize(clusters.size());
            for (size_t c = 0; c < clusters.size(); c++) { usize[c] = clusters[c].usize; csize[c] = clusters[c].size; cids[c] = clusters[c].id; }
            // the result as it stands after phase 1, written NOW: the device is still scoring / building its CSR and the host has
            // nothing to do -- after the loop only the leftovers that joined are patched (:67-68 below)
            for (uint32_t q = 0; q < n; q++) cluster_id[q] = cluster_of[q] >= 0 ? clusters[cluster_of[q]].id : (int32_t)q;
            if (result_order)
                for (size_t c = 0; c < clusters.size(); c++) result_order[c] = clusters[c].id;
            prefilled = true;
            device_done = hooks->device_loop(cluster_of.data(), usize, csize, cids, leftover, join_slot);
        }
        if (fast && !device_done && hooks && hooks->precheck) {   // the adjacency is still on the GPU: pre-check there
            std::vector<int32_t> usize(clusters.size());
            for (size_t c = 0; c < clusters.size(); c++) usize[c] = clusters[c].usize;
            have_cand = hooks->precheck(cluster_of.data(), usize, leftover, cand_start, cand);
            if (!have_cand) { cand_start.assign(nl + 1, 0); cand.clear(); }
        }
        if (!device_done && !need_all_rows()) return HMK_INTERNAL_ROWS_FAILED;   // every other path below reads rows
        if (fast && !have_cand && !device_done) {
            const unsigned T = std::max(1u, std::min(16u, usable_cpus()));
            const size_t nc = clusters.size();
            std::vector<std::vector<Cand>> found(T);     // thread t covers a contiguous run of leftovers
            // most neighbours are in no cluster at all: a bitmap over the sequences (n / 8 bytes, cache resident)
            // answers that before the 4-byte-per-sequence cluster_of[] has to be touched
            std::vector<uint64_t> in_cluster(((size_t)n + 63) / 64, 0);
            for (uint32_t s2 = 0; s2 < n; s2++)
                if (cluster_of[s2] >= 0) in_cluster[s2 >> 6] |= 1ull << (s2 & 63);
            auto work = [&](unsigned t) {
                std::vector<int32_t> c2(nc, 0), m2(nc, 0);
                std::vector<int32_t> seen;
                const size_t lo = nl * t / T, hi = nl * (t + 1) / T;
                for (size_t q = lo; q < hi; q++) {
                    const uint32_t y = leftover[q];
                    seen.clear();
                    for (uint64_t e = start[y]; e < start[y + 1]; e++) {
                        const uint32_t id = adj[e].id();
                        if (!((in_cluster[id >> 6] >> (id & 63)) & 1ull)) continue;
                        const int32_t c = cluster_of[id];
                        if (c2[c]++ == 0) { seen.push_back(c); m2[c] = adj[e].score(); }
                        else if (adj[e].score() < m2[c]) m2[c] = adj[e].score();
                    }
                    uint32_t k = 0;
                    for (int32_t c : seen) {
                        if (c2[c] == clusters[c].usize) { found[t].push_back(Cand{c, m2[c], 0}); k++; }
                        c2[c] = 0;
                    }
                    cand_start[q + 1] = k;
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; t++) pool.emplace_back(work, t);
            work(0);
            for (auto &th : pool) th.join();
            for (size_t q = 0; q < nl; q++) cand_start[q + 1] += cand_start[q];
            cand.reserve(cand_start[nl]);
            for (unsigned t = 0; t < T; t++) cand.insert(cand.end(), found[t].begin(), found[t].end());
        }
        const double t_pre = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (hooks && hooks->times) { hooks->times->phase1_ms = t_phase1; hooks->times->host_precheck_ms = t_pre - t_phase1; }
        if (opt.timing)
            fprintf(stderr, "[hmk greedy] phase1 %.2f ms, pre-check %.2f ms (%zu leftovers)\n", t_phase1, t_pre - t_phase1, nl);
        // "Subscribers": per cluster, the leftovers that listed it as a candidate.  When y joins cluster c,
        // y's neighbours are stamped into a sequence-indexed scratch (one sequential pass over adj[y]) and
        // only c's later subscribers are visited: a subscriber that is a neighbour of y counts one more
        // covered member (and folds the score into its min); at its own turn a candidate is still feasible
        // iff covered == members joined since the pre-check.
        const bool use_subs = fast && symmetric_scores && !device_done;
        struct Sub { uint32_t w; int32_t k; };           // the subscriber (sequence id), index of its candidate in `cand`;
                                                         // per cluster in leftover order = increasing id
        std::vector<uint32_t> sub_start;                 // CSR of subscribers per cluster
        std::vector<Sub> subs;
        std::vector<uint32_t> sub_pos;                   // per cluster: first subscriber that may still be undecided
        std::vector<int32_t> joined;                     // members that joined each cluster in this loop
        std::vector<uint64_t> stamped;                   // sequence -> (stamp of the join that last touched it) << 32 | score,
                                                         // one word so that stamping and testing touch one cache line
        if (use_subs) {
            const size_t nc = clusters.size();
            sub_start.assign(nc + 1, 0);
            for (const Cand &cd : cand) sub_start[cd.c + 1]++;
            for (size_t c = 0; c < nc; c++) sub_start[c + 1] += sub_start[c];
            subs.resize(sub_start[nc]);
            std::vector<uint32_t> fill(sub_start.begin(), sub_start.end() - 1);
            for (size_t q = 0; q < nl; q++)
                for (uint32_t k = cand_start[q]; k < cand_start[q + 1]; k++) subs[fill[cand[k].c]++] = Sub{leftover[q], (int32_t)k};
            joined.assign(nc, 0);
            stamped.assign(n, 0);
            sub_pos.assign(sub_start.begin(), sub_start.end() - 1);
        }
        uint32_t stamp = 0;
        std::vector<uint32_t> rest;
        double t_scan = 0, t_push = 0;
        const bool timing = opt.timing;
        auto now = []() { return std::chrono::steady_clock::now(); };
        for (size_t q = 0; q < nl; q++) {
            const uint32_t y = leftover[q];
            Found F{NEAR_NULL, -1, 0};                                              // :60
            if (device_done) {   // decided on the device (k_loop_*); joins are applied in loop order
                if (join_slot[q] >= 0) { insert_into(join_slot[q], y); cluster_id[y] = clusters[join_slot[q]].id; }   // :61-62
                else rest.push_back(y);                                              // :64
                continue;
            }
            if (!use_subs) {
                auto ta = now();
                if (!fast || cand_start[q + 1] > cand_start[q]) F = nearest_cluster(y);
                if (timing) t_scan += std::chrono::duration<double, std::milli>(now() - ta).count();
            } else {
                for (uint32_t k = cand_start[q]; k < cand_start[q + 1]; k++) {
                    const Cand cd = cand[k];
                    if (cd.covered != joined[cd.c]) continue;            // some new member is not a neighbour of y
                    if (F.kind == NEAR_NULL ||
                        better(cd.mn, clusters[cd.c].size, clusters[cd.c].id, F.score, clusters[F.slot].size, clusters[F.slot].id))
                        F = Found{NEAR_REAL, cd.c, cd.mn};
                }
            }
            if (F.kind == NEAR_REAL) {
                insert_into(F.slot, y);                     // :61-62 (score >= threshold by construction)
                if (use_subs) {
                    auto tb = now();
                    joined[F.slot]++;
                    stamp++;
                    const uint64_t hi = (uint64_t)stamp << 32;
                    // only still undecided leftovers matter, and they all have ids above y (the leftover list is in
                    // increasing id order): with an "upper neighbours first" row that is its leading section
                    const uint64_t e_end = upper ? start[y] + upper[y] : start[y + 1];
                    for (uint64_t e = start[y]; e < e_end; e++)
                        stamped[adj[e].id()] = hi | (uint32_t)adj[e].score();
                    uint32_t &first = sub_pos[F.slot];       // subscribers with an id up to y are decided: skip them for good
                    const uint32_t last = sub_start[F.slot + 1];
                    while (first < last && subs[first].w <= y) first++;
                    for (uint32_t u = first; u < last; u++) {
                        const Sub sb = subs[u];
                        const uint64_t sw = stamped[sb.w];
                        if ((uint32_t)(sw >> 32) != stamp) continue; // w is not a neighbour of the new member
                        Cand &cw = cand[sb.k];               // one cache line: counter and min score together
                        cw.covered++;
                        const int32_t sc = (int32_t)(uint32_t)sw;
                        if (sc < cw.mn) cw.mn = sc;
                    }
                    if (timing) t_push += std::chrono::duration<double, std::milli>(now() - tb).count();
                }
            } else {
                rest.push_back(y);                          // :64
            }
        }
        if (opt.timing) {
            size_t surv = 0;
            for (size_t q = 0; q < nl; q++) surv += !fast || cand_start[q + 1] > cand_start[q];
            fprintf(stderr, "[hmk greedy] full scans %.2f ms, join propagation %.2f ms (%zu candidate subscriptions, subscribers=%d)\n",
                    t_scan, t_push, subs.size(), (int)use_subs);
            fprintf(stderr, "[hmk greedy] sequential part done at %.2f ms (%zu of %zu passed the pre-check)\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), surv,
                    leftover.size());
        }
        // ---- :67-68 ---------------------------------------------------------
        int32_t out = 0;
        const bool patched = prefilled && device_done;   // (ids and the clusters' part of the list were written before the loop)
        for (const ClusterRec &c : clusters) {
            if (result_order && !patched) result_order[out] = c.id;
            out++;
        }
        if (result_order && !rest.empty()) std::memcpy(result_order + out, rest.data(), rest.size() * sizeof(int32_t));   // (ids < 2^31)
        out += (int32_t)rest.size();
        if (!patched)
            for (uint32_t q = 0; q < n; q++)
                cluster_id[q] = cluster_of[q] >= 0 ? clusters[cluster_of[q]].id : (int32_t)q;
        st->n_result_clusters = out;
        st->n_multi = (int32_t)clusters.size();
    }
    st->greedy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (hooks && hooks->times) hooks->times->sequential_ms = st->greedy_ms - hooks->times->phase1_ms - hooks->times->host_precheck_ms;
    if (opt.timing) fprintf(stderr, "[hmk greedy] total %.2f ms\n", st->greedy_ms);
    return HMK_OK;

crash:
    if (err)
        *err = "the reference throws NullPointerException here (LimitedGreedySequenceClusterer.java:97/104/108): "
               "case " + std::to_string(st->crash_case) + " at index " + std::to_string(st->crash_index);
    st->greedy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return HMK_ERR_REFERENCE_WOULD_CRASH;
}

int greedy_from_csr(uint32_t n, const int32_t *sizes, const uint64_t *start, const Nbr *adj, const uint32_t *upper,
                    const GreedyHooks *hooks, bool symmetric_scores, int max_clusters, int32_t *cluster_id,
                    int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st, std::string *err, const GreedyOptions &opt) {
    return greedy_from_csr_impl<Nbr>(n, sizes, start, adj, upper, hooks, symmetric_scores, max_clusters, cluster_id,
                                     result_order, member_rank, st, err, opt);
}

int greedy_from_csr_packed(uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrPacked *adj,
                           const uint32_t *upper, const GreedyHooks *hooks, bool symmetric_scores, int max_clusters,
                           int32_t *cluster_id, int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *st,
                           std::string *err, const GreedyOptions &opt) {
    return greedy_from_csr_impl<NbrPacked>(n, sizes, start, adj, upper, hooks, symmetric_scores, max_clusters, cluster_id,
                                           result_order, member_rank, st, err, opt);
}

}  // namespace hmk
