// k_edges.hip -- what happens to the edge segments after the neighbour pass, on the device: CSR adjacency for the
// host greedy merge, the contiguous 8-byte block and the 4-byte "row blocks" of the multi-GPU exchange.
#include "hmk_device.h"

namespace hmk {

// -----------------------------------------------------------------------------
// edge list -> CSR adjacency on the device (feeds the host greedy merge)
// -----------------------------------------------------------------------------
// The neighbour kernel leaves HMK_EDGE_SHARDS segments of packed edges.  Three small
// passes turn them into start[n + 1] / adj[] (each undirected edge stored under both
// ends when the matrix is symmetric): degree count, exclusive scan, scatter.  Order
// inside a row is arbitrary (atomic cursors); the merge does not depend on it.
// score_range[0] / [1]: smallest / largest edge score (decides whether the 4-byte adjacency fits);
// score_range[2]: edges that name a sequence outside [0, n) or a self pair (not counted; the caller gives up)
__global__ void __launch_bounds__(256)
k_edge_degree(const EdgeSegs segs, uint32_t *__restrict__ deg, int symmetric, int *__restrict__ score_range, uint32_t n,
              uint32_t row_lo, uint32_t row_limit) {
    // only rows [row_lo, row_limit) are counted: [0, n) for the whole graph, [0, R) for the "band" a first phase-1 hand-over
    // needs, the range of rows a device owns in a multi-device call (deg[] is indexed by the row itself)
    const EdgeSeg sg = segs.s[blockIdx.y];
    const uint64_t cnt = min((uint64_t)*sg.count, sg.cap);
    const uint64_t *seg = sg.edges;
    int lo = INT_MAX, hi = INT_MIN;
    // wave-uniform loop: the x side of a wave's 64 consecutive edges has a handful of distinct values
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); k0 < cnt; k0 += (uint64_t)gridDim.x * 256) {
        const uint64_t k = k0 + (threadIdx.x & 63);
        bool valid = k < cnt;
        const uint64_t e = valid ? seg[k] : 0;
        if (valid && (HMK_EDGE_X(e) >= n || HMK_EDGE_M(e) >= n || HMK_EDGE_X(e) == HMK_EDGE_M(e))) {
            atomicAdd(&score_range[2], 1);
            valid = false;
        }
        // symmetric: the edge is stored under both ends
        const uint32_t ea = symmetric ? min(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_X(e);
        const uint32_t eb = symmetric ? max(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_M(e);
        const bool va = valid && ea >= row_lo && ea < row_limit;
        const WaveGroup g = wave_groups(ea, va);
        if (va && g.rank == 0) atomicAdd(&deg[ea], g.size);
        if (valid && symmetric && eb >= row_lo && eb < row_limit) atomicAdd(&deg[eb], 1u);
        if (valid) {
            const int sc = HMK_EDGE_SCORE(e);
            lo = min(lo, sc);
            hi = max(hi, sc);
        }
    }
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, __shfl_down(lo, o, 64));
        hi = max(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&score_range[0], lo);
        atomicMax(&score_range[1], hi);
    }
}

// score_range = {INT_MAX, INT_MIN, 0} on the stream (no host buffer whose lifetime an async copy would depend on)
__global__ void k_init_range(int *__restrict__ score_range) {
    if (threadIdx.x == 0) { score_range[0] = INT_MAX; score_range[1] = INT_MIN; score_range[2] = 0; }
}

// Exclusive scan of deg[n] -> start[n + 1] in three coalesced passes over tiles of 2048 counters:
// tile sums, scan of the tile sums (one block), tile-local scan + offset.  T = uint32 or uint64.
constexpr uint32_t SCAN_TILE = 2048;

__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *__restrict__ deg, const uint32_t *__restrict__ deg_b,
                                                        uint64_t *__restrict__ tile_sum, uint32_t n) {   // deg_b (may be null): a second addend per row
    __shared__ uint64_t red[4];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint64_t v = 0;
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {
        const uint32_t k = base + j * 256 + threadIdx.x;
        if (k < n) v += (uint64_t)deg[k] + (deg_b ? deg_b[k] : 0u);
    }
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one block: exclusive scan of the n_tiles tile sums in place; tile_sum[n_tiles] = grand total.
// 256 threads on purpose: this kernel also runs on the band stream WHILE the neighbour kernel saturates every CU, and a
// 1024-thread workgroup (16 waves on one CU at once) was seen waiting 308 ms for a slot there (rocprofv3 trace, 10^6).
__global__ void __launch_bounds__(256) k_scan_tile_offsets(uint64_t *__restrict__ tile_sum, uint32_t n_tiles) {
    __shared__ uint64_t part[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n_tiles + 255) / 256;
    const uint32_t lo = min(n_tiles, tid * chunk), hi = min(n_tiles, lo + chunk);
    uint64_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += tile_sum[k];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan of the 256 partial sums
        const uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    for (uint32_t k = lo; k < hi; k++) { const uint64_t d = tile_sum[k]; tile_sum[k] = run; run += d; }
    if (tid == 255) tile_sum[n_tiles] = part[255];
}

template <typename T>
__global__ void __launch_bounds__(256)
k_scan_tiles(const uint32_t *__restrict__ deg, const uint32_t *__restrict__ deg_b, const uint64_t *__restrict__ tile_off,
             T *__restrict__ start, uint32_t n, uint32_t n_tiles, const uint32_t *__restrict__ tail_word) {
    __shared__ T v[SCAN_TILE];   // the tile-local prefix is as wide as the result: 2048 degrees of up to 2^24 pass 2^32
    __shared__ T wsum[4];
    const uint32_t base = blockIdx.x * SCAN_TILE, tid = threadIdx.x;
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {  // coalesced load
        const uint32_t k = base + j * 256 + tid;
        v[j * 256 + tid] = k < n ? (T)deg[k] + (T)(deg_b ? deg_b[k] : 0u) : 0;
    }
    __syncthreads();
    T loc[SCAN_TILE / 256], sum = 0;                  // thread tid owns the 8 consecutive counters tid * 8 ..
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) { loc[j] = sum; sum += v[tid * (SCAN_TILE / 256) + j]; }
    T inc = sum;                                      // inclusive scan of the thread sums inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(inc, o, 64);
        if ((tid & 63) >= (uint32_t)o) inc += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    T woff = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) woff += wsum[w];
    const T excl = woff + inc - sum;
    __syncthreads();
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) v[tid * (SCAN_TILE / 256) + j] = excl + loc[j];
    __syncthreads();
    const uint64_t off = tile_off[blockIdx.x];
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {  // coalesced store
        const uint32_t k = base + j * 256 + tid;
        if (k < n) start[k] = (T)(off + v[j * 256 + tid]);
    }
    if (blockIdx.x == 0 && tid == 0) {
        start[n] = (T)tile_off[n_tiles];
        // tail word (pack_rows: the misfit count): a 32-bit row_start[] that wrapped makes the block unusable as well
        if (tail_word) start[n + 1] = (T)*tail_word + (T)((sizeof(T) == 4 && tile_off[n_tiles] > 0xFFFFFFFFull) ? 1 : 0);
    }
}

// tile_scratch: uint64[ceil(n / SCAN_TILE) + 1]
template <typename T>
static void launch_scan(const uint32_t *deg, T *start, uint32_t n, uint64_t *tile_scratch, const uint32_t *tail_word,
                        hipStream_t s, const uint32_t *deg_b = nullptr) {
    const uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, s, deg, deg_b, tile_scratch, n);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(256), 0, s, tile_scratch, n_tiles);
    hipLaunchKernelGGL((k_scan_tiles<T>), dim3(n_tiles), dim3(256), 0, s, deg, deg_b, tile_scratch, start, n, n_tiles, tail_word);
}

// NbrT = Nbr: {m, score}.  NbrT = NbrPacked: m << 8 | (score - base), the caller has checked the score range.
// symmetric: row r = [neighbours with id > r | neighbours with id < r]; cursor = uint32[2 n], one cursor per section:
// the upper section grows from the row's start, the lower one backwards from its end, so no per-row split point has
// to be known beforehand -- afterwards cursor[r] IS the number of upper neighbours ("up[r]").  Consumers that only
// care about later (larger-id) neighbours walk just the first section.
template <class NbrT>
__global__ void __launch_bounds__(256)
k_edge_scatter(const EdgeSegs segs, const uint64_t *__restrict__ start, uint32_t *__restrict__ cursor, NbrT *__restrict__ adj,
               int symmetric, int base, uint32_t row_lo, uint32_t row_limit) {
    // cursor: uint32[2 * row_limit]; only rows [row_lo, row_limit) are stored
    const EdgeSeg sg = segs.s[blockIdx.y];
    const uint64_t cnt = min((uint64_t)*sg.count, sg.cap);
    const uint64_t *seg = sg.edges;
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); k0 < cnt; k0 += (uint64_t)gridDim.x * 256) {
        const uint64_t k = k0 + (threadIdx.x & 63);
        const bool valid = k < cnt;
        const uint64_t e = valid ? seg[k] : 0;
        const uint32_t x = symmetric ? min(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_X(e);
        const uint32_t m = symmetric ? max(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_M(e);
        const int32_t s = HMK_EDGE_SCORE(e);
        const bool vx = valid && x >= row_lo && x < row_limit;
        const WaveGroup g = wave_groups(x, vx);   // one atomic per distinct x of the wave
        uint32_t basex = 0;
        if (vx && g.rank == 0) basex = atomicAdd(&cursor[x], g.size);
        basex = (uint32_t)__shfl((int)basex, (int)g.leader, 64);
        if (!valid) continue;
        const bool vm = symmetric && m >= row_lo && m < row_limit;
        if constexpr (sizeof(NbrT) == 4) {
            const uint32_t rel = (uint32_t)(s - base) & 0xFFu;
            if (vx) adj[start[x] + basex + g.rank] = NbrT{(m << 8) | rel};
            if (vm) adj[start[m + 1] - 1 - atomicAdd(&cursor[row_limit + m], 1u)] = NbrT{(x << 8) | rel};
        } else {
            if (vx) adj[start[x] + basex + g.rank] = NbrT{m, s};
            if (vm) adj[start[m + 1] - 1 - atomicAdd(&cursor[row_limit + m], 1u)] = NbrT{x, s};
        }
    }
}

// ---- the lower sections by bucket ----------------------------------------------------------------------------------------
// At 10^6 sequences the scatter above is bound by its random 4-byte writes: the lower section of a row receives its 1,300
// entries one at a time over the whole pass, every write is a partial line in HBM (63-71 ms for 1.28 x 10^9 edges, with or
// without atomics).  Here the (row m, entry) records are first dealt into buckets of 2^shift consecutive rows -- by a few fat
// workgroups, so that the lines they have open (one per bucket and workgroup) stay in their XCD's L2 until they are full --
// and then one workgroup per bucket places its records with LDS counters: all its writes fall into the bucket's own few MB of
// the adjacency.  Three streaming passes over the edges instead of one pass of random writes.
constexpr uint32_t LB_MAX_BUCKETS = 4096;   // LDS tables of the partition kernel: 2 x 16 KB
constexpr uint32_t LB_MAX_ROWS = 4096;      // rows of a bucket (shift <= 12)
// Edges a thread of the dealing kernel holds (all of them loads in flight at once: 4 / 8 / 16 gave 22.1 / 21.5 / 20.4 ms at 10^6); a workgroup
// deals LOADS x 1024 edges at a time.  Small graphs take 4: the 12.8 M edges of a 10^5 call are 195 chunks of 16,384 -- fewer than the grid's
// workgroups, each a 160 us chain -- but 780 of 4,096.
constexpr int LB_LOADS_LARGE = 16, LB_LOADS_SMALL = 4;

// an edge the degree pass counted (k_edge_degree: both ends in [0, n), no self pair); ~0 marks "no edge" in the unrolled loads
__device__ __forceinline__ bool lower_record_ok(uint64_t e, uint32_t n) {
    return e != ~0ull && HMK_EDGE_X(e) < n && HMK_EDGE_M(e) < n && HMK_EDGE_X(e) != HMK_EDGE_M(e);
}

__global__ void __launch_bounds__(1024)
k_lower_count(const EdgeSegs segs, uint32_t shift, uint32_t nb, uint32_t n, uint32_t row_lo, uint32_t row_hi, unsigned long long *__restrict__ bucket_cnt) {
    __shared__ uint32_t hist[LB_MAX_BUCKETS];
    for (uint32_t b = threadIdx.x; b < nb; b += 1024) hist[b] = 0;
    __syncthreads();
    for (uint32_t sgi = 0; sgi < segs.n; sgi++) {
        const EdgeSeg sg = segs.s[sgi];
        const uint64_t cnt = min((uint64_t)*sg.count, sg.cap);
        // four loads in flight per thread: with one, a workgroup moves 8 KB per memory round trip
        const uint64_t stride = (uint64_t)gridDim.x * 1024;
        for (uint64_t k = (uint64_t)blockIdx.x * 1024 + threadIdx.x; k < cnt; k += 4 * stride) {
            uint64_t e[4];
#pragma unroll
            for (int q = 0; q < 4; q++) e[q] = k + q * stride < cnt ? sg.edges[k + q * stride] : ~0ull;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (lower_record_ok(e[q], n)) {
                    const uint32_t m = max(HMK_EDGE_X(e[q]), HMK_EDGE_M(e[q]));
                    if (m >= row_lo && m < row_hi) atomicAdd(&hist[m >> shift], 1u);
                }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 1024)
        if (hist[b]) atomicAdd(&bucket_cnt[b], (unsigned long long)hist[b]);
}

// the same counts from the rows' LOWER degrees (counted by the neighbour pass itself while it wrote the edges): one workgroup per
// bucket adds up its rows' counters -- 4 MB read instead of the 10 GB edge list at 10^6
__global__ void __launch_bounds__(256)
k_lower_count_rows(const uint32_t *__restrict__ row_lower, uint32_t shift, uint32_t n, unsigned long long *__restrict__ bucket_cnt) {
    __shared__ unsigned long long part[4];
    const uint32_t row0 = blockIdx.x << shift;
    const uint32_t rows = min(1u << shift, n - row0);
    unsigned long long sum = 0;
    for (uint32_t i = threadIdx.x; i < rows; i += 256) sum += row_lower[row0 + i];
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bucket_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// bucket_off[0 .. nb] = exclusive prefix sums of bucket_cnt; bucket_fill[] = 0
__global__ void __launch_bounds__(1024)
k_lower_offsets(const unsigned long long *__restrict__ bucket_cnt, uint32_t nb, unsigned long long *__restrict__ bucket_off,
                unsigned long long *__restrict__ bucket_fill) {
    __shared__ unsigned long long part[1024];
    constexpr uint32_t PER = LB_MAX_BUCKETS / 1024;
    unsigned long long v[PER], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t b = threadIdx.x * PER + q;
        v[q] = b < nb ? bucket_cnt[b] : 0;
        sum += v[q];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const unsigned long long add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t b = threadIdx.x * PER + q;
        if (b < nb) { bucket_off[b] = run; bucket_fill[b] = 0; }
        run += v[q];
    }
    if (threadIdx.x == 1023) bucket_off[nb] = part[1023];
}

// records: row m << 32 | the packed entry (x << 8 | score - base) of m's lower section
// The same pass over the edges also writes the entries of the UPPER sections (row x = the smaller end) straight into the
// adjacency, with k_edge_scatter's wave-grouped atomics -- one read of the edge list less (10 GB at 10^6).
template <int LB_LOADS>
__global__ void __launch_bounds__(1024)
k_lower_partition(const EdgeSegs segs, uint32_t shift, uint32_t nb, uint32_t n, uint32_t row_lo, uint32_t row_hi, int base, const unsigned long long *__restrict__ bucket_off,
                  unsigned long long *__restrict__ bucket_fill, uint64_t *__restrict__ recs, const uint64_t *__restrict__ start,
                  uint32_t *__restrict__ cursor, NbrPacked *__restrict__ adj) {
    __shared__ uint32_t hist[LB_MAX_BUCKETS];
    __shared__ uint32_t first[LB_MAX_BUCKETS];   // where this chunk's records of a bucket start, relative to bucket_off (< 2^32: a bucket holds < 2^12 rows x 2^20)
    constexpr uint32_t LB_CHUNK = LB_LOADS * 1024;
    uint32_t g = blockIdx.x;                     // chunk index over the concatenated segments
    for (uint32_t sgi = 0; sgi < segs.n; sgi++) {
        const EdgeSeg sg = segs.s[sgi];
        const uint64_t cnt = min((uint64_t)*sg.count, sg.cap);
        const uint32_t chunks = (uint32_t)((cnt + LB_CHUNK - 1) / LB_CHUNK);
        while (g < chunks) {
            const uint64_t k0 = (uint64_t)g * LB_CHUNK;
            const uint32_t len = (uint32_t)min((uint64_t)LB_CHUNK, cnt - k0);
            for (uint32_t b = threadIdx.x; b < nb; b += 1024) hist[b] = 0;
            __syncthreads();
            // the chunk is read once and stays in registers for both passes (LB_LOADS edges per thread, all loads in flight)
            static_assert(LB_CHUNK == LB_LOADS * 1024, "a chunk is one round of loads");
            uint64_t ev[LB_LOADS];
#pragma unroll
            for (int q = 0; q < LB_LOADS; q++) ev[q] = threadIdx.x + q * 1024 < len ? sg.edges[k0 + threadIdx.x + q * 1024] : ~0ull;
            // (row_lo / row_hi: the rows this device owns -- a multi-device call builds every row's sections where the row lives;
            // an end outside the range is some other device's entry)
            auto lower_here = [&](uint64_t e) { const uint32_t m = max(HMK_EDGE_X(e), HMK_EDGE_M(e)); return lower_record_ok(e, n) && m >= row_lo && m < row_hi; };
            auto upper_here = [&](uint64_t e) { const uint32_t x = min(HMK_EDGE_X(e), HMK_EDGE_M(e)); return lower_record_ok(e, n) && x >= row_lo && x < row_hi; };
#pragma unroll
            for (int q = 0; q < LB_LOADS; q++)
                if (lower_here(ev[q])) atomicAdd(&hist[max(HMK_EDGE_X(ev[q]), HMK_EDGE_M(ev[q])) >> shift], 1u);
            __syncthreads();
            for (uint32_t b = threadIdx.x; b < nb; b += 1024) {
                const uint32_t h = hist[b];
                if (h) first[b] = (uint32_t)atomicAdd(&bucket_fill[b], (unsigned long long)h);
                hist[b] = 0;
            }
            __syncthreads();
            // LB_WB edges of the thread at a time: first every returning atomic (the upper sections' row cursors, one per distinct x
            // of the wave) and every start[x] load of the batch is ISSUED, then the entries are written -- one edge after the other
            // (atomic -> wait -> load -> wait -> store, 16 times) the write phase was 16 dependent memory round trips per chunk
            constexpr int LB_WB = LB_LOADS < 8 ? LB_LOADS : 8;
            static_assert(LB_LOADS % LB_WB == 0, "write batches");
#pragma unroll
            for (int q0 = 0; q0 < LB_LOADS; q0 += LB_WB) {
                uint32_t basex[LB_WB], lr[LB_WB];
                uint64_t st[LB_WB];
                {   // (whole waves: wave_groups needs all 64 lanes; a wave's edges come from a handful of rows)
#pragma unroll
                    for (int u = 0; u < LB_WB; u++) {
                        const uint64_t e = ev[q0 + u];
                        const bool ok = upper_here(e);
                        const uint32_t x = min(HMK_EDGE_X(e), HMK_EDGE_M(e));
                        const WaveGroup g = wave_groups(x, ok);   // one atomic per distinct x of the wave
                        basex[u] = 0;
                        if (ok && g.rank == 0) basex[u] = atomicAdd(&cursor[x], g.size);
                        st[u] = ok ? start[x] : 0ull;
                        lr[u] = g.leader | g.rank << 8;
                    }
#pragma unroll
                    for (int u = 0; u < LB_WB; u++) {
                        const uint64_t e = ev[q0 + u];
                        const bool ok = upper_here(e);
                        const uint32_t m = max(HMK_EDGE_X(e), HMK_EDGE_M(e));
                        const uint32_t rel = (uint32_t)(HMK_EDGE_SCORE(e) - base) & 0xFFu;
                        const uint32_t bx = (uint32_t)__shfl((int)basex[u], (int)(lr[u] & 0xFFu), 64);
                        if (ok) adj[st[u] + bx + (lr[u] >> 8)] = NbrPacked{(m << 8) | rel};
                    }
                }
#pragma unroll
                for (int u = 0; u < LB_WB; u++) {
                    const uint64_t e = ev[q0 + u];
                    if (!lower_here(e)) continue;
                    const uint32_t x = min(HMK_EDGE_X(e), HMK_EDGE_M(e)), m = max(HMK_EDGE_X(e), HMK_EDGE_M(e));
                    const uint32_t rel = (uint32_t)(HMK_EDGE_SCORE(e) - base) & 0xFFu;
                    const uint32_t b = m >> shift;
                    const uint32_t at = first[b] + atomicAdd(&hist[b], 1u);
                    recs[bucket_off[b] + at] = ((uint64_t)m << 32) | (uint64_t)((x << 8) | rel);
                }
            }
            __syncthreads();
            g += gridDim.x;
        }
        g -= chunks;
    }
}

// one workgroup per bucket: the lower section of row m is filled backwards from the row's end, as k_edge_scatter does;
// cursor[row_limit + m] receives its size
__global__ void __launch_bounds__(512)
k_lower_place(const uint64_t *__restrict__ recs, const unsigned long long *__restrict__ bucket_off, uint32_t shift, uint32_t n,
              const uint64_t *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t row_limit, NbrPacked *__restrict__ adj) {
    __shared__ uint64_t row_end[LB_MAX_ROWS];
    __shared__ uint32_t filled[LB_MAX_ROWS];
    const uint32_t row0 = blockIdx.x << shift;
    const uint32_t rows = min(1u << shift, n - row0);
    for (uint32_t i = threadIdx.x; i < rows; i += 512) { row_end[i] = start[row0 + i + 1]; filled[i] = 0; }
    __syncthreads();
    const unsigned long long k1 = bucket_off[blockIdx.x + 1];
    for (unsigned long long k = bucket_off[blockIdx.x] + threadIdx.x; k < k1; k += 4 * 512) {   // four loads in flight per thread
        uint64_t rec[4];
#pragma unroll
        for (int q = 0; q < 4; q++) rec[q] = k + q * 512 < k1 ? recs[k + q * 512] : ~0ull;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (rec[q] == ~0ull) continue;
            const uint32_t i = (uint32_t)(rec[q] >> 32) - row0;
            adj[row_end[i] - 1 - atomicAdd(&filled[i], 1u)] = NbrPacked{(uint32_t)rec[q]};
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < rows; i += 512) cursor[row_limit + row0 + i] = filled[i];
}

// The same for buckets of <= 512 rows, with full-line writes: a batch of 8,192 records is counting-sorted by row in LDS first, so a
// row's entries of the batch (16 on average) leave as one run of consecutive addresses instead of 16 single dwords at 16
// different times -- 96 buckets share an XCD's 4 MB L2, which therefore cannot collect the lines for them.
constexpr uint32_t LP_ROWS = 512, LP_BATCH = 8192;
__global__ void __launch_bounds__(512)
k_lower_place_sorted(const uint64_t *__restrict__ recs, const unsigned long long *__restrict__ bucket_off, uint32_t shift, uint32_t n,
                     const uint64_t *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t row_limit, NbrPacked *__restrict__ adj) {
    __shared__ uint64_t row_end[LP_ROWS];
    __shared__ uint32_t filled[LP_ROWS], hist[LP_ROWS], offs[LP_ROWS], scan[2][LP_ROWS];
    __shared__ uint32_t sorted[LP_BATCH];
    __shared__ uint16_t srow[LP_BATCH];
    constexpr int PER = LP_BATCH / 512;
    const uint32_t t = threadIdx.x;
    const uint32_t row0 = blockIdx.x << shift;
    const uint32_t rows = min(1u << shift, n - row0);   // <= LP_ROWS (the launcher checks the shift)
    row_end[t] = t < rows ? start[row0 + t + 1] : 0;
    filled[t] = 0;
    hist[t] = 0;
    __syncthreads();
    const unsigned long long k1 = bucket_off[blockIdx.x + 1];
    for (unsigned long long k0 = bucket_off[blockIdx.x]; k0 < k1; k0 += LP_BATCH) {
        const uint32_t len = (uint32_t)min((unsigned long long)LP_BATCH, k1 - k0);
        uint64_t rec[PER];
        uint32_t rk[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) rec[q] = t + q * 512 < len ? recs[k0 + t + q * 512] : ~0ull;
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (rec[q] != ~0ull) rk[q] = atomicAdd(&hist[(uint32_t)(rec[q] >> 32) - row0], 1u);
        __syncthreads();
        // exclusive scan of hist over the 512 rows (one per thread): inside a wave by shuffles, across the 8 waves through 8
        // LDS words -- two barriers instead of the ten of a workgroup-wide doubling scan (a batch is a dozen barrier-to-barrier
        // phases of a microsecond each, and that, not bandwidth, is what this kernel's time is made of)
        const uint32_t own = hist[t];
        uint32_t incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((int)(t & 63u) >= d) incl += v;
        }
        if ((t & 63u) == 63u) scan[0][t >> 6] = incl;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (uint32_t w = 0; w < LP_ROWS / 64; w++) before += w < (t >> 6) ? scan[0][w] : 0u;
        offs[t] = before + incl - own;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (rec[q] != ~0ull) {
                const uint32_t i = (uint32_t)(rec[q] >> 32) - row0;
                sorted[offs[i] + rk[q]] = (uint32_t)rec[q];
                srow[offs[i] + rk[q]] = (uint16_t)i;
            }
        __syncthreads();
        for (uint32_t p = t; p < len; p += 512) {
            const uint32_t i = srow[p];
            adj[row_end[i] - 1 - filled[i] - (p - offs[i])] = NbrPacked{sorted[p]};
        }
        __syncthreads();
        filled[t] += own;
        hist[t] = 0;
        __syncthreads();
    }
    if (t < rows) cursor[row_limit + row0 + t] = filled[t];
}

// The HMK_EDGE_SHARDS output segments -> one contiguous block (device to device), so a
// fixed-size collective can ship a rank's edges without any host round trip.
__global__ void __launch_bounds__(256)
k_compact_edges(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
                uint64_t *__restrict__ out, uint64_t out_capacity, unsigned long long *__restrict__ total) {
    const uint32_t shard = blockIdx.y;
    uint64_t base = 0, all = 0;
    for (uint32_t q = 0; q < HMK_EDGE_SHARDS; q++) {
        const uint64_t c = min((uint64_t)counts[q], cap_per_shard);
        if (q < shard) base += c;
        all += c;
    }
    if (blockIdx.x == 0 && shard == 0 && threadIdx.x == 0) *total = all;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < cnt; k += (uint64_t)gridDim.x * 256)
        if (base + k < out_capacity) out[base + k] = seg[k];
}

// -----------------------------------------------------------------------------
// "row blocks": the 4-byte-per-edge exchange format of the multi-GPU path
// -----------------------------------------------------------------------------
// A rank's edge segments regrouped by x: row_start[n + 2] (uint32; [n] = total, [n + 1] =
// number of edges whose score - threshold did not fit 8 bits, must be 0) and one uint32 per
// edge, m << 8 | (score - threshold).  Halves the bytes the all-gather ships over xGMI.
constexpr int ROWS_UNROLL = 4;  // independent 64-edge groups per wave iteration (memory-level parallelism)

__global__ void __launch_bounds__(256)
k_rows_degree(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
              uint32_t *__restrict__ deg, uint32_t *__restrict__ misfit, int threshold) {
    const uint32_t shard = blockIdx.y, lane = threadIdx.x & 63;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t k0 = wave * (64 * ROWS_UNROLL); k0 < cnt; k0 += n_waves * (64 * ROWS_UNROLL)) {
        uint64_t e[ROWS_UNROLL];
        bool valid[ROWS_UNROLL];
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint64_t k = k0 + u * 64 + lane;
            valid[u] = k < cnt;
            e[u] = valid[u] ? seg[k] : 0;
        }
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint32_t x = HMK_EDGE_X(e[u]);
            const WaveGroup g = wave_groups(x, valid[u]);
            if (valid[u] && g.rank == 0) atomicAdd(&deg[x], g.size);
            const int32_t rel = HMK_EDGE_SCORE(e[u]) - threshold;
            if (valid[u] && (rel < 0 || rel > 255)) atomicAdd(misfit, 1u);
        }
    }
}

__global__ void __launch_bounds__(256)
k_rows_scatter(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
               const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t *__restrict__ adj,
               uint64_t adj_capacity, int threshold) {
    const uint32_t shard = blockIdx.y, lane = threadIdx.x & 63;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t k0 = wave * (64 * ROWS_UNROLL); k0 < cnt; k0 += n_waves * (64 * ROWS_UNROLL)) {
        uint64_t e[ROWS_UNROLL];
        bool valid[ROWS_UNROLL];
        WaveGroup g[ROWS_UNROLL];
        uint32_t base[ROWS_UNROLL];
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint64_t k = k0 + u * 64 + lane;
            valid[u] = k < cnt;
            e[u] = valid[u] ? seg[k] : 0;
        }
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {  // the group leaders' atomics of all groups are in flight together
            const uint32_t x = HMK_EDGE_X(e[u]);
            g[u] = wave_groups(x, valid[u]);
            base[u] = 0;
            if (valid[u] && g[u].rank == 0) base[u] = start[x] + atomicAdd(&cursor[x], g[u].size);
        }
#pragma unroll
        for (int u = 0; u < ROWS_UNROLL; u++) {
            const uint32_t b = (uint32_t)__shfl((int)base[u], (int)g[u].leader, 64);
            if (valid[u]) {
                const uint64_t pos = (uint64_t)b + g[u].rank;
                if (pos < adj_capacity)
                    adj[pos] = (HMK_EDGE_M(e[u]) << 8) | (uint32_t)((HMK_EDGE_SCORE(e[u]) - threshold) & 0xFF);
            }
        }
    }
}

// row blocks -> packed 8-byte edges, out[start[x] + k]; one wave per row, rows strided over the grid
__global__ void __launch_bounds__(256)
k_rows_unpack(const uint32_t *__restrict__ start, const uint32_t *__restrict__ adj, uint32_t n, int threshold,
              uint64_t *__restrict__ out, uint64_t out_capacity) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (uint32_t x = wave; x < n; x += n_waves) {
        const uint32_t b = start[x], e = start[x + 1];
        for (uint32_t k = b + lane; k < e; k += 64) {
            const uint32_t a = adj[k];
            const int32_t sc = (int32_t)(a & 0xFF) + threshold;
            if (k < out_capacity) out[k] = ((uint64_t)x << 40) | ((uint64_t)(a >> 8) << 16) | (uint64_t)(uint16_t)(int16_t)sc;
        }
    }
}

// -----------------------------------------------------------------------------
// pre-check of the greedy merge's second loop, on the adjacency while it is still on the device
// -----------------------------------------------------------------------------
// For every leftover sequence y (one wave each): which clusters have ALL their members among y's neighbours, and
// with which lowest score (LimitedGreedySequenceClusterer.java:60 asks that of every cluster; complete linkage is
// monotone, so clusters that fail now can never be joined later -- DESIGN.md "Exact greedy").  The neighbours'
// clusters are counted in a per-wave LDS hash table (key = cluster, count, min score); a row with more distinct clusters
// than the table takes is walked once per class of clusters (see the kernel); only beyond 64 classes does it raise
// *overflow (the host then runs its own pre-check).
// waves (= leftovers in flight) per workgroup of the pre-check: a wave's tables are 14 KB, so 4 / 2 / 1 waves per workgroup put
// 8 / 10 / 11 waves on a CU: 14.9 / 12.8 / 12.6 ms at 10^6 (22.3 / 19.4 in the reference's default order), no difference at 10^5
constexpr int PRE_WAVES = 2;
constexpr int PRE_UNROLL = 16;   // row entries in flight per lane (see scan_row in the kernel)
constexpr int PRE_SLOTS = 1024;   // per wave; 14 KB of tables.  (PRE_SLOTS_SMALL: the first stage of the single pass, see below)
constexpr int PRE_SLOTS_SMALL = 128, PRE_SLOTS_MEDIUM = 512;

__device__ __forceinline__ uint32_t nbr_id(const Nbr &a) { return a.m; }
__device__ __forceinline__ int32_t nbr_score(const Nbr &a) { return a.s; }
__device__ __forceinline__ uint32_t nbr_id(const NbrPacked &a) { return a.v >> 8; }
__device__ __forceinline__ int32_t nbr_score(const NbrPacked &a) { return (int32_t)(a.v & 0xFFu); }

// one bit per sequence: is it in a cluster?  Most neighbours are not (5 % at 10^6), and the 4-byte cluster_of[] gather
// per neighbour (a 4 MB table, the size of an XCD's L2, next to 10 GB of streamed adjacency) is what the pre-check
// waits for; the bitmap is 32 x smaller and answers first.
__global__ void __launch_bounds__(256) k_cluster_bitmap(const int32_t *__restrict__ cluster_of, uint32_t n, uint32_t *__restrict__ bitmap) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w * 32 >= n) return;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32 && w * 32 + b < n; b++) bits |= (cluster_of[w * 32 + b] >= 0 ? 1u : 0u) << b;
    bitmap[w] = bits;
}

// MODE 0 counts the candidates of each leftover (cand_cnt), MODE 1 writes them at cand_start[q] (the prefix sums of those
// counts), MODE 2 does both in one pass: the wave takes its block of entries from a counter, writes cand_start[q] = the
// block's first entry and cand_cnt[q].  ONE counter would be one address for ~10^5..10^6 returning atomics, ~25 ns each
// (0.5 ms of the 10^5 call, 20 ms at 10^6): the buffer is cut into PRE_REGIONS regions of `capacity` entries with a counter each
// (total[PRE_REGIONS]), a workgroup uses region blockIdx % PRE_REGIONS; entries beyond a region's end are not written (the
// caller sees the counter above `capacity` and falls back to the two passes).  The table's occupied slots are kept in a
// list, so a leftover costs what its row and its few distinct clusters cost -- clearing and scanning all 1,024 slots per
// leftover was three quarters of the kernel at 10^5, where a row has 250 entries.
// SLOTS: the table size.  With 1,024 slots per wave a CU holds two workgroups, and a leftover is a chain of dependent gathers
// (row start, row, bitmap, cluster_of): the kernel is bound by latency at 8 waves per CU.  When rows have few neighbours
// inside clusters, a first stage runs with 128 slots (8 workgroups per CU); a row that overflows them goes on `retry`
// (retry_count of them) and the 1,024-slot kernel takes just those (work / work_count; null = all leftovers).
enum { PRE_COUNT = 0, PRE_FILL = 1, PRE_SINGLE = 2 };
constexpr uint32_t PRE_REGIONS = 256;

template <class NbrT, int MODE, int SLOTS>
__global__ void __launch_bounds__(64 * PRE_WAVES)
k_greedy_precheck(const uint32_t *__restrict__ work, const uint32_t *__restrict__ work_count, uint32_t *__restrict__ retry,
                  uint32_t *__restrict__ retry_count,
                  const uint64_t *__restrict__ start, const NbrT *__restrict__ adj, const int32_t *__restrict__ cluster_of,
                  const uint32_t *__restrict__ in_cluster,
                  const int32_t *__restrict__ usize, const uint32_t *__restrict__ leftover, uint32_t nl,
                  uint32_t *__restrict__ cand_cnt, uint32_t *__restrict__ cand_start, GreedyCand *__restrict__ cand,
                  uint32_t *__restrict__ overflow, unsigned long long *__restrict__ total, unsigned long long capacity,
                  uint32_t own_lo, uint32_t own_hi, uint32_t region_base, uint32_t region_count) {
    // own_lo / own_hi: only the leftovers with an id in [own_lo, own_hi) are this launch's (a multi-device call checks every
    // leftover where its adjacency row lives); region_base / region_count: the regions of cand[] this launch fills (PRE_SINGLE)
    __shared__ int32_t keys_all[PRE_WAVES * SLOTS];
    __shared__ uint32_t cnt_all[PRE_WAVES * SLOTS];
    __shared__ int32_t mn_all[PRE_WAVES * SLOTS];
    __shared__ uint16_t used_all[PRE_WAVES * SLOTS];
    __shared__ uint32_t n_used_all[PRE_WAVES];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t *keys = keys_all + wv * SLOTS;
    uint32_t *cnt = cnt_all + wv * SLOTS;
    int32_t *mn = mn_all + wv * SLOTS;
    uint16_t *used = used_all + wv * SLOTS;
    uint32_t *n_used = n_used_all + wv;
    for (uint32_t sl = lane; sl < (uint32_t)SLOTS; sl += 64) { keys[sl] = -1; cnt[sl] = 0; mn[sl] = INT_MAX; }
    if (lane == 0) *n_used = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const uint32_t n_work = work ? *work_count : nl;
    constexpr uint32_t HASH_SHIFT = SLOTS == 1024 ? 22 : SLOTS == 512 ? 23 : SLOTS == 256 ? 24 : 25;
    static_assert(SLOTS == 1024 || SLOTS == 512 || SLOTS == 256 || SLOTS == 128, "table sizes");
    for (uint32_t w = blockIdx.x * PRE_WAVES + wv; w < n_work; w += gridDim.x * PRE_WAVES) {
        const uint32_t q = work ? work[w] : w;
        const uint32_t y = leftover[q];
        if (y < own_lo || y >= own_hi) continue;   // (wave-uniform)
        const uint64_t b = start[y], e = start[y + 1];
        bool full = false;
        auto insert = [&](const NbrT nb, int32_t c) {
            uint32_t sl = ((uint32_t)c * 2654435761u) >> HASH_SHIFT;
            int probes = 0;
            for (;;) {
                const int32_t old = atomicCAS(&keys[sl], -1, c);
                if (old == -1) used[atomicAdd(n_used, 1u)] = (uint16_t)sl;   // (at most SLOTS slots can be taken)
                if (old == -1 || old == c) {
                    atomicAdd(&cnt[sl], 1u);
                    atomicMin(&mn[sl], nbr_score(nb));
                    break;
                }
                sl = (sl + 1) & (SLOTS - 1);
                if (++probes >= SLOTS) { full = true; break; }
            }
        };
        // the row's neighbours inside clusters go into the table -- all of them (parts == 1) or those whose cluster falls
        // into one of `parts` classes (c mod parts == part), for a row that touches more clusters than the table takes.
        // Sixteen entries per lane and step: the row is a chain of dependent gathers (entry -> bitmap word -> cluster_of), and
        // the full-size tables leave room for two workgroups per CU only, so the kernel waits for memory at every link unless
        // a lane has many entries in flight (10^6 sequences, 10 GB of rows: 2 / 4 / 8 / 12 / 16 / 24 entries per lane:
        // 22.0 / 18.6 / 16.3 / 15.3 / 14.9 / 14.6 ms).
        auto scan_row = [&](uint32_t part, uint32_t parts) {
            for (uint64_t k0 = b; k0 < e; k0 += 64 * PRE_UNROLL) {   // wave-uniform
                NbrT nb[PRE_UNROLL];
                bool in[PRE_UNROLL];
#pragma unroll
                for (int u = 0; u < PRE_UNROLL; u++) {
                    const uint64_t k = k0 + (uint64_t)u * 64 + lane;
                    in[u] = k < e;
                    nb[u] = in[u] ? adj[k] : adj[b];
                }
#pragma unroll
                for (int u = 0; u < PRE_UNROLL; u++) {
                    const uint32_t id = nbr_id(nb[u]);
                    in[u] = in[u] && ((in_cluster[id >> 5] >> (id & 31)) & 1u);
                }
                int32_t cs[PRE_UNROLL];
#pragma unroll
                for (int u = 0; u < PRE_UNROLL; u++) cs[u] = in[u] ? cluster_of[nbr_id(nb[u])] : -1;
#pragma unroll
                for (int u = 0; u < PRE_UNROLL; u++)
                    if (in[u] && ((uint32_t)cs[u] & (parts - 1)) == part) insert(nb[u], cs[u]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        };
        // the table's clusters of which EVERY member is a neighbour of y: counts them; with `write`, stores them from base + at
        auto harvest = [&](bool write, unsigned long long base, unsigned long long end, uint32_t at) -> uint32_t {
            const uint32_t nu = *n_used;
            uint32_t n_ok = 0;
            for (uint32_t i0 = 0; i0 < nu; i0 += 64) {
                const uint32_t i = i0 + lane;
                bool ok = false;
                uint32_t sl = 0;
                int32_t c = -1;
                if (i < nu) { sl = used[i]; c = keys[sl]; ok = (int32_t)cnt[sl] == usize[c]; }
                const uint64_t mask = __ballot(ok);
                if (write && ok) {
                    const unsigned long long pos = base + at + n_ok + mbcnt64(mask);
                    if (pos < end) cand[pos] = GreedyCand{c, mn[sl], 0};
                }
                n_ok += (uint32_t)__popcll(mask);
            }
            return n_ok;
        };
        auto clean = [&]() {   // back to an empty table
            const uint32_t nu = *n_used;
            for (uint32_t i = lane; i < nu; i += 64) { const uint32_t sl = used[i]; keys[sl] = -1; cnt[sl] = 0; mn[sl] = INT_MAX; }
            if (lane == 0) *n_used = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        };
        // where leftover q's `found` entries go (PRE_SINGLE: a block taken from the region's counter; PRE_FILL: the prefix sum)
        unsigned long long region_end = ~0ull;
        auto place = [&](uint32_t found) -> unsigned long long {
            unsigned long long base = 0;
            if (MODE == PRE_SINGLE) {
                const uint32_t region = region_base + blockIdx.x % region_count;
                if (lane == 0 && found) base = atomicAdd(&total[region], (unsigned long long)found);
                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)base);
                region_end = (unsigned long long)(region + 1) * capacity;
                base += (unsigned long long)region * capacity;
                if (lane == 0) { cand_start[q] = (uint32_t)base; cand_cnt[q] = found; }
            } else if (MODE == PRE_FILL) {
                base = cand_start[q];
            } else if (lane == 0) {
                cand_cnt[q] = found;
            }
            return base;
        };

        scan_row(0, 1);
        // (a table more than 3/4 full counts as overflowed too in the first stage: probing it is slow, and the next stage has room)
        bool overflowed = __ballot(full) != 0 || (retry && *n_used > (uint32_t)SLOTS * 3 / 4);
        if (!overflowed) {
            const uint32_t found = MODE == PRE_COUNT ? harvest(false, 0, 0, 0) : MODE == PRE_SINGLE ? harvest(false, 0, 0, 0) : 0;
            const unsigned long long base = place(found);
            if (MODE != PRE_COUNT) harvest(true, base, region_end, 0);
            clean();
            continue;
        }
        clean();
        if (retry) {                                   // the larger table's turn
            if (lane == 0) retry[atomicAdd(retry_count, 1u)] = q;
            continue;
        }
        // A row that touches more clusters than the table takes (the reference's default order puts 25,000 seeds next to each
        // other at 10^6, and their neighbours see over a thousand of those clusters): the clusters are taken in 2, 4, ... 64
        // classes, one table fill per class; one sweep counts, a second one writes.
        uint32_t parts = 2, found = 0;
        for (; parts <= 64; parts *= 2) {
            found = 0;
            bool fits = true;
            for (uint32_t part = 0; part < parts && fits; part++) {
                full = false;
                scan_row(part, parts);
                fits = __ballot(full) == 0;
                if (fits) found += harvest(false, 0, 0, 0);
                clean();
            }
            if (fits) break;
        }
        if (parts > 64) {                              // more than ~45,000 clusters next to one sequence: the host's turn
            if (lane == 0) { atomicAdd(overflow, 1u); if (MODE == PRE_SINGLE) { cand_start[q] = 0; cand_cnt[q] = 0; } }
            continue;
        }
        const unsigned long long base = place(found);
        if (MODE != PRE_COUNT) {
            uint32_t at = 0;
            for (uint32_t part = 0; part < parts; part++) {
                full = false;
                scan_row(part, parts);
                at += harvest(true, base, region_end, at);
                clean();
            }
        }
    }
}

// -----------------------------------------------------------------------------
// the second loop itself on the device, in optimistic rounds
// -----------------------------------------------------------------------------
// LimitedGreedySequenceClusterer.java:59-66 is sequential: leftover w joins the best cluster that is feasible given
// every join made by the leftovers before it.  Complete linkage is monotone in this loop -- clusters only grow, so a
// cluster that is infeasible for w now is infeasible for good -- and that makes the loop parallel without changing
// its result.  State per (leftover, candidate cluster) entry: covered = members that joined the cluster in this loop
// and are neighbours of the leftover; the entry is feasible iff covered == joined[cluster].  A leftover is OPEN while it
// is undecided; its pick = the best of its feasible entries by (score, Cluster.size(), smaller id)
// (ClinkageSequenceClusterer.java:163-173,275-289); without a feasible entry it never joins anything (final).
//   first[c] = the earliest open leftover for which c is feasible.
//   An open leftover q is ACCEPTED (its pick is final: it is what the sequential loop does at q's turn) iff first[c] == q
//   for its pick and for every feasible cluster that ties with the pick on score: no earlier open leftover can still join
//   any of those, and they are the only clusters whose change before q's turn could alter q's pick -- a candidate with a
//   strictly lower score can never overtake (scores only fall as members join, size and id only break ties), and one that
//   becomes infeasible was not the pick anyway.  (Joins already made by LATER leftovers only touched clusters that were
//   infeasible for q then, hence now.)  The earliest open leftover with a feasible entry always passes.
// One round:
//   eval    re-picks the leftovers whose entries changed in the previous round (all of them in the first round)
//   first   per cluster: first[c] by moving a cursor along the cluster's subscriber list, which is sorted by leftover --
//           open -> decided and feasible -> infeasible are both one-way, so the cursor only moves forward and the whole
//           loop walks every list once
//   accept  per cluster: the leftover at first[c], if c is its pick and the rule above holds; several first/accept passes
//           run per round, because an accepted leftover stops blocking the other clusters it lists.  A cluster that took a
//           joiner in this round (taken[c] == stamp) is not what the entries describe until `apply` has run: whoever
//           picks it or ties with it waits.  At most one join per cluster and round.
//   apply   every accepted join is pushed along the joiner's row: each later open subscriber of the cluster that is a
//           neighbour counts one more covered member and folds the pair's score into its minimum; then joined[c] and
//           Cluster.size() advance, which turns the non-neighbours' entries infeasible.  Every subscriber whose entry was
//           feasible goes on the next round's eval list.
// Rounds repeat until one accepts nobody.  The host applies the joins in leftover order afterwards.
enum : uint8_t { LS_UNDECIDED = 0, LS_NEVER = 1, LS_JOINED = 2 };
constexpr uint32_t LS_SINGLE = 0x80000000u, LS_ENTRY = 0x7FFFFFFFu;   // a subscriber record's low word: flag | candidate entry index

// subscriber lists: per cluster the (leftover, candidate entry) pairs that list it, as uint64 = leftover << 32 | entry, bit 31 of
// the entry index set when it is the leftover's ONLY candidate entry (k_loop_apply's chains; the driver keeps the entry count
// below 2^31);
// pass 0 counts, pass 1 fills (in arbitrary order: k_loop_sort_subs sorts them).  Leftover q's entries are
// cand[cand_start[q] .. cand_start[q] + cand_cnt[q]) -- the blocks need not be in leftover order (single-pass pre-check).
__global__ void __launch_bounds__(256)
k_loop_subscribers(uint32_t nl, const uint32_t *__restrict__ cand_start, const uint32_t *__restrict__ cand_cnt,
                   const GreedyCand *__restrict__ cand, uint32_t *__restrict__ cursor, const uint32_t *__restrict__ sub_start,
                   unsigned long long *__restrict__ subs, int fill) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nl) return;
    for (uint32_t k = cand_start[q], ke = k + cand_cnt[q]; k < ke; k++) {
        const uint32_t pos = atomicAdd(&cursor[cand[k].c], 1u);
        if (fill) subs[(size_t)sub_start[cand[k].c] + pos] = ((unsigned long long)q << 32) | k | (cand_cnt[q] == 1u ? LS_SINGLE : 0u);
    }
}

// Sorts every cluster's subscriber list by leftover (the high word).  One workgroup per cluster: bitonic sort of runs of SORT_RUN
// entries in LDS, then -- for the rare longer list -- merges of neighbouring runs through `tmp` (every element finds its
// place by a binary search in the other run; keys are unique).
constexpr uint32_t SORT_RUN = 4096;

__global__ void __launch_bounds__(256)
k_loop_sort_subs(const uint32_t *__restrict__ sub_start, unsigned long long *subs, unsigned long long *tmp) {
    __shared__ unsigned long long run[SORT_RUN];
    const uint32_t b = sub_start[blockIdx.x], n = sub_start[blockIdx.x + 1] - b;
    if (n < 2) return;
    unsigned long long *list = subs + b, *other = tmp + b;
    for (uint32_t r0 = 0; r0 < n; r0 += SORT_RUN) {
        const uint32_t len = min(SORT_RUN, n - r0);
        uint32_t p2 = 2;
        while (p2 < len) p2 <<= 1;
        for (uint32_t e = threadIdx.x; e < p2; e += 256) run[e] = e < len ? list[r0 + e] : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= p2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t e = threadIdx.x; e < p2; e += 256) {
                    const uint32_t x = e ^ j;
                    if (x > e) {
                        const unsigned long long a = run[e], c = run[x];
                        if ((a > c) == ((e & k) == 0)) { run[e] = c; run[x] = a; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t e = threadIdx.x; e < len; e += 256) list[r0 + e] = run[e];
        __syncthreads();
    }
    if (n <= SORT_RUN) return;
    __threadfence_block();
    unsigned long long *src = list, *dst = other;
    for (uint32_t w = SORT_RUN; w < n; w <<= 1) {
        for (uint32_t e = threadIdx.x; e < n; e += 256) {
            const uint32_t pair0 = e / (2 * w) * (2 * w), mid = min(pair0 + w, n), end = min(pair0 + 2 * w, n);
            const unsigned long long v = src[e];
            uint32_t lo, hi;                         // number of elements of the OTHER run that come before v
            if (e < mid) { lo = mid; hi = end; } else { lo = pair0; hi = mid; }
            const uint32_t base = lo;
            while (lo < hi) {
                const uint32_t m = (lo + hi) >> 1;
                if (src[m] < v) lo = m + 1; else hi = m;
            }
            const uint32_t before = lo - base;
            dst[pair0 + (e < mid ? e - pair0 : e - mid) + before] = v;
        }
        __threadfence_block();
        __syncthreads();
        unsigned long long *t = src; src = dst; dst = t;
    }
    if (src != list) {
        for (uint32_t e = threadIdx.x; e < n; e += 256) list[e] = src[e];
    }
}

// per-cluster state of the loop in one 16-byte record (one load instead of three gathers per candidate entry)
struct __attribute__((aligned(16))) LoopCluster {
    int32_t joined;     // members that joined in this loop
    int32_t id;         // Cluster.getId()
    long long size;     // Cluster.size()
};

// counters (device uint32[8]): 1 = joins accepted in this round, 3 = the same, as the host polls it (written by apply),
// 4 + w = length of eval list w
//
// A leftover's candidate list is walked by 8 lanes (entry k by lane k % 8): one lane per leftover makes the kernel as slow as
// the longest list -- two dependent gathers (entry, then cluster record) of ~1 us per entry.  Round 2 went back to one lane per
// leftover as soon as the eval list was longer than the grid's 8-lane groups take in one go (32,768); several passes of short
// chains beat one pass of the longest chain far beyond that: allowing 1 / 4 / 16 / 64 passes, the 10^6 loop takes 27.0 / 26.0 /
// 23.8 / 23.5 ms (44 / 36 / 35 ms in the reference's default order), no difference at 10^5 and 3 x 10^5.
constexpr uint32_t LOOP_GRID = 1024;
constexpr uint32_t LOOP_EVAL_PASSES = 32;
__device__ __forceinline__ void
loop_eval(uint32_t block, uint32_t n_blocks, const uint32_t *__restrict__ list, uint32_t which, const uint32_t *__restrict__ cand_start,
          const uint32_t *__restrict__ cand_cnt, const GreedyCand *__restrict__ cand, const LoopCluster *__restrict__ cl, uint8_t *__restrict__ status,
          uint32_t *__restrict__ choice, uint32_t *__restrict__ dirty, uint32_t *__restrict__ counters) {
    const uint32_t t = block * 256 + threadIdx.x;
    if (t == 0) { counters[1] = 0; counters[4 + (which ^ 1u)] = 0; }   // accepted joins of this round; the next eval list starts empty
    const uint32_t n_list = counters[4 + which];
    // lanes per leftover: 8 while the list is at most LOOP_EVAL_PASSES times what the grid's 8-lane groups take in one go
    // (a second and third pass over short chains beat one pass whose time is the longest chain), else 1
    const uint32_t lw = n_list > LOOP_EVAL_PASSES * (n_blocks * 256u >> 3) ? 0u : 3u, W = 1u << lw;
    const uint32_t sub = t & (W - 1), groups = n_blocks * 256 >> lw;
    for (uint32_t i = t >> lw; i < n_list; i += groups) {   // (the W lanes of a leftover stay together)
        const uint32_t q = list[i];
        const uint32_t kb = cand_start[q], ke = kb + cand_cnt[q];
        int has = 0, b_mn = 0, b_id = 0;
        long long b_size = 0;
        uint32_t b_k = 0;
        for (uint32_t k = kb + sub; k < ke; k += W) {
            const GreedyCand cd = cand[k];
            const LoopCluster c = cl[cd.c];
            if (cd.covered != c.joined) continue;          // some new member is not a neighbour: infeasible for good
            if (!has || cd.mn > b_mn || (cd.mn == b_mn && (c.size > b_size || (c.size == b_size && c.id < b_id)))) {
                has = 1; b_mn = cd.mn; b_size = c.size; b_id = c.id; b_k = k;
            }
        }
        // the best of the W partial picks: (score, Cluster.size(), smaller id) is a total order over clusters, so the order
        // in which entries are compared does not matter
        for (uint32_t m = 1; m < W; m <<= 1) {
            const int o_has = __shfl_xor(has, m), o_mn = __shfl_xor(b_mn, m), o_id = __shfl_xor(b_id, m);
            const long long o_size = ((long long)__shfl_xor((int)(b_size >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)b_size, m);
            const uint32_t o_k = (uint32_t)__shfl_xor((int)b_k, m);
            if (o_has && (!has || o_mn > b_mn || (o_mn == b_mn && (o_size > b_size || (o_size == b_size && o_id < b_id))))) {
                has = 1; b_mn = o_mn; b_size = o_size; b_id = o_id; b_k = o_k;
            }
        }
        if (sub != 0) continue;
        dirty[q] = 0;
        if (status[q] != LS_UNDECIDED) continue;           // (it joined in the round that put it on the list)
        if (!has) status[q] = LS_NEVER;                    // :64, whatever happens later
        else choice[q] = b_k;
    }
}

// first[c]: one wave per cluster
__device__ __forceinline__ void
loop_first(uint32_t block, uint32_t n_clusters, const uint32_t *__restrict__ sub_start, const unsigned long long *__restrict__ subs,
           uint32_t *__restrict__ cursor, const GreedyCand *__restrict__ cand, const LoopCluster *__restrict__ cl,
           const uint8_t *__restrict__ status, const uint32_t *__restrict__ taken, uint32_t stamp, uint32_t *__restrict__ first) {
    const uint32_t c = block * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= n_clusters) return;
    // a cluster that took a joiner in this round is closed until `apply` has run (k_loop_accept tests taken[] before
    // first[]); its cursor must stay on the joiner, which is where `apply` starts from
    if (taken[c] == stamp) return;
    const uint32_t end = sub_start[c + 1];
    uint32_t cu = cursor[c];
    const int32_t joined = cl[c].joined;
    auto valid_at = [&](uint32_t idx, uint32_t *q) -> bool {
        const unsigned long long e = subs[idx];
        *q = (uint32_t)(e >> 32);
        return status[*q] == LS_UNDECIDED && cand[(uint32_t)e & LS_ENTRY].covered == joined;
    };
    if (cu < end) {                                       // usually the holder of the previous pass still stands
        uint32_t q = 0;
        bool ok = false;
        if (lane == 0) ok = valid_at(cu, &q);
        if (__builtin_amdgcn_readfirstlane((int)ok)) {
            if (lane == 0) first[c] = q;
            return;
        }
        cu++;
    }
    for (; cu < end; cu += 64) {                          // wave-uniform
        uint32_t q = 0;
        const bool ok = cu + lane < end && valid_at(cu + lane, &q);
        const uint64_t found = __ballot(ok);
        if (found) {
            const uint32_t at = (uint32_t)__ffsll((long long)found) - 1;
            if (lane == at) { first[c] = q; cursor[c] = cu + at; }
            return;
        }
    }
    if (lane == 0) { first[c] = 0xFFFFFFFFu; cursor[c] = end; }
}

// The round's eval and its first `first` pass in one launch (blocks [0, eval_blocks) evaluate): `first` looks at statuses and
// at entries' covered counts, not at picks, and a leftover that eval is just marking LS_NEVER has no feasible entry, so it is
// not a valid holder whichever status `first` reads.
__global__ void __launch_bounds__(256)
k_loop_eval_first(uint32_t eval_blocks, const uint32_t *__restrict__ list, uint32_t which, const uint32_t *__restrict__ cand_start,
                  const uint32_t *__restrict__ cand_cnt, const GreedyCand *__restrict__ cand, const LoopCluster *__restrict__ cl, uint8_t *__restrict__ status,
                  uint32_t *__restrict__ choice, uint32_t *__restrict__ dirty, uint32_t *__restrict__ counters,
                  uint32_t n_clusters, const uint32_t *__restrict__ sub_start, const unsigned long long *__restrict__ subs,
                  uint32_t *__restrict__ cursor, const uint32_t *__restrict__ taken, uint32_t stamp, uint32_t *__restrict__ first) {
    if (blockIdx.x < eval_blocks) loop_eval(blockIdx.x, eval_blocks, list, which, cand_start, cand_cnt, cand, cl, status, choice, dirty, counters);
    else loop_first(blockIdx.x - eval_blocks, n_clusters, sub_start, subs, cursor, cand, cl, status, taken, stamp, first);
}

__global__ void __launch_bounds__(256)
k_loop_first(uint32_t n_clusters, const uint32_t *__restrict__ sub_start, const unsigned long long *__restrict__ subs,
             uint32_t *__restrict__ cursor, const GreedyCand *__restrict__ cand, const LoopCluster *__restrict__ cl,
             const uint8_t *__restrict__ status, const uint32_t *__restrict__ taken, uint32_t stamp, uint32_t *__restrict__ first) {
    loop_first(blockIdx.x, n_clusters, sub_start, subs, cursor, cand, cl, status, taken, stamp, first);
}

// ACCEPT_LANES lanes per cluster: the leftover at first[c], if c is its pick
constexpr uint32_t ACCEPT_LANES = 8;
__global__ void __launch_bounds__(256)
k_loop_accept(uint32_t n_clusters, const uint32_t *__restrict__ cand_start, const uint32_t *__restrict__ cand_cnt,
              const GreedyCand *__restrict__ cand,
              const LoopCluster *__restrict__ cl, uint8_t *__restrict__ status, const uint32_t *__restrict__ choice,
              const uint32_t *__restrict__ first, uint32_t *__restrict__ taken, uint32_t stamp,
              uint32_t *__restrict__ accepted, int32_t *__restrict__ join_slot, uint32_t *__restrict__ counters) {
    __shared__ uint32_t n_here, base_here;
    // ACCEPT_LANES lanes per cluster: they share the walk over the leftover's candidate entries (two dependent gathers per entry;
    // with one thread per cluster the longest list of the round set the kernel's time)
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) / ACCEPT_LANES, part = threadIdx.x % ACCEPT_LANES;
    if (threadIdx.x == 0) n_here = 0;
    __syncthreads();
    uint32_t q = 0xFFFFFFFFu;
    bool accept = false;
    GreedyCand pick{0, 0, 0};
    if (c < n_clusters) {
        q = first[c];
        if (q != 0xFFFFFFFFu && taken[c] != stamp) {
            pick = cand[choice[q]];
            accept = (uint32_t)pick.c == c;                // (else q picks another cluster: that cluster's lanes look at it)
        }
    }
    bool blocked = false;                                  // a tie that may still grow
    if (accept)
        for (uint32_t k = cand_start[q] + part, ke = cand_start[q] + cand_cnt[q]; k < ke; k += ACCEPT_LANES) {
            const GreedyCand cd = cand[k];
            if (cd.mn == pick.mn && cd.covered == cl[cd.c].joined && (first[cd.c] != q || taken[cd.c] == stamp)) { blocked = true; break; }
        }
    // (the cluster's lanes sit side by side in one wave: fold their verdicts)
    {
        const uint64_t bl = __ballot(blocked);
        const uint32_t lane = threadIdx.x & 63u, g0 = lane & ~(uint32_t)(ACCEPT_LANES - 1);
        if ((bl >> g0) & ((1ull << ACCEPT_LANES) - 1ull)) accept = false;
    }
    accept = accept && part == 0;                          // one lane speaks for the cluster
    // the round's list of joins: one global atomic per workgroup (one per join was up to 4,000 returning atomics on one address)
    uint32_t at = 0;
    if (accept) at = atomicAdd(&n_here, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && n_here) base_here = atomicAdd(&counters[1], n_here);
    __syncthreads();
    if (accept) {
        status[q] = LS_JOINED;                             // :61-62
        join_slot[q] = pick.c;
        taken[c] = stamp;
        accepted[base_here + at] = q;
    }
}

// One workgroup per accepted join (y -> c): y's later neighbours go into an LDS hash table, chunk by chunk, and the
// cluster's subscribers after y probe it -- a subscriber that is a neighbour counts one more covered member and folds
// the pair's score into its minimum; the others' entries turn infeasible when joined[c] advances.
constexpr int APPLY_SUBS = 8;    // subscribers a thread of k_loop_apply has in flight
constexpr int APPLY_ROW = 8;     // row entries a thread of k_loop_apply loads before it inserts them
constexpr int APPLY_GRID = 2048; // workgroups of k_loop_apply (it loops over the round's joins)
constexpr int APPLY_SLOTS = 8192, APPLY_CHUNK = APPLY_SLOTS / 2;   // 32 KB (4-byte entries) / 64 KB of LDS, load factor <= 1/2 (4,096 slots and 1,280 workgroups: no faster)
constexpr int APPLY_SHIFT = 19;  // top 13 bits of the hash

template <class NbrT>
__global__ void __launch_bounds__(256)
k_loop_apply(const RowPieces rows,
             const uint32_t *__restrict__ leftover, uint8_t *status, GreedyCand *__restrict__ cand,
             const uint32_t *__restrict__ choice, const uint32_t *__restrict__ accepted, const uint32_t *__restrict__ sub_start,
             const unsigned long long *__restrict__ subs, uint32_t *cursor, LoopCluster *__restrict__ cl,
             const int32_t *__restrict__ seq_size, uint32_t *__restrict__ dirty, uint32_t *__restrict__ next_list, uint32_t which,
             uint32_t *__restrict__ counters, unsigned long long *host_word, uint32_t stamp,
             const uint32_t *__restrict__ cand_cnt, int32_t *join_slot, int chain) {
    // Packed adjacency (4-byte entries id << 8 | score): the table holds the entry itself -- 32 KB instead of 64, four workgroups
    // per CU instead of two, and a join is bound by its chain of dependent gathers (37 us on average at 10^6: 9 table build, 19
    // subscribers, 9 the loads before them), so the joins in flight are what counts.  Empty slot = 0: an entry of an UPPER section
    // has an id >= 1.  8-byte entries keep separate key and value arrays (empty = 0xFFFFFFFF).
    constexpr bool PACKED = sizeof(NbrT) == 4;
    constexpr uint32_t EMPTY = PACKED ? 0u : 0xFFFFFFFFu;
    __shared__ uint32_t keys[APPLY_SLOTS];
    __shared__ int32_t vals[PACKED ? 1 : APPLY_SLOTS];
    __shared__ uint32_t list_count, list_base;
    __shared__ uint32_t next_idx;   // the earliest later subscriber that stays feasible after this join (chains, below)
    const uint32_t n_acc = counters[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {                     // what the host polls: a round without a join is the end
        counters[3] = n_acc;
        // (n_acc == 0: this kernel changes nothing, so the word may go out before it ends; otherwise the host only uses the
        // round number to bound how far ahead it enqueues)
        if (host_word) __hip_atomic_store(host_word, ((unsigned long long)stamp << 32) | n_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (uint32_t a = blockIdx.x; a < n_acc; a += gridDim.x) {   // workgroup-uniform loop
        uint32_t q = accepted[a];
        const int32_t c = cand[choice[q]].c;
        int32_t joined = cl[c].joined;
        uint32_t sb = cursor[c] + 1;                             // the list is sorted and the cursor stands at q itself
        const uint32_t se = sub_start[c + 1];
        long long size_add = 0;
        uint32_t links = 0;
        // CHAINS.  After y has joined, the cluster's next holder is the earliest later subscriber that is still open and stays
        // feasible (it is a neighbour of y) -- in the round scheme it is found by the next round's `first`, picked by `eval` and
        // let in by `accept`: three launches per member of a family that joins one cluster (the reference's antibodies example:
        // 151 rounds).  If that subscriber has ONE candidate entry in all (this cluster's: every other cluster was infeasible for
        // it at the pre-check, for good), its pick is this cluster whatever else happens, no other cluster's list holds it, no
        // other workgroup ever reads or writes its status or its entry -- so it joins here and now, exactly as the sequential
        // loop has it join at its turn (LimitedGreedySequenceClusterer.java:59-66: every earlier open subscriber of the cluster
        // was decided or infeasible, which is what "earliest that stays feasible" says), and the walk goes on from it.  A
        // subscriber with several candidates ends the chain: its pick needs `eval`.
        for (;;) {
        const uint32_t y = leftover[q];
        // the joiner's row lives on the device that owns it (one piece: this device's CSR; a multi-device call reads a peer's
        // memory over xGMI -- a few KB per join)
        const uint32_t piece = rows.rows_per ? y / rows.rows_per : 0u;
        const NbrT *__restrict__ adj = (const NbrT *)rows.adj[piece];
        const uint64_t b = rows.start[piece][y], e = b + rows.up[piece][y];   // later leftovers have larger ids: the upper section
        if (threadIdx.x == 0) next_idx = 0xFFFFFFFFu;            // (read after the barriers of the chunk loop)
        bool first_chunk = true;
        for (uint64_t c0 = b; c0 < e || first_chunk; c0 += APPLY_CHUNK) {
            for (uint32_t sl = threadIdx.x; sl < (uint32_t)APPLY_SLOTS; sl += 256) keys[sl] = EMPTY;
            __syncthreads();
            const uint64_t c1 = min(e, c0 + (uint64_t)APPLY_CHUNK);
            for (uint64_t k0 = c0 + threadIdx.x; k0 < c1; k0 += 256 * APPLY_ROW) {   // APPLY_ROW entries of the row in flight per thread
                NbrT nbs[APPLY_ROW];
#pragma unroll
                for (int u = 0; u < APPLY_ROW; u++) nbs[u] = k0 + (uint64_t)u * 256 < c1 ? adj[k0 + (uint64_t)u * 256] : adj[c0];
#pragma unroll
                for (int u = 0; u < APPLY_ROW; u++) {
                    if (k0 + (uint64_t)u * 256 >= c1) continue;
                    const NbrT nb = nbs[u];
                    const uint32_t id = nbr_id(nb);
                    uint32_t sl = (id * 2654435761u) >> APPLY_SHIFT;   // top bits
                    uint32_t word = id;
                    if constexpr (PACKED) __builtin_memcpy(&word, &nb, 4);
                    for (;;) {
                        const uint32_t old = atomicCAS(&keys[sl], EMPTY, word);
                        if (old == EMPTY) { if constexpr (!PACKED) vals[sl] = nbr_score(nb); break; }   // (ids inside a row are distinct)
                        sl = (sl + 1) & (APPLY_SLOTS - 1);
                    }
                }
            }
            __syncthreads();
            // the subscribers after the cursor, APPLY_SUBS per thread and step: a subscriber is a chain of dependent gathers (list
            // entry -> status and candidate entry -> its id -> the table), and with one per thread the workgroup waits for memory
            // at every link of every 256 subscribers (popular clusters have thousands)
            for (uint32_t s0 = sb; s0 < se; s0 += 256 * APPLY_SUBS) {   // workgroup-uniform (barriers inside)
                unsigned long long sub[APPLY_SUBS];
                bool live[APPLY_SUBS];
#pragma unroll
                for (int u = 0; u < APPLY_SUBS; u++) {
                    const uint32_t s2 = s0 + (uint32_t)u * 256 + threadIdx.x;
                    live[u] = s2 < se;
                    sub[u] = live[u] ? subs[s2] : 0ull;
                }
                // only a feasible entry matters (covered == joined; joined + 1 once an earlier chunk has counted y), and it
                // changes either way -- one more covered member, or infeasible from now on: its leftover re-picks
#pragma unroll
                for (int u = 0; u < APPLY_SUBS; u++)
                    live[u] = live[u] && status[(uint32_t)(sub[u] >> 32)] == LS_UNDECIDED && cand[(uint32_t)sub[u] & LS_ENTRY].covered == joined;
                // (the ids are asked for AFTER this test on purpose: few subscribers pass it, and with the id gathered for every
                // subscriber beside its status and entry -- one link less in the chain -- the loop got slower, 19.1 -> 20.1 ms)
                uint32_t ids[APPLY_SUBS];
#pragma unroll
                for (int u = 0; u < APPLY_SUBS; u++) ids[u] = live[u] ? leftover[(uint32_t)(sub[u] >> 32)] : 0u;
                uint32_t n_marks = 0, marks = 0;                 // this lane's new entries of the next eval list (bit u)
                uint32_t best_next = 0xFFFFFFFFu;                // chains: (list index << 1 | "has other candidates") of this lane's earliest subscriber that stays feasible
#pragma unroll
                for (int u = 0; u < APPLY_SUBS; u++) {
                    const uint32_t q2 = (uint32_t)(sub[u] >> 32), k2 = (uint32_t)sub[u] & LS_ENTRY;
                    if (live[u]) {
                        const uint32_t id = ids[u];
                        uint32_t sl = (id * 2654435761u) >> APPLY_SHIFT;
                        for (;;) {
                            const uint32_t kk = keys[sl];
                            if (kk == EMPTY) break;
                            if ((PACKED ? kk >> 8 : kk) == id) { // this entry belongs to leftover q2 alone; one join per cluster and round
                                const int32_t sc = PACKED ? (int32_t)(kk & 0xFFu) : vals[sl];
                                cand[k2].covered = joined + 1;
                                if (sc < cand[k2].mn) cand[k2].mn = sc;
                                best_next = min(best_next, ((s0 + (uint32_t)u * 256 + threadIdx.x) << 1) | (((uint32_t)sub[u] & LS_SINGLE) ? 0u : 1u));
                                break;
                            }
                            sl = (sl + 1) & (APPLY_SLOTS - 1);
                        }
                        if (first_chunk && atomicExch(&dirty[q2], 1u) == 0u) { marks |= 1u << u; n_marks++; }
                    }
                }
                if (chain && __ballot(best_next != 0xFFFFFFFFu)) {   // one LDS atomic per wave that found any
#pragma unroll
                    for (int o = 32; o; o >>= 1) best_next = min(best_next, (uint32_t)__shfl_xor((int)best_next, o));
                    if ((threadIdx.x & 63) == 0) atomicMin(&next_idx, best_next);
                }
                if (first_chunk) {
                    // next round's eval list: ONE global atomic per workgroup and step.  (One per wave and subscriber slot was
                    // 4-5,000 returning atomics on a single address per round, ~25 ns each: most of this kernel's 0.14 ms.)
                    if (threadIdx.x == 0) list_count = 0;
                    __syncthreads();
                    uint32_t at = 0;
                    if (n_marks) at = atomicAdd(&list_count, n_marks);
                    __syncthreads();
                    if (threadIdx.x == 0 && list_count) list_base = atomicAdd(&counters[4 + (which ^ 1u)], list_count);
                    __syncthreads();
                    if (n_marks) {
                        uint32_t w = list_base + at;
#pragma unroll
                        for (int u = 0; u < APPLY_SUBS; u++)
                            if (marks >> u & 1u) next_list[w++] = (uint32_t)(sub[u] >> 32);
                    }
                }
            }
            __syncthreads();
            first_chunk = false;
        }
        joined += 1;
        size_add += seq_size ? (long long)seq_size[y] : 1ll;
        links++;
        const uint32_t nx = next_idx;                            // (uniform: written before the chunk loop's last barrier)
        if (nx == 0xFFFFFFFFu || (nx & 1u)) break;               // nobody stays feasible / the earliest one has other candidates: `eval` picks
        const uint32_t ni = nx >> 1;
        const uint32_t q2 = (uint32_t)(subs[ni] >> 32);
        __syncthreads();                                         // (every thread has read next_idx)
        if (threadIdx.x == 0) {
            status[q2] = LS_JOINED;                              // :61-62
            join_slot[q2] = c;
        }
        q = q2;
        sb = ni + 1;
        __threadfence_block();                                   // this link's entry updates and the status byte, for the next link's reads
        __syncthreads();
        }
        if (threadIdx.x == 0) {
            cl[c].joined = joined;
            cl[c].size += size_add;
            if (links > 1) cursor[c] = sb - 1;                   // stands at the chain's last member
        }
    }
}

// -----------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------
EdgeSegs shard_segments(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t first, uint32_t count) {
    EdgeSegs sg{};
    sg.n = count;
    for (uint32_t q = 0; q < count; q++) sg.s[q] = EdgeSeg{edges + (uint64_t)(first + q) * cap_per_shard, counts + first + q, cap_per_shard};
    return sg;
}

// Workgroups per segment of the kernels that walk an edge list restricted to rows < row_limit (the band hand-over, which runs on
// its own stream BESIDE the neighbour pass): every one of them needs a slot on a machine that is full, and 512 x 64 segments =
// 32,768 workgroups for the 1.5 x 10^6 band edges of a 10^5 call slowed the pass they ran beside from 2.95 to 3.35 ms (the whole
// "cost of the band" of DESIGN.md 4.2).  The kernels loop over their segment, so the grid is
// sized for the edges to expect -- rows x n x 0.3 % -- at ~4,096 per workgroup, 4 .. 512 (10^5: 6 per segment; 2 .. 13 measure the
// same, 32 costs the pass 0.1 ms, 128 and more 0.45 ms).
// (a multi-device root's list ends with one segment per peer that holds that peer's WHOLE block -- 64 times an ordinary segment:
// `block` asks for the grid of those; the launchers below give the two kinds a launch each)
static uint32_t band_grid_x(const EdgeSegs &segs, uint32_t n, uint32_t row_limit, bool block = false) {
    if (row_limit >= n) return 512;
    const double devices = segs.n > HMK_EDGE_SHARDS ? (double)(segs.n - HMK_EDGE_SHARDS + 1) : 1.0;
    const double per_device = (double)row_limit * (double)n * 0.003 / devices;
    const double per_seg = block ? per_device : per_device / HMK_EDGE_SHARDS;
    return (uint32_t)std::max(4.0, std::min(512.0, per_seg / 4096.0));
}
// the peers' block segments of a multi-device list as a list of their own (empty: a single-device list)
static EdgeSegs peer_blocks(const EdgeSegs &segs) {
    EdgeSegs b{};
    for (uint32_t q = HMK_EDGE_SHARDS; q < segs.n; q++) b.s[b.n++] = segs.s[q];
    return b;
}
static EdgeSegs own_segments(const EdgeSegs &segs) {
    EdgeSegs o = segs;
    o.n = std::min<uint32_t>(segs.n, HMK_EDGE_SHARDS);
    return o;
}

hipError_t launch_csr_degree_scan(const EdgeSegs &segs, uint32_t n, uint32_t row_limit, bool symmetric, uint32_t *deg,
                                  uint64_t *start, uint64_t *tile_scratch, int *score_range, hipStream_t s, uint32_t row_lo, uint32_t scan_rows) {
    if (segs.n == 0 || segs.n > HMK_MAX_SEGS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_init_range, dim3(1), dim3(64), 0, s, score_range);
    if (row_limit < n && segs.n > HMK_EDGE_SHARDS) {   // band of a multi-device root: its own segments and the peers' blocks, each at its size
        const EdgeSegs own = own_segments(segs), blocks = peer_blocks(segs);
        hipLaunchKernelGGL(k_edge_degree, dim3(band_grid_x(segs, n, row_limit), own.n), dim3(256), 0, s, own, deg, symmetric ? 1 : 0, score_range, n, row_lo, row_limit);
        hipLaunchKernelGGL(k_edge_degree, dim3(band_grid_x(segs, n, row_limit, true), blocks.n), dim3(256), 0, s, blocks, deg, symmetric ? 1 : 0, score_range, n, row_lo, row_limit);
    } else {
        hipLaunchKernelGGL(k_edge_degree, dim3(band_grid_x(segs, n, row_limit), segs.n), dim3(256), 0, s, segs, deg, symmetric ? 1 : 0, score_range, n, row_lo, row_limit);
    }
    launch_scan<uint64_t>(deg, start, scan_rows ? scan_rows : row_limit, tile_scratch, nullptr, s);   // (scan_rows: deg[] covers more rows than were counted)
    return hipGetLastError();
}

// the degrees were counted while the edges were written (NeighborParams::deg): only the scan remains
hipError_t launch_csr_scan_only(const uint32_t *deg, const uint32_t *deg_lo, uint64_t *start, uint32_t n, uint64_t *tile_scratch,
                                int *score_range, hipStream_t s) {
    hipLaunchKernelGGL(k_init_range, dim3(1), dim3(64), 0, s, score_range);
    launch_scan<uint64_t>(deg, start, n, tile_scratch, nullptr, s, deg_lo);
    return hipGetLastError();
}

hipError_t launch_csr_scatter(const EdgeSegs &segs, bool symmetric, const uint64_t *start, uint32_t *cursor, void *adj,
                              bool packed, int base, uint32_t row_limit, hipStream_t s, uint32_t n, uint32_t row_lo) {
    auto go = [&](const EdgeSegs &sg, uint32_t gx) {
        if (sg.n == 0) return;
        if (packed)
            hipLaunchKernelGGL((k_edge_scatter<NbrPacked>), dim3(gx, sg.n), dim3(256), 0, s, sg, start, cursor, (NbrPacked *)adj,
                               symmetric ? 1 : 0, base, row_lo, row_limit);
        else
            hipLaunchKernelGGL((k_edge_scatter<Nbr>), dim3(gx, sg.n), dim3(256), 0, s, sg, start, cursor, (Nbr *)adj,
                               symmetric ? 1 : 0, base, row_lo, row_limit);
    };
    if (n && row_limit < n && segs.n > HMK_EDGE_SHARDS) {   // (n given: the band's rows beside a running pass; multi-device root: two kinds of segment)
        go(own_segments(segs), band_grid_x(segs, n, row_limit));
        go(peer_blocks(segs), band_grid_x(segs, n, row_limit, true));
    } else {
        go(segs, n ? band_grid_x(segs, n, row_limit) : 512);
    }
    return hipGetLastError();
}

// The packed symmetric CSR with the lower sections dealt by bucket (k_lower_*): scratch = 3 x (LB_MAX_BUCKETS + 1) uint64,
// recs = one uint64 per edge.  n < 2^24 (edge format), so 2^shift rows per bucket with shift <= 12 always give <= 4096 buckets.
constexpr uint32_t CSR_PARTITION_GRID = 512;   // two 1,024-thread workgroups per CU (256 / 512 / 1024: 19.8 / 19.4 / 19.5 ms for the CSR at 10^6)
uint32_t csr_partition_shift(uint32_t n, int forced_shift) {
    // 512 rows per bucket (2,048 workgroups of the placing kernel at 10^6); small graphs take smaller buckets so that the placing kernel still
    // has a few hundred workgroups (10^5: 128 rows, 782 buckets -- with 512 rows its 196 workgroups placed 65,536 records each, one after the other)
    uint32_t shift = 9;
    while (shift > 6 && (n >> shift) < 512) shift--;
    if (forced_shift > 0) shift = (uint32_t)std::min(12, std::max(9, forced_shift));   // tests (HMK_CSR_BUCKET_SHIFT): the wide buckets of n > 2^21
    while (((uint64_t)n + (1u << shift) - 1) >> shift > LB_MAX_BUCKETS) shift++;
    return shift;
}
size_t csr_partition_scratch_bytes() { return 3 * ((size_t)LB_MAX_BUCKETS + 1) * sizeof(unsigned long long); }
hipError_t launch_csr_scatter_partitioned(const EdgeSegs &segs, const uint64_t *start, uint32_t *cursor, void *adj, int base, uint32_t n,
                                          uint64_t *recs, void *scratch, const uint32_t *row_lower, int forced_shift, hipStream_t s,
                                          uint32_t row_lo, uint32_t row_hi) {
    if (row_hi > n) row_hi = n;
    const uint32_t shift = csr_partition_shift(n, forced_shift);
    const uint32_t nb = (uint32_t)(((uint64_t)n + (1u << shift) - 1) >> shift);
    if (shift > 12 || nb > LB_MAX_BUCKETS) return hipErrorInvalidValue;
    unsigned long long *cnt = (unsigned long long *)scratch, *off = cnt + LB_MAX_BUCKETS + 1, *fill = off + LB_MAX_BUCKETS + 1;
    if (row_lower) {   // the pass counted the rows' lower degrees itself
        hipLaunchKernelGGL(k_lower_count_rows, dim3(nb), dim3(256), 0, s, row_lower, shift, n, cnt);
    } else {
        hipError_t e = hipMemsetAsync(cnt, 0, (size_t)nb * sizeof(unsigned long long), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_lower_count, dim3(512), dim3(1024), 0, s, segs, shift, nb, n, row_lo, row_hi, cnt);
    }
    hipLaunchKernelGGL(k_lower_offsets, dim3(1), dim3(1024), 0, s, cnt, nb, off, fill);
    // (10^6 sequences, round 2: 64 / 128 / 256 / 512 workgroups gave a CSR in 57 / 42 / 35 / 36 ms with ONE load in flight per thread;
    // the kernel is bound by memory latency: 4 loads in flight 22.1 ms, 8: 21.5, 16 with the chunk kept in registers: 19.8)
    // the upper sections in the same pass over the edges
    uint64_t held = 0;   // edges the segments can hold: which chunk size the dealing takes
    for (uint32_t q = 0; q < segs.n; q++) held += segs.s[q].cap;
    if (held >= (uint64_t)CSR_PARTITION_GRID * LB_LOADS_LARGE * 1024 * 8)
        hipLaunchKernelGGL(k_lower_partition<LB_LOADS_LARGE>, dim3(CSR_PARTITION_GRID), dim3(1024), 0, s, segs, shift, nb, n, row_lo, row_hi, base, off, fill, recs,
                           start, cursor, (NbrPacked *)adj);
    else
        hipLaunchKernelGGL(k_lower_partition<LB_LOADS_SMALL>, dim3(CSR_PARTITION_GRID), dim3(1024), 0, s, segs, shift, nb, n, row_lo, row_hi, base, off, fill, recs,
                           start, cursor, (NbrPacked *)adj);
    if ((1u << shift) <= LP_ROWS)
        hipLaunchKernelGGL(k_lower_place_sorted, dim3(nb), dim3(512), 0, s, recs, off, shift, n, start, cursor, n, (NbrPacked *)adj);
    else
        hipLaunchKernelGGL(k_lower_place, dim3(nb), dim3(512), 0, s, recs, off, shift, n, start, cursor, n, (NbrPacked *)adj);
    return hipGetLastError();
}

// Workgroups per segment of the kernels that regroup a pass's edges for the exchange: in the pipelined step they run on the
// communication stream BESIDE the next pass (band_grid_x above: every workgroup that wants a slot on a full machine costs the
// pass it runs beside).  Sized from the segment's capacity -- about twice what a pass writes -- at ~4,096 entries each.
static uint32_t pack_grid_x(uint64_t cap_per_shard) {
    return (uint32_t)std::max<uint64_t>(4, std::min<uint64_t>(128, cap_per_shard / 4096));
}

hipError_t launch_compact_edges(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts,
                                uint64_t *out, uint64_t out_capacity, unsigned long long *total, hipStream_t s) {
    hipLaunchKernelGGL(k_compact_edges, dim3(pack_grid_x(cap_per_shard), HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, out,
                       out_capacity, total);
    return hipGetLastError();
}

hipError_t launch_pack_rows(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t n,
                            int threshold, uint32_t *scratch, uint32_t *row_start, uint32_t *adj, uint64_t adj_capacity,
                            hipStream_t s) {
    // scratch: uint32 deg[n], cursor[n], misfit, pad to 8 bytes, then the scan's uint64 tile sums
    hipError_t e = hipMemsetAsync(scratch, 0, ((size_t)2 * n + 1) * 4, s);
    if (e != hipSuccess) return e;
    uint32_t *deg = scratch, *cursor = scratch + n, *misfit = scratch + 2 * (size_t)n;
    uint64_t *tile_scratch = (uint64_t *)(scratch + 2 * (size_t)n + 2);
    hipLaunchKernelGGL(k_rows_degree, dim3(pack_grid_x(cap_per_shard), HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, deg, misfit,
                       threshold);
    launch_scan<uint32_t>(deg, row_start, n, tile_scratch, misfit, s);
    hipLaunchKernelGGL(k_rows_scatter, dim3(pack_grid_x(cap_per_shard), HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, row_start,
                       cursor, adj, adj_capacity, threshold);
    return hipGetLastError();
}

size_t pack_rows_scratch_bytes(uint32_t n) { return ((size_t)2 * n + 2) * 4 + scan_scratch_bytes(n); }
size_t scan_scratch_bytes(uint32_t n) { return ((size_t)(n + SCAN_TILE - 1) / SCAN_TILE + 1) * 8; }
size_t scan_total_index(uint32_t n) { return (size_t)(n + SCAN_TILE - 1) / SCAN_TILE; }

hipError_t launch_unpack_rows(const uint32_t *row_start, const uint32_t *adj, uint32_t n, int threshold, uint64_t *out,
                              uint64_t out_capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_rows_unpack, dim3(1024), dim3(256), 0, s, row_start, adj, n, threshold, out, out_capacity);
    return hipGetLastError();
}

// counts (cand_cnt[nl]) or fills (cand at cand_start[q]) the candidate lists; adj is Nbr[] or NbrPacked[]
hipError_t launch_cluster_bitmap(const int32_t *cluster_of, uint32_t n, uint32_t *bitmap, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cluster_bitmap, dim3(((n + 31) / 32 + 255) / 256), dim3(256), 0, s, cluster_of, n, bitmap);
    return hipGetLastError();
}

hipError_t launch_greedy_precheck(int mode, bool packed, const uint64_t *start, const void *adj, const int32_t *cluster_of,
                                  const uint32_t *in_cluster, const int32_t *usize, const uint32_t *leftover, uint32_t nl, uint32_t *cand_cnt,
                                  uint32_t *cand_start, GreedyCand *cand, uint32_t *overflow, unsigned long long *total,
                                  unsigned long long capacity, uint32_t *retry, uint32_t *retry_count, int first_stage_slots, hipStream_t s,
                                  uint32_t own_lo, uint32_t own_hi, uint32_t region_base, uint32_t region_count) {
    if (nl == 0) return hipSuccess;
    if (region_count == 0) { region_base = 0; region_count = PRE_REGIONS; }
    const dim3 block(64 * PRE_WAVES);
#define HMK_PRE(T, F, SL, GRID, WORK, WCNT, RETRY, RCNT)                                                                              \
    hipLaunchKernelGGL((k_greedy_precheck<T, F, SL>), dim3(GRID), block, 0, s, WORK, WCNT, RETRY, RCNT, start, (const T *)adj, cluster_of, \
                       in_cluster, usize, leftover, nl, cand_cnt, cand_start, cand, overflow, total, capacity, own_lo, own_hi, region_base, region_count)
    const uint32_t grid_big = std::min<uint32_t>((nl + PRE_WAVES - 1) / PRE_WAVES, 256 * 12 * 4 / PRE_WAVES),
                   grid_small = std::min<uint32_t>((nl + PRE_WAVES - 1) / PRE_WAVES, 256 * 32 * 4 / PRE_WAVES);
    if (mode == PRE_SINGLE && retry) {
        // two stages: small tables for every leftover, the full-size ones for the rows that did not fit (their number is only
        // known on the device: the second launch is sized for all of them and reads the count)
        if (first_stage_slots <= PRE_SLOTS_SMALL) {
            if (packed) HMK_PRE(NbrPacked, PRE_SINGLE, PRE_SLOTS_SMALL, grid_small, nullptr, nullptr, retry, retry_count);
            else HMK_PRE(Nbr, PRE_SINGLE, PRE_SLOTS_SMALL, grid_small, nullptr, nullptr, retry, retry_count);
        } else {
            if (packed) HMK_PRE(NbrPacked, PRE_SINGLE, PRE_SLOTS_MEDIUM, grid_small, nullptr, nullptr, retry, retry_count);
            else HMK_PRE(Nbr, PRE_SINGLE, PRE_SLOTS_MEDIUM, grid_small, nullptr, nullptr, retry, retry_count);
        }
        if (packed) HMK_PRE(NbrPacked, PRE_SINGLE, PRE_SLOTS, grid_big, retry, retry_count, nullptr, nullptr);
        else HMK_PRE(Nbr, PRE_SINGLE, PRE_SLOTS, grid_big, retry, retry_count, nullptr, nullptr);
    } else if (packed) {
        if (mode == PRE_SINGLE) HMK_PRE(NbrPacked, PRE_SINGLE, PRE_SLOTS, grid_big, nullptr, nullptr, nullptr, nullptr);
        else if (mode == PRE_FILL) HMK_PRE(NbrPacked, PRE_FILL, PRE_SLOTS, grid_big, nullptr, nullptr, nullptr, nullptr);
        else HMK_PRE(NbrPacked, PRE_COUNT, PRE_SLOTS, grid_big, nullptr, nullptr, nullptr, nullptr);
    } else {
        if (mode == PRE_SINGLE) HMK_PRE(Nbr, PRE_SINGLE, PRE_SLOTS, grid_big, nullptr, nullptr, nullptr, nullptr);
        else if (mode == PRE_FILL) HMK_PRE(Nbr, PRE_FILL, PRE_SLOTS, grid_big, nullptr, nullptr, nullptr, nullptr);
        else HMK_PRE(Nbr, PRE_COUNT, PRE_SLOTS, grid_big, nullptr, nullptr, nullptr, nullptr);
    }
#undef HMK_PRE
    return hipGetLastError();
}

// exclusive scan of uint32 counts into uint32 start[n + 1] (tile_scratch: scan_scratch_bytes(n))
hipError_t launch_scan_u32(const uint32_t *counts, uint32_t *start, uint32_t n, uint64_t *tile_scratch, hipStream_t s) {
    launch_scan<uint32_t>(counts, start, n, tile_scratch, nullptr, s);
    return hipGetLastError();
}

}  // namespace hmk

namespace hmk {

__global__ void __launch_bounds__(256)
k_loop_init(uint32_t n_clusters, const long long *__restrict__ csize, const int32_t *__restrict__ cid, LoopCluster *__restrict__ cl,
            const uint32_t *__restrict__ sub_start, uint32_t *__restrict__ cursor, uint32_t nl, uint32_t *__restrict__ list,
            uint32_t *__restrict__ dirty, uint32_t *__restrict__ counters, uint32_t *__restrict__ taken, uint8_t *__restrict__ status,
            int32_t *__restrict__ join_slot) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_clusters) { cl[t] = LoopCluster{0, cid[t], csize[t]}; cursor[t] = sub_start[t]; taken[t] = 0; }
    if (t < nl) { list[t] = t; dirty[t] = 1; status[t] = LS_UNDECIDED; join_slot[t] = -1; }   // the first round evaluates every leftover
    if (t < 16) counters[t] = t == 4 ? nl : 0u;      // (these were four fills of their own: ~8 us each on the call's critical path)
}

// state of the device-side second loop before its first round: per-cluster records (16 bytes each) from the uploaded sizes
// and ids, the cursors at the heads of the (sorted) subscriber lists, eval list 0 = every leftover (counters: zeroed by
// the caller beforehand)
hipError_t launch_loop_init(uint32_t n_clusters, const long long *csize, const int32_t *cid, void *cl, const uint32_t *sub_start,
                            uint32_t *cursor, uint32_t nl, uint32_t *list, uint32_t *dirty, uint32_t *counters, uint32_t *taken,
                            uint8_t *status, int32_t *join_slot, hipStream_t s) {
    const uint32_t m = std::max<uint32_t>(std::max(n_clusters, nl), 16);
    hipLaunchKernelGGL(k_loop_init, dim3((m + 255) / 256), dim3(256), 0, s, n_clusters, csize, cid, (LoopCluster *)cl, sub_start, cursor,
                       nl, list, dirty, counters, taken, status, join_slot);
    return hipGetLastError();
}

// subscriber lists of the device-side second loop: cursor = zeroed uint32[n_clusters]; pass 0 leaves the counts in it
hipError_t launch_loop_subscribers(bool fill, uint32_t nl, const uint32_t *cand_start, const uint32_t *cand_cnt, const GreedyCand *cand,
                                   uint32_t *cursor, const uint32_t *sub_start, uint64_t *subs, hipStream_t s) {
    if (nl == 0) return hipSuccess;
    hipLaunchKernelGGL(k_loop_subscribers, dim3((nl + 255) / 256), dim3(256), 0, s, nl, cand_start, cand_cnt, cand, cursor, sub_start,
                       (unsigned long long *)subs, fill ? 1 : 0);
    return hipGetLastError();
}

// every cluster's list sorted by leftover; tmp: as large as subs
hipError_t launch_loop_sort_subscribers(uint32_t n_clusters, const uint32_t *sub_start, uint64_t *subs, uint64_t *tmp, hipStream_t s) {
    if (n_clusters == 0) return hipSuccess;
    hipLaunchKernelGGL(k_loop_sort_subs, dim3(n_clusters), dim3(256), 0, s, sub_start, (unsigned long long *)subs, (unsigned long long *)tmp);
    return hipGetLastError();
}

// one round of the device-side second loop (see k_loop_eval): eval, `passes` x (first, accept), apply.  counters: device
// uint32[8]; lists2: two uint32[nl] eval lists, list (round & 1) is read and the other written; first, taken, cursor:
// uint32[n_clusters] each (taken zeroed before the first round)
hipError_t launch_loop_round(bool packed, const RowPieces &rows, const uint32_t *leftover,
                             uint32_t nl, const uint32_t *cand_start, const uint32_t *cand_cnt, GreedyCand *cand, uint8_t *status,
                             uint32_t *choice, uint32_t *lists2, uint32_t *dirty, uint32_t round, uint32_t *first, uint32_t *taken,
                             uint32_t *cursor, uint32_t n_clusters, int passes, uint32_t *accepted, int32_t *join_slot,
                             const uint32_t *sub_start, const uint64_t *subs, void *clusters, const int32_t *seq_size,
                             uint32_t *counters, unsigned long long *host_word, int chain_mode, hipStream_t s) {
    if (nl == 0 || n_clusters == 0) return hipSuccess;
    LoopCluster *cl = (LoopCluster *)clusters;
    const dim3 grid((uint32_t)std::min<uint64_t>(LOOP_GRID, ((uint64_t)nl * 8 + 255) / 256)), block(256);
    const uint32_t which = round & 1u, stamp = round + 1;
    const uint32_t *list = lists2 + (size_t)which * nl;
    uint32_t *list_next = lists2 + (size_t)(which ^ 1u) * nl;
    const unsigned long long *sb = (const unsigned long long *)subs;
    hipLaunchKernelGGL(k_loop_eval_first, dim3(grid.x + (n_clusters + 3) / 4), block, 0, s, grid.x, list, which, cand_start, cand_cnt, cand, cl, status,
                       choice, dirty, counters, n_clusters, sub_start, sb, cursor, taken, stamp, first);
    for (int p = 0; p < passes; p++) {
        if (p > 0)
            hipLaunchKernelGGL(k_loop_first, dim3((n_clusters + 3) / 4), block, 0, s, n_clusters, sub_start, sb, cursor, cand, cl, status, taken, stamp, first);
        hipLaunchKernelGGL(k_loop_accept, dim3((n_clusters * ACCEPT_LANES + 255) / 256), block, 0, s, n_clusters, cand_start, cand_cnt, cand, cl, status, choice,
                           first, taken, stamp, accepted, join_slot, counters);
    }
    const dim3 agrid(APPLY_GRID);
    // Chains (k_loop_apply) pay when ONE cluster's joins are the loop's critical path -- families of near-duplicates that all join
    // one cluster: thousands of rounds become a handful.  Where they are not (10^5 / 10^6 random peptides, the reference's antibodies
    // example: 30 / 183 / 155 rounds, with or without), a chained join only makes its round longer: the joins it takes would have
    // ridden along in later rounds for free (loop 0.70 -> 0.81 ms, 29.5 -> 29.7 ms, 3.6 -> 4.0 ms).  So they start once a loop has
    // shown itself to be long; chain_mode 0 / 1 (HMK_LOOP_CHAIN) forces never / from the first round.
    const int chain = chain_mode >= 0 ? chain_mode : (round >= 256u ? 1 : 0);
    if (packed)
        hipLaunchKernelGGL((k_loop_apply<NbrPacked>), agrid, block, 0, s, rows, leftover, status, cand,
                           choice, accepted, sub_start, sb, cursor, cl, seq_size, dirty, list_next, which, counters, host_word, stamp,
                           cand_cnt, join_slot, chain);
    else
        hipLaunchKernelGGL((k_loop_apply<Nbr>), agrid, block, 0, s, rows, leftover, status, cand, choice,
                           accepted, sub_start, sb, cursor, cl, seq_size, dirty, list_next, which, counters, host_word, stamp,
                           cand_cnt, join_slot, chain);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------
// the band, prepared for phase 1 (BandPack, hmk_internal.h)
// -----------------------------------------------------------------------------
// firstPhase (LimitedGreedySequenceClusterer.java:77-120) reads the first rows of the graph in order.  Round 4 shipped the band's
// whole adjacency to the host (522 MB over PCIe at 10^6) and the host walked every row in full; here the device splits every
// band row x < R by where a neighbour lies:
//   near   (id < R)   kept, the ids above x first: the only neighbours whose state changes step by step;
//   far    (id >= R)  reduced to the row's FT best candidates in the reference's order (score, Cluster.size(), smaller id --
//                     ClinkageSequenceClusterer.java:166-173, :275-289) + a flag "there are more";
// and transposes the far part (far sequence -> the band rows that have it as a neighbour), of which the host gets the lists of the
// sequences that are some row's first or second far candidate -- what a new cluster {x, candidate} needs (:99-101, :108-110).
// Input: the band's CSR as k_edge_scatter leaves it (rows [upper | lower], bup[x] = the upper section's size), packed entries.
// One wave per row.
__device__ __forceinline__ unsigned long long band_far_key(uint32_t entry, const int32_t *__restrict__ seq_size) {
    const uint32_t id = entry >> 8;
    const uint32_t size = seq_size ? (uint32_t)seq_size[id] : 1u;
    return ((unsigned long long)(entry & 0xFFu) << 56) | ((unsigned long long)size << 24) | (unsigned long long)(~id & 0xFFFFFFu);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned long long w = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// Rows of at most BAND_REG_ROW upper entries keep their far keys in registers (one read of the row); longer rows are read again for
// every candidate taken (they come from the L2: a row is a few KB).
constexpr uint32_t BAND_REG_KEYS = 8, BAND_REG_ROW = 64 * BAND_REG_KEYS;
__global__ void __launch_bounds__(256)
k_band_split(const uint64_t *__restrict__ bstart, const uint32_t *__restrict__ bup, const uint32_t *__restrict__ badj, uint32_t R, uint32_t ft,
             const int32_t *__restrict__ seq_size, uint32_t *__restrict__ near_cnt, uint32_t *__restrict__ near_up,
             uint32_t *__restrict__ far_top, uint8_t *__restrict__ far_more, uint32_t *__restrict__ fdeg, uint32_t nt,
             uint32_t *__restrict__ near_top) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t x = blockIdx.x * 4 + (threadIdx.x >> 6); x < R; x += gridDim.x * 4) {   // wave-uniform
        const uint64_t b = bstart[x], e = bstart[x + 1];
        const uint32_t up = bup[x];
        const bool in_regs = up <= BAND_REG_ROW;   // wave-uniform
        unsigned long long keys[BAND_REG_KEYS];   // of every upper entry; is_far: which side it is on
        bool is_far[BAND_REG_KEYS];
#pragma unroll
        for (uint32_t j = 0; j < BAND_REG_KEYS; j++) { keys[j] = 0; is_far[j] = false; }
        uint32_t n_near_up = 0, n_far = 0;
        auto visit = [&](uint32_t k, unsigned long long *key_out, bool *far_out) {
            const uint32_t ent = k < up ? badj[b + k] : 0u;
            const bool far = k < up && (ent >> 8) >= R;
            if (far) atomicAdd(&fdeg[ent >> 8], 1u);
            if (key_out) { *key_out = k < up ? band_far_key(ent, seq_size) : 0ull; *far_out = far; }
            n_far += (uint32_t)__popcll(__ballot(far));
            n_near_up += (uint32_t)__popcll(__ballot(k < up && !far));
        };
        if (in_regs) {
#pragma unroll
            for (uint32_t j = 0; j < BAND_REG_KEYS; j++)
                if (j * 64 < up) visit(j * 64 + lane, &keys[j], &is_far[j]);   // wave-uniform test
        } else {
            for (uint32_t k0 = 0; k0 < up; k0 += 64) visit(k0 + lane, nullptr, nullptr);
        }
        // the FT best far candidates, one per turn: the largest key below the last one taken (keys are unique: they end in the id;
        // a key is never 0: Cluster.size() >= 1)
        // (the same for the near neighbours above x -- the candidates whose state changes step by step: near_top lists the nt best, and the
        // host scans the whole near row only when every listed one has been taken)
        for (int side = 0; side < 2; side++) {   // 0: far, 1: near
            const uint32_t cnt_side = side == 0 ? n_far : n_near_up, take = side == 0 ? ft : nt;
            uint32_t *top = side == 0 ? far_top : near_top;
            unsigned long long bound = ~0ull;
            for (uint32_t t = 0; t < take; t++) {
                unsigned long long best = 0;
                if (t < cnt_side) {
                    if (in_regs) {
#pragma unroll
                        for (uint32_t j = 0; j < BAND_REG_KEYS; j++)
                            if (is_far[j] == (side == 0) && keys[j] < bound && keys[j] > best) best = keys[j];
                    } else {
                        for (uint32_t k = lane; k < up; k += 64) {
                            const uint32_t ent = badj[b + k];
                            if (((ent >> 8) >= R) != (side == 0)) continue;
                            const unsigned long long key = band_far_key(ent, seq_size);
                            if (key < bound && key > best) best = key;
                        }
                    }
                    best = wave_max_u64(best);
                }
                if (lane == 0) top[(size_t)x * take + t] = best ? ((uint32_t)(~best & 0xFFFFFFull) << 8) | (uint32_t)(best >> 56) : ~0u;
                bound = best;   // (0 once the row has no more: nothing lies below it)
            }
        }
        if (lane == 0) {
            near_up[x] = n_near_up;
            near_cnt[x] = n_near_up + (uint32_t)(e - b - up);   // + the lower section: ids below x, all of them near
            far_more[x] = n_far > ft ? 1 : 0;
        }
    }
}

// Where every far sequence's transposed list starts: fstart[id] = a block of fdeg[id] entries taken from one counter, a wave's 64
// blocks with ONE atomic (prefix sums inside the wave) -- the lists need no order among themselves.  *counter zeroed.
// The same launch also sizes what travels for every slot u = TR * x + t (row x's t-th far candidate B): room for the later band rows
// that have BOTH x and B as neighbours, at most min(rows that have B, x's near neighbours above it) -- k_band_isect writes the list.
__global__ void __launch_bounds__(256)
k_band_falloc(const uint32_t *__restrict__ fdeg, uint32_t n, uint32_t *__restrict__ fstart, uint32_t *__restrict__ counter,
              const uint32_t *__restrict__ far_top, uint32_t R, uint32_t ft, uint32_t tr_per_row, const uint32_t *__restrict__ near_up,
              uint32_t *__restrict__ tr_cnt) {
    for (uint32_t u = blockIdx.x * 256 + threadIdx.x; u < R * tr_per_row; u += gridDim.x * 256) {
        const uint32_t x = u / tr_per_row, t = u - x * tr_per_row;
        const uint32_t ent = t < ft ? far_top[(size_t)x * ft + t] : ~0u;
        tr_cnt[u] = ent != ~0u ? min(fdeg[ent >> 8], near_up[x]) : 0u;
    }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; k0 < n; k0 += gridDim.x * 256) {   // wave-uniform
        const uint32_t id = k0 + lane;
        const uint32_t v = id < n ? fdeg[id] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63, 64);
        uint32_t base = 0;
        if (lane == 0 && tot) base = atomicAdd(counter, tot);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (id < n) fstart[id] = base + incl - v;
    }
}

// One workgroup of 256 threads (NOT more: a larger workgroup waits for a CU with that many free wave slots while the scoring pass
// keeps every CU full of 256-thread workgroups -- a 1,024-thread one waited for the END of the pass), between the split and the
// fill: exclusive prefix sums near_cnt -> near_start[R + 1] and tr_cnt -> tr_start[R * TR + 1]; totals[0] = near entries,
// totals[1] = entries of the travelling lists.
__device__ void band_block_scan(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t n, uint32_t *wsum) {   // wsum: LDS uint32[8]
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint32_t carry = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += 1024) {   // tiles of 4 consecutive counters per thread (coalesced), workgroup-uniform
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t k = k0 + t * 4 + q; v[q] = k < n ? src[k] : 0u; sum += v[q]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((int)lane >= d) incl += u;
        }
        __syncthreads();   // (wsum of the previous tile has been read)
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wv; w++) before += wsum[w];
        uint32_t run = before + incl - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t k = k0 + t * 4 + q; if (k < n) dst[k] = run; run += v[q]; }
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (t == 0) dst[n] = carry;
}
__global__ void __launch_bounds__(256)
k_band_offsets(uint32_t R, uint32_t tr_per_row, const uint32_t *__restrict__ near_cnt, uint32_t *__restrict__ near_start,
               const uint32_t *__restrict__ tr_cnt, uint32_t *__restrict__ tr_start, uint32_t *__restrict__ totals) {
    __shared__ uint32_t wsum[8];
    band_block_scan(near_cnt, near_start, R, wsum);
    __syncthreads();
    band_block_scan(tr_cnt, tr_start, R * tr_per_row, wsum);
    __syncthreads();
    if (threadIdx.x == 0) { totals[0] = near_start[R]; totals[1] = tr_start[R * tr_per_row]; }
}

// near rows (upper section's near entries, then the lower section) and the transposed far part: fadj[fstart[id] ..) = the band
// rows that have far sequence id as a neighbour, band row << 8 | (score - base).
// `near` and the h_* arrays are the HOST's block (pinned, mapped): the kernel's stores travel over PCIe as they are made -- no copy
// afterwards (a copy is a launch of its own that waits for a slot beside the pass: 0.9 ms for 2.5 MB at 10^5).
__global__ void __launch_bounds__(256)
k_band_fill(const uint64_t *__restrict__ bstart, const uint32_t *__restrict__ bup, const uint32_t *__restrict__ badj, uint32_t R, uint32_t ft,
            const uint32_t *__restrict__ near_start, const uint32_t *__restrict__ near_up, const uint32_t *__restrict__ far_top,
            const uint8_t *__restrict__ far_more, uint32_t *__restrict__ near, const uint32_t *__restrict__ fstart,
            uint32_t *__restrict__ fcur, uint32_t *__restrict__ fadj, uint32_t *__restrict__ h_near_start, uint32_t *__restrict__ h_near_up,
            uint32_t *__restrict__ h_far_top, uint8_t *__restrict__ h_far_more, uint32_t nt, const uint32_t *__restrict__ near_top,
            uint32_t *__restrict__ h_near_top) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t x = blockIdx.x * 4 + (threadIdx.x >> 6); x < R; x += gridDim.x * 4) {   // wave-uniform
        const uint64_t b = bstart[x], e = bstart[x + 1];
        const uint32_t up = bup[x], len = (uint32_t)(e - b);
        uint32_t w = near_start[x];
        if (lane == 0) { h_near_start[x] = w; h_near_up[x] = near_up[x]; h_far_more[x] = far_more[x]; if (x + 1 == R) h_near_start[R] = near_start[R]; }
        if (lane < ft) h_far_top[(size_t)x * ft + lane] = far_top[(size_t)x * ft + lane];
        if (lane < nt) h_near_top[(size_t)x * nt + lane] = near_top[(size_t)x * nt + lane];
        for (uint32_t k0 = 0; k0 < len; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t ent = k < len ? badj[b + k] : 0u;
            const bool far = k < up && (ent >> 8) >= R;
            const bool keep = k < len && !far;
            const uint64_t mask = __ballot(keep);
            if (keep) near[w + mbcnt64(mask)] = ent;
            w += (uint32_t)__popcll(mask);
            if (far) fadj[fstart[ent >> 8] + atomicAdd(&fcur[ent >> 8], 1u)] = (x << 8) | (ent & 0xFFu);
        }
    }
}

// What seeding a cluster {x, B} needs (LimitedGreedySequenceClusterer.java:99-101, :108-110 with a far B): the LATER band rows that have
// both x and B as neighbours, each with min(score(x, row), score(B, row)) -- the new cluster's score for that row
// (ClinkageClusterScorer.java:36-48).  Round 5 first shipped B's whole transposed list and let the host filter it against x's stamped
// row (a third of phase 1's host time); here the device intersects: one workgroup per band row x hashes x's near neighbours above it
// (id -> score) in LDS, and wave t walks the list of x's t-th far candidate and writes the matches straight into the host's block:
// tr[tr_start[u] ..) = row << 8 | min score, h_tr_cnt[u] = how many.  A row with more near neighbours than the table takes, or a list
// the block has no room for, gets h_tr_cnt[u] = ~0: the host then asks for B's whole list and filters it itself (the old way).
constexpr uint32_t ISECT_SLOTS = 4096, ISECT_MAX_KEYS = 2800;
__global__ void __launch_bounds__(256)
k_band_isect(const uint64_t *__restrict__ bstart, const uint32_t *__restrict__ bup, const uint32_t *__restrict__ badj, uint32_t R, uint32_t ft,
             uint32_t tr_per_row, const uint32_t *__restrict__ near_up, const uint32_t *__restrict__ far_top, const uint32_t *__restrict__ fstart,
             const uint32_t *__restrict__ fdeg, const uint32_t *__restrict__ fadj, const uint32_t *__restrict__ tr_start, uint64_t tr_cap,
             uint32_t *__restrict__ tr, uint32_t *__restrict__ h_tr_start, uint32_t *__restrict__ h_tr_cnt) {
    __shared__ uint32_t table[ISECT_SLOTS];   // id + 1 << 8 | score; 0 = empty
    const uint32_t lane = threadIdx.x & 63u, t = threadIdx.x >> 6;
    for (uint32_t x = blockIdx.x; x < R; x += gridDim.x) {   // workgroup-uniform
        const uint64_t b = bstart[x];
        const uint32_t up = bup[x], keys = near_up[x];
        const bool hashed = keys <= ISECT_MAX_KEYS;
        __syncthreads();   // (the previous row's probes are done)
        if (hashed) {
            for (uint32_t k = threadIdx.x; k < ISECT_SLOTS; k += 256) table[k] = 0u;
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < up; k += 256) {
                const uint32_t ent = badj[b + k], id = ent >> 8;
                if (id >= R) continue;                         // far
                const uint32_t v = ((id + 1u) << 8) | (ent & 0xFFu);
                uint32_t h = (id * 2654435761u) >> 20;     // 12 bits
                while (atomicCAS(&table[h], 0u, v) != 0u) h = (h + 1u) & (ISECT_SLOTS - 1u);
            }
        }
        __syncthreads();
        if (t >= tr_per_row) continue;
        const uint32_t u = x * tr_per_row + t;
        const uint32_t ent = t < ft ? far_top[(size_t)x * ft + t] : ~0u;
        const uint32_t dst = tr_start[u];
        if (lane == 0) { h_tr_start[u] = dst; if (u + 1 == R * tr_per_row) h_tr_start[u + 1] = tr_start[u + 1]; }
        if (ent == ~0u) { if (lane == 0) h_tr_cnt[u] = 0u; continue; }
        const uint32_t id = ent >> 8, src = fstart[id], cnt = fdeg[id];
        if (!hashed || (uint64_t)dst + min(cnt, keys) > tr_cap) { if (lane == 0) h_tr_cnt[u] = ~0u; continue; }
        uint32_t w = 0;
        for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {            // wave-uniform
            const uint32_t k = k0 + lane;
            const uint32_t fe = k < cnt ? fadj[src + k] : 0u, y = fe >> 8;
            uint32_t found = 0;
            if (k < cnt && y > x) {
                uint32_t h = (y * 2654435761u) >> 20;
                for (uint32_t v = table[h]; v != 0u; v = table[h]) {
                    if ((v >> 8) == y + 1u) { found = v; break; }
                    h = (h + 1u) & (ISECT_SLOTS - 1u);
                }
            }
            const uint64_t mask = __ballot(found != 0u);
            if (found) tr[dst + w + mbcnt64(mask)] = (y << 8) | min(found & 0xFFu, fe & 0xFFu);
            w += (uint32_t)__popcll(mask);
        }
        if (lane == 0) h_tr_cnt[u] = w;
    }
}

// Workgroups of the hand-over's kernels: they run BESIDE the scoring pass, where every workgroup has to wait for a slot and costs the
// pass (band_grid_x above) -- but a wave walks its rows one after the other, each a chain of dependent loads: sized for the band's
// entries at ~4,096 per workgroup, 32 .. 1024 (10^5: 366 for 1.5 x 10^6 entries, four rows per wave).
static uint32_t band_prep_grid(uint64_t entries, uint32_t items) {
    const uint64_t by_work = std::max<uint64_t>(32, std::min<uint64_t>(1024, entries / 4096));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(by_work, ((uint64_t)items + 3) / 4));
}
// The whole preparation behind the band's CSR, one stream, no host round trip: split -> list allocation + room for what travels -> the two
// prefix sums -> near rows and transposed far part -> the intersections.  h_*: the host's pinned block (device-visible addresses).
hipError_t launch_band_prepare(const uint64_t *bstart, const uint32_t *bup, const void *badj, uint32_t R, uint64_t entries, uint32_t n, uint32_t ft,
                               uint32_t tr_per_row, const int32_t *seq_size, uint32_t *near_cnt, uint32_t *near_up, uint32_t *near_start, uint32_t *far_top,
                               uint8_t *far_more, uint32_t *fdeg, uint32_t *fcur, uint32_t *totals, uint32_t *fstart, uint32_t *fadj,
                               uint32_t *tr_cnt, uint32_t *tr_start, uint64_t tr_cap, uint32_t *h_near_start, uint32_t *h_near_up, uint32_t *h_far_top,
                               uint8_t *h_far_more, uint32_t *h_near, uint32_t *h_tr_cnt, uint32_t *h_tr_start, uint32_t *h_tr, uint32_t nt,
                               uint32_t *near_top, uint32_t *h_near_top, hipStream_t s) {
    if (R == 0) return hipSuccess;
    if (tr_per_row > 4) return hipErrorInvalidValue;   // (k_band_isect: one wave of its workgroup per candidate)
    const uint32_t *adj = (const uint32_t *)badj;
    hipLaunchKernelGGL(k_band_split, dim3(band_prep_grid(entries, R)), dim3(256), 0, s, bstart, bup, adj, R, ft, seq_size, near_cnt, near_up, far_top, far_more, fdeg, nt, near_top);
    hipLaunchKernelGGL(k_band_falloc, dim3(std::max<uint32_t>(8, std::min<uint32_t>(256, n / 16384))), dim3(256), 0, s, fdeg, n, fstart, totals + 2,
                       far_top, R, ft, tr_per_row, near_up, tr_cnt);
    hipLaunchKernelGGL(k_band_offsets, dim3(1), dim3(256), 0, s, R, tr_per_row, near_cnt, near_start, tr_cnt, tr_start, totals);
    hipLaunchKernelGGL(k_band_fill, dim3(band_prep_grid(entries, R)), dim3(256), 0, s, bstart, bup, adj, R, ft, near_start, near_up, far_top, far_more, h_near,
                       fstart, fcur, fadj, h_near_start, h_near_up, h_far_top, h_far_more, nt, near_top, h_near_top);
    hipLaunchKernelGGL(k_band_isect, dim3(std::min<uint32_t>(R, band_prep_grid(entries, R * 4))), dim3(256), 0, s, bstart, bup, adj, R, ft, tr_per_row, near_up,
                       far_top, fstart, fdeg, fadj, tr_start, tr_cap, h_tr, h_tr_start, h_tr_cnt);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------
// multi-device calls: every row's adjacency is built on the device that OWNS the row
// -----------------------------------------------------------------------------
// Device d owns the rows [d * rows_per, (d + 1) * rows_per).  A device has scored a shard of the pair space; each of its edges
// (x, m) goes to owner(x) and to owner(m) (once if they are the same device): one block per destination, counted first so that the
// blocks lie back to back in `out` (off[dst] .. off[dst + 1]).  The blocks then travel device to device (all pairs at once: every
// xGMI link carries its own pair's traffic, none more than 2 / G of a shard) instead of all of them to one root.
__global__ void __launch_bounds__(256)
k_route_count(const EdgeSegs segs, uint32_t rows_per, uint32_t g, unsigned long long *__restrict__ cnt) {
    // (per-thread counters in registers, folded per wave at the end: one LDS atomic per edge on a handful of addresses -- every lane of a
    // wave hitting the same counter -- made this pass 1-5 ms for the 1.3 GB of a 1/8 shard at 10^6)
    __shared__ uint32_t hist[HMK_MAX_DEVICES];
    if (threadIdx.x < HMK_MAX_DEVICES) hist[threadIdx.x] = 0;
    __syncthreads();
    const EdgeSeg sg = segs.s[blockIdx.y];
    const uint64_t n_edges = min((uint64_t)*sg.count, sg.cap);
    uint32_t c[HMK_MAX_DEVICES];
#pragma unroll
    for (uint32_t o = 0; o < HMK_MAX_DEVICES; o++) c[o] = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_edges; k += (uint64_t)gridDim.x * 256) {
        const uint64_t e = sg.edges[k];
        const uint32_t dx = min(HMK_EDGE_X(e) / rows_per, g - 1), dm = min(HMK_EDGE_M(e) / rows_per, g - 1);
#pragma unroll
        for (uint32_t o = 0; o < HMK_MAX_DEVICES; o++) c[o] += (uint32_t)(dx == o) + (uint32_t)((dm == o) & (dm != dx));
    }
#pragma unroll
    for (uint32_t o = 0; o < HMK_MAX_DEVICES; o++) {
        uint32_t v = c[o];
#pragma unroll
        for (int d = 32; d; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&hist[o], v);
    }
    __syncthreads();
    if (threadIdx.x < g && hist[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}
// off[0 .. g] = prefix sums of cnt, cur[] = 0 (one thread)
__global__ void k_route_offsets(const unsigned long long *__restrict__ cnt, uint32_t g, unsigned long long *__restrict__ off, unsigned long long *__restrict__ cur) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long run = 0;
    for (uint32_t d = 0; d < g; d++) { off[d] = run; run += cnt[d]; cur[d] = 0; }
    off[g] = run;
}
// ROUTE_ITEMS edges per thread and step: a step ends in one returning atomic per owner on the block cursors and three barriers, and with
// 256 edges per step those were most of the kernel (7.4 ms for a 1/8 shard at 10^6, whatever else ran beside it); a block's share of a
// step is a run of a few hundred edges now.
constexpr int ROUTE_ITEMS = 8;
__global__ void __launch_bounds__(256)
k_route_fill(const EdgeSegs segs, uint32_t rows_per, uint32_t g, const unsigned long long *__restrict__ off, unsigned long long *__restrict__ cur,
             uint64_t *__restrict__ out, unsigned long long out_cap) {
    __shared__ uint32_t hist[HMK_MAX_DEVICES];
    __shared__ unsigned long long base[HMK_MAX_DEVICES];
    const EdgeSeg sg = segs.s[blockIdx.y];
    const uint64_t n_edges = min((uint64_t)*sg.count, sg.cap);
    for (uint64_t k0 = (uint64_t)blockIdx.x * (256 * ROUTE_ITEMS); k0 < n_edges; k0 += (uint64_t)gridDim.x * (256 * ROUTE_ITEMS)) {   // workgroup-uniform
        if (threadIdx.x < HMK_MAX_DEVICES) hist[threadIdx.x] = 0;
        __syncthreads();
        uint64_t e[ROUTE_ITEMS];
        uint32_t px[ROUTE_ITEMS], pm[ROUTE_ITEMS];
#pragma unroll
        for (int u = 0; u < ROUTE_ITEMS; u++) {   // (all loads in flight before the first LDS atomic)
            const uint64_t k = k0 + (uint64_t)u * 256 + threadIdx.x;
            e[u] = k < n_edges ? sg.edges[k] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < ROUTE_ITEMS; u++) {
            const uint64_t k = k0 + (uint64_t)u * 256 + threadIdx.x;
            px[u] = pm[u] = 0;
            if (k < n_edges) {
                const uint32_t dx = min(HMK_EDGE_X(e[u]) / rows_per, g - 1), dm = min(HMK_EDGE_M(e[u]) / rows_per, g - 1);
                px[u] = atomicAdd(&hist[dx], 1u);
                if (dm != dx) pm[u] = atomicAdd(&hist[dm], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < g && hist[threadIdx.x]) base[threadIdx.x] = off[threadIdx.x] + atomicAdd(&cur[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ROUTE_ITEMS; u++) {
            const uint64_t k = k0 + (uint64_t)u * 256 + threadIdx.x;
            if (k < n_edges) {
                const uint32_t dx = min(HMK_EDGE_X(e[u]) / rows_per, g - 1), dm = min(HMK_EDGE_M(e[u]) / rows_per, g - 1);
                if (base[dx] + px[u] < out_cap) out[base[dx] + px[u]] = e[u];
                if (dm != dx && base[dm] + pm[u] < out_cap) out[base[dm] + pm[u]] = e[u];
            }
        }
        __syncthreads();
    }
}
hipError_t launch_route_edges(const EdgeSegs &segs, uint32_t rows_per, uint32_t g, unsigned long long *cnt, unsigned long long *off, unsigned long long *cur,
                              uint64_t *out, uint64_t out_cap, hipStream_t s) {
    if (g == 0 || g > HMK_MAX_DEVICES || segs.n == 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(cnt, 0, HMK_MAX_DEVICES * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_route_count, dim3(128, segs.n), dim3(256), 0, s, segs, rows_per, g, cnt);
    hipLaunchKernelGGL(k_route_offsets, dim3(1), dim3(64), 0, s, cnt, g, off, cur);
    hipLaunchKernelGGL(k_route_fill, dim3(128, segs.n), dim3(256), 0, s, segs, rows_per, g, off, cur, out, (unsigned long long)out_cap);
    return hipGetLastError();
}

// The rows' degree counters of the OWNING device: every device counted upper and lower degrees (deg[0, n) and deg[n, 2n)) of ALL rows over
// the edges it scored; the owner of [r0, r1) adds the other devices' counters of its rows (slices[d]: up[r1 - r0] then lo[r1 - r0], g - 1 of
// them) to its own and clears the rows it does not own -- the CSR it builds holds its rows only.
struct DegSlices { const uint32_t *p[HMK_MAX_DEVICES]; uint32_t n; };
__global__ void __launch_bounds__(256)
k_owned_degrees(uint32_t *__restrict__ deg, uint32_t n, uint32_t r0, uint32_t r1, const DegSlices slices) {
    const uint32_t len = r1 - r0;
    for (uint32_t x = blockIdx.x * 256 + threadIdx.x; x < n; x += gridDim.x * 256) {
        uint32_t up = 0, lo = 0;
        if (x >= r0 && x < r1) {
            up = deg[x];
            lo = deg[n + x];
            for (uint32_t d = 0; d < slices.n; d++) { up += slices.p[d][x - r0]; lo += slices.p[d][len + x - r0]; }
        }
        deg[x] = up;
        deg[n + x] = lo;
    }
}
hipError_t launch_owned_degrees(uint32_t *deg, uint32_t n, uint32_t r0, uint32_t r1, const uint32_t *const *slices, uint32_t n_slices, hipStream_t s) {
    if (n_slices > HMK_MAX_DEVICES) return hipErrorInvalidValue;
    DegSlices ds{};
    ds.n = n_slices;
    for (uint32_t d = 0; d < n_slices; d++) ds.p[d] = slices[d];
    hipLaunchKernelGGL(k_owned_degrees, dim3(std::min<uint32_t>((n + 255) / 256, 2048)), dim3(256), 0, s, deg, n, r0, r1, ds);
    return hipGetLastError();
}

// dst[k] += src[k]: the row degrees the peers of a multi-device call counted while they wrote their edges, added to the root's
__global__ void __launch_bounds__(256) k_add_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t n) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) dst[k] += src[k];
}
hipError_t launch_add_u32(uint32_t *dst, const uint32_t *src, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_u32, dim3(std::min<uint32_t>((n + 255) / 256, 2048)), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}

// loads this translation unit's code object (HIP defers that to the first launch: 5-10 ms of the first call otherwise)
hipError_t warm_edges_module() {
    // ... and resolves the kernels of the default clustering path (4-byte adjacency), each a one-time 0.1-0.3 ms otherwise
    const void *kernels[] = {
        (const void *)&k_init_range, (const void *)&k_edge_degree, (const void *)&k_scan_tile_sums, (const void *)&k_scan_tile_offsets,
        (const void *)&k_scan_tiles<uint64_t>, (const void *)&k_scan_tiles<uint32_t>, (const void *)&k_edge_scatter<NbrPacked>,
        (const void *)&k_cluster_bitmap, (const void *)&k_greedy_precheck<NbrPacked, PRE_SINGLE, PRE_SLOTS>, (const void *)&k_greedy_precheck<NbrPacked, PRE_SINGLE, PRE_SLOTS_SMALL>,
        (const void *)&k_loop_subscribers, (const void *)&k_loop_sort_subs, (const void *)&k_loop_init, (const void *)&k_loop_eval_first,
        (const void *)&k_loop_first, (const void *)&k_loop_accept, (const void *)&k_loop_apply<NbrPacked>};
    hipError_t e = hipSuccess;
    for (const void *k : kernels) {
        hipFuncAttributes a;
        const hipError_t r = hipFuncGetAttributes(&a, k);
        if (r != hipSuccess) e = r;
    }
    return e;
}

}  // namespace hmk
