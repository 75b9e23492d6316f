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
k_edge_degree(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
              uint32_t *__restrict__ deg, uint32_t *__restrict__ up, int symmetric, int *__restrict__ score_range, uint32_t n) {
    const uint32_t shard = blockIdx.y;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
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
        // symmetric: the edge is stored under both ends; under its SMALLER end it is an "upper" neighbour (id above
        // the row's own), counted in up[] as well -- rows are laid out upper neighbours first (k_edge_scatter)
        const uint32_t ea = symmetric ? min(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_X(e);
        const uint32_t eb = symmetric ? max(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_M(e);
        const WaveGroup g = wave_groups(ea, valid);
        if (valid && g.rank == 0) {
            atomicAdd(&deg[ea], g.size);
            if (symmetric) atomicAdd(&up[ea], g.size);
        }
        if (valid && symmetric) atomicAdd(&deg[eb], 1u);
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

// Exclusive scan of deg[n] -> start[n + 1] in three coalesced passes over tiles of 2048 counters:
// tile sums, scan of the tile sums (one block), tile-local scan + offset.  T = uint32 or uint64.
constexpr uint32_t SCAN_TILE = 2048;

__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *__restrict__ deg, uint64_t *__restrict__ tile_sum, uint32_t n) {
    __shared__ uint64_t red[4];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint64_t v = 0;
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {
        const uint32_t k = base + j * 256 + threadIdx.x;
        if (k < n) v += deg[k];
    }
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one block: exclusive scan of the n_tiles tile sums in place; tile_sum[n_tiles] = grand total
__global__ void __launch_bounds__(1024) k_scan_tile_offsets(uint64_t *__restrict__ tile_sum, uint32_t n_tiles) {
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n_tiles + 1023) / 1024;
    const uint32_t lo = min(n_tiles, tid * chunk), hi = min(n_tiles, lo + chunk);
    uint64_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += tile_sum[k];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    for (uint32_t k = lo; k < hi; k++) { const uint64_t d = tile_sum[k]; tile_sum[k] = run; run += d; }
    if (tid == 1023) tile_sum[n_tiles] = part[1023];
}

template <typename T>
__global__ void __launch_bounds__(256)
k_scan_tiles(const uint32_t *__restrict__ deg, const uint64_t *__restrict__ tile_off, T *__restrict__ start, uint32_t n,
             uint32_t n_tiles, const uint32_t *__restrict__ tail_word) {
    __shared__ uint32_t v[SCAN_TILE];
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * SCAN_TILE, tid = threadIdx.x;
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) {  // coalesced load
        const uint32_t k = base + j * 256 + tid;
        v[j * 256 + tid] = k < n ? deg[k] : 0;
    }
    __syncthreads();
    uint32_t loc[SCAN_TILE / 256], sum = 0;           // thread tid owns the 8 consecutive counters tid * 8 ..
    for (uint32_t j = 0; j < SCAN_TILE / 256; j++) { loc[j] = sum; sum += v[tid * (SCAN_TILE / 256) + j]; }
    uint32_t inc = sum;                               // inclusive scan of the thread sums inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if ((tid & 63) >= (uint32_t)o) inc += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) woff += wsum[w];
    const uint32_t excl = woff + inc - sum;
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
        if (tail_word) start[n + 1] = (T)*tail_word;
    }
}

// tile_scratch: uint64[ceil(n / SCAN_TILE) + 1]
template <typename T>
static void launch_scan(const uint32_t *deg, T *start, uint32_t n, uint64_t *tile_scratch, const uint32_t *tail_word,
                        hipStream_t s) {
    const uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, s, deg, tile_scratch, n);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(1024), 0, s, tile_scratch, n_tiles);
    hipLaunchKernelGGL((k_scan_tiles<T>), dim3(n_tiles), dim3(256), 0, s, deg, tile_scratch, start, n, n_tiles, tail_word);
}

// NbrT = Nbr: {m, score}.  NbrT = NbrPacked: m << 8 | (score - base), the caller has checked the score range.
// symmetric: row r = [neighbours with id > r (up[r] of them) | neighbours with id < r]; cursor = uint32[2 n],
// one cursor per section.  The host merge only has to look at the first section when a sequence joins a cluster.
template <class NbrT>
__global__ void __launch_bounds__(256)
k_edge_scatter(const uint64_t *__restrict__ edges, uint64_t cap_per_shard, const unsigned long long *__restrict__ counts,
               const uint64_t *__restrict__ start, const uint32_t *__restrict__ up, uint32_t *__restrict__ cursor,
               NbrT *__restrict__ adj, int symmetric, int base, uint32_t n) {
    const uint32_t shard = blockIdx.y;
    const uint64_t cnt = min((uint64_t)counts[shard], cap_per_shard);
    const uint64_t *seg = edges + (uint64_t)shard * cap_per_shard;
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); k0 < cnt; k0 += (uint64_t)gridDim.x * 256) {
        const uint64_t k = k0 + (threadIdx.x & 63);
        const bool valid = k < cnt;
        const uint64_t e = valid ? seg[k] : 0;
        const uint32_t x = symmetric ? min(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_X(e);
        const uint32_t m = symmetric ? max(HMK_EDGE_X(e), HMK_EDGE_M(e)) : HMK_EDGE_M(e);
        const int32_t s = HMK_EDGE_SCORE(e);
        const WaveGroup g = wave_groups(x, valid);   // one atomic per distinct x of the wave
        uint32_t basex = 0;
        if (valid && g.rank == 0) basex = atomicAdd(&cursor[x], g.size);
        basex = (uint32_t)__shfl((int)basex, (int)g.leader, 64);
        if (!valid) continue;
        const uint64_t px = start[x] + basex + g.rank;
        if constexpr (sizeof(NbrT) == 4) {
            const uint32_t rel = (uint32_t)(s - base) & 0xFFu;
            adj[px] = NbrT{(m << 8) | rel};
            if (symmetric) adj[start[m] + up[m] + atomicAdd(&cursor[n + m], 1u)] = NbrT{(x << 8) | rel};
        } else {
            adj[px] = NbrT{m, s};
            if (symmetric) adj[start[m] + up[m] + atomicAdd(&cursor[n + m], 1u)] = NbrT{x, s};
        }
    }
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
// clusters are counted in a per-wave LDS hash table (key = cluster, count, min score).  FILL = false counts the
// candidates of each leftover, FILL = true writes them at cand_start[q]; a row with more distinct clusters than
// the table takes raises *overflow (the host then runs its own pre-check).
constexpr int PRE_SLOTS = 1024;   // per wave; 3 x 4 KB

__device__ __forceinline__ uint32_t nbr_id(const Nbr &a) { return a.m; }
__device__ __forceinline__ int32_t nbr_score(const Nbr &a) { return a.s; }
__device__ __forceinline__ uint32_t nbr_id(const NbrPacked &a) { return a.v >> 8; }
__device__ __forceinline__ int32_t nbr_score(const NbrPacked &a) { return (int32_t)(a.v & 0xFFu); }

template <class NbrT, bool FILL>
__global__ void __launch_bounds__(256)
k_greedy_precheck(const uint64_t *__restrict__ start, const NbrT *__restrict__ adj, const int32_t *__restrict__ cluster_of,
                  const int32_t *__restrict__ usize, const uint32_t *__restrict__ leftover, uint32_t nl,
                  uint32_t *__restrict__ cand_cnt, const uint32_t *__restrict__ cand_start, GreedyCand *__restrict__ cand,
                  uint32_t *__restrict__ overflow) {
    __shared__ int32_t keys_all[4 * PRE_SLOTS];
    __shared__ uint32_t cnt_all[4 * PRE_SLOTS];
    __shared__ int32_t mn_all[4 * PRE_SLOTS];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t *keys = keys_all + wv * PRE_SLOTS;
    uint32_t *cnt = cnt_all + wv * PRE_SLOTS;
    int32_t *mn = mn_all + wv * PRE_SLOTS;
    for (uint32_t q = blockIdx.x * 4 + wv; q < nl; q += gridDim.x * 4) {
        for (uint32_t sl = lane; sl < PRE_SLOTS; sl += 64) { keys[sl] = -1; cnt[sl] = 0; mn[sl] = INT_MAX; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const uint32_t y = leftover[q];
        const uint64_t b = start[y], e = start[y + 1];
        bool full = false;
        for (uint64_t k = b + lane; k < e; k += 64) {
            const NbrT nb = adj[k];
            const int32_t c = cluster_of[nbr_id(nb)];
            if (c < 0) continue;
            uint32_t sl = ((uint32_t)c * 2654435761u) >> 22;   // top 10 bits
            int probes = 0;
            for (;;) {
                const int32_t old = atomicCAS(&keys[sl], -1, c);
                if (old == -1 || old == c) {
                    atomicAdd(&cnt[sl], 1u);
                    atomicMin(&mn[sl], nbr_score(nb));
                    break;
                }
                sl = (sl + 1) & (PRE_SLOTS - 1);
                if (++probes >= PRE_SLOTS) { full = true; break; }
            }
        }
        if (__ballot(full) != 0) {
            if (lane == 0) atomicAdd(overflow, 1u);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        uint32_t found = 0;
        for (uint32_t s0 = 0; s0 < PRE_SLOTS; s0 += 64) {
            const uint32_t sl = s0 + lane;
            const int32_t c = keys[sl];
            const bool ok = c >= 0 && (int32_t)cnt[sl] == usize[c];   // every member of c is a neighbour of y
            const uint64_t mask = __ballot(ok);
            if (FILL && ok) cand[cand_start[q] + found + mbcnt64(mask)] = GreedyCand{c, mn[sl], 0};
            found += (uint32_t)__popcll(mask);
        }
        if (!FILL && lane == 0) cand_cnt[q] = found;
    }
}

// -----------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------
hipError_t launch_csr_degree_scan(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts, uint32_t n,
                                  bool symmetric, uint32_t *deg, uint32_t *up, uint64_t *start, uint64_t *tile_scratch,
                                  int *score_range, hipStream_t s) {
    const int init[3] = {INT_MAX, INT_MIN, 0};
    hipError_t e = hipMemcpyAsync(score_range, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_edge_degree, dim3(512, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, deg, up,
                       symmetric ? 1 : 0, score_range, n);
    launch_scan<uint64_t>(deg, start, n, tile_scratch, nullptr, s);
    return hipGetLastError();
}

hipError_t launch_csr_scatter(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts,
                              bool symmetric, const uint64_t *start, const uint32_t *up, uint32_t *cursor, void *adj,
                              bool packed, int base, uint32_t n, hipStream_t s) {
    if (packed)
        hipLaunchKernelGGL((k_edge_scatter<NbrPacked>), dim3(512, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard,
                           counts, start, up, cursor, (NbrPacked *)adj, symmetric ? 1 : 0, base, n);
    else
        hipLaunchKernelGGL((k_edge_scatter<Nbr>), dim3(512, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts,
                           start, up, cursor, (Nbr *)adj, symmetric ? 1 : 0, base, n);
    return hipGetLastError();
}

hipError_t launch_compact_edges(const uint64_t *edges, uint64_t cap_per_shard, const unsigned long long *counts,
                                uint64_t *out, uint64_t out_capacity, unsigned long long *total, hipStream_t s) {
    hipLaunchKernelGGL(k_compact_edges, dim3(128, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, out,
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
    hipLaunchKernelGGL(k_rows_degree, dim3(128, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, deg, misfit,
                       threshold);
    launch_scan<uint32_t>(deg, row_start, n, tile_scratch, misfit, s);
    hipLaunchKernelGGL(k_rows_scatter, dim3(128, HMK_EDGE_SHARDS), dim3(256), 0, s, edges, cap_per_shard, counts, row_start,
                       cursor, adj, adj_capacity, threshold);
    return hipGetLastError();
}

size_t pack_rows_scratch_bytes(uint32_t n) { return ((size_t)2 * n + 2) * 4 + scan_scratch_bytes(n); }
size_t scan_scratch_bytes(uint32_t n) { return ((size_t)(n + SCAN_TILE - 1) / SCAN_TILE + 1) * 8; }

hipError_t launch_unpack_rows(const uint32_t *row_start, const uint32_t *adj, uint32_t n, int threshold, uint64_t *out,
                              uint64_t out_capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_rows_unpack, dim3(1024), dim3(256), 0, s, row_start, adj, n, threshold, out, out_capacity);
    return hipGetLastError();
}

// counts (cand_cnt[nl]) or fills (cand at cand_start[q]) the candidate lists; adj is Nbr[] or NbrPacked[]
hipError_t launch_greedy_precheck(bool fill, bool packed, const uint64_t *start, const void *adj, const int32_t *cluster_of,
                                  const int32_t *usize, const uint32_t *leftover, uint32_t nl, uint32_t *cand_cnt,
                                  const uint32_t *cand_start, GreedyCand *cand, uint32_t *overflow, hipStream_t s) {
    if (nl == 0) return hipSuccess;
    const dim3 grid(std::min<uint32_t>((nl + 3) / 4, 256 * 12)), block(256);
#define HMK_PRE(T, F) hipLaunchKernelGGL((k_greedy_precheck<T, F>), grid, block, 0, s, start, (const T *)adj, cluster_of, usize, \
                                         leftover, nl, cand_cnt, cand_start, cand, overflow)
    if (packed) { if (fill) HMK_PRE(NbrPacked, true); else HMK_PRE(NbrPacked, false); }
    else { if (fill) HMK_PRE(Nbr, true); else HMK_PRE(Nbr, false); }
#undef HMK_PRE
    return hipGetLastError();
}

// exclusive scan of uint32 counts into uint32 start[n + 1] (tile_scratch: scan_scratch_bytes(n))
hipError_t launch_scan_u32(const uint32_t *counts, uint32_t *start, uint32_t n, uint64_t *tile_scratch, hipStream_t s) {
    launch_scan<uint32_t>(counts, start, n, tile_scratch, nullptr, s);
    return hipGetLastError();
}

}  // namespace hmk
