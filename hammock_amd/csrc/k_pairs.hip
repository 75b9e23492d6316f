// k_pairs.hip -- one pair per lane, literal loops: the parity probes behind hmk_score_pairs_* and the
// literal tier of hmk_score_block_* (ShiftedScorer.java:48-95, LocalAlignmentScorer.java:27-86).
#include "hmk_device.h"

namespace hmk {

template <int SCORER>  // 0 shifted, 1 local
__global__ void __launch_bounds__(256)
k_pairs(const uint8_t *__restrict__ res32, const uint8_t *__restrict__ len, const int32_t *__restrict__ Mg,
        const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj, uint64_t n_pairs,
        uint32_t block_r0, uint32_t block_c0, uint32_t block_w,  // block mode when pi == nullptr
        int a, int b, int32_t *__restrict__ out, int32_t *__restrict__ out_shift) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *M = reinterpret_cast<int *>(smem);                                   // 576 dwords
    uint32_t *seqs = reinterpret_cast<uint32_t *>(smem + 2304);               // 256 * 2 * 9 dwords
    uint32_t *dp = seqs + 256 * 2 * SEQ_STRIDE_DW;                            // local: 33 * 256 dwords
    const int tid = threadIdx.x;
    for (int e = tid; e < 576; e += 256) M[e] = Mg[e];
    __syncthreads();
    uint32_t *s1 = seqs + tid * 2 * SEQ_STRIDE_DW;
    uint32_t *s2 = s1 + SEQ_STRIDE_DW;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + tid; k < n_pairs; k += (uint64_t)gridDim.x * 256) {
        uint32_t i, j;
        if (pi) { i = pi[k]; j = pj[k]; }
        else { i = block_r0 + (uint32_t)(k / block_w); j = block_c0 + (uint32_t)(k % block_w); }
        stage_sequence(s1, res32, i);
        stage_sequence(s2, res32, j);
        const int l1 = len[i], l2 = len[j];
        int score;
        if (SCORER == 0) {
            int shift = 0;
            score = shifted_score_literal(M, reinterpret_cast<const uint8_t *>(s1), l1,
                                          reinterpret_cast<const uint8_t *>(s2), l2, a, b, &shift);
            if (out_shift) out_shift[k] = shift;
        } else
            score = local_score_literal(M, reinterpret_cast<const uint8_t *>(s1), l1,
                                        reinterpret_cast<const uint8_t *>(s2), l2, a, b, dp + tid, 256);
        out[k] = score;
    }
}

// -----------------------------------------------------------------------------
// launcher
// -----------------------------------------------------------------------------
hipError_t launch_pairs(int scorer, const uint8_t *res32, const uint8_t *len, const int32_t *d_matrix,
                        const uint32_t *pi, const uint32_t *pj, uint64_t n_pairs, uint32_t r0, uint32_t c0,
                        uint32_t width, int a, int b, int32_t *out, int32_t *out_shift, hipStream_t s) {
    if (n_pairs == 0) return hipSuccess;
    uint64_t blocks = (n_pairs + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const size_t lds_shift = 2304 + 256 * 2 * SEQ_STRIDE_DW * 4;
    const size_t lds_local = lds_shift + 33 * 256 * 4;
    if (scorer == 0)
        hipLaunchKernelGGL(k_pairs<0>, dim3((uint32_t)blocks), dim3(256), lds_shift, s, res32, len, d_matrix, pi, pj,
                           n_pairs, r0, c0, width, a, b, out, out_shift);
    else
        hipLaunchKernelGGL(k_pairs<1>, dim3((uint32_t)blocks), dim3(256), lds_local, s, res32, len, d_matrix, pi, pj,
                           n_pairs, r0, c0, width, a, b, out, out_shift);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------
// do two streams run side by side?  (hmk_pass.cpp: the side streams of a mixed-length pass)
// -----------------------------------------------------------------------------
// One wave that notes when it started, spins for `ticks` of the 100 MHz wall clock and notes when it ended.
__global__ void k_probe_spin(unsigned long long *when, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) { when[0] = (unsigned long long)t0; when[1] = (unsigned long long)wall_clock64(); }
}
hipError_t launch_probe_spin(unsigned long long *when, long long ticks, hipStream_t s) {
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, s, when, ticks);
    return hipGetLastError();
}

}  // namespace hmk
