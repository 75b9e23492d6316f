// micro-benchmark of the seed walk's filter pass on the host CPU (scratch; not part of the product)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <cstring>
#include <immintrin.h>
static uint64_t rng_s = 88172645463325252ull;
static inline uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 11); }
int main(int argc, char **argv) {
    const uint32_t R1 = 51024, STEPS = 25000, LEN = argc > 1 ? atoi(argv[1]) : 260, NUP = 207;
    const bool cold = argc > 2 && atoi(argv[2]);
    // lists laid out at random places of a big block (as the pack's tr lists)
    const size_t block = (size_t)64 << 20;   // 256 MB of uint32
    uint32_t *tr = (uint32_t *)aligned_alloc(4096, block * 4);
    for (size_t i = 0; i < block; i++) tr[i] = (rnd() % R1) << 8 | (rnd() & 0xFF);
    std::vector<uint32_t> where(STEPS), up((size_t)STEPS * NUP);
    { const char *sq = getenv("SEQ"); uint32_t kk = 0; for (auto &w : where) { w = sq ? kk * (uint32_t)atoi(sq) : rnd() % (block - 4096); kk++; } }
    for (auto &u : up) u = (rnd() % R1) << 8 | (rnd() & 0xFF);
    std::vector<uint32_t> stamp(R1, 0), picked(4096);
    std::vector<uint64_t> bits((R1 + 63) / 64, 0);
    std::vector<uint8_t> state(1000000, 0);
    for (uint32_t i = 0; i < R1; i++) state[i] = (rnd() % 10) == 0;
    for (int variant = 0; variant < 4; variant++) {
        uint64_t total = 0, t_stamp = 0, t_p1 = 0, t_near = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t k = 0; k < STEPS; k++) {
            const uint32_t step = k + 1;
            const uint32_t *row = &up[(size_t)k * NUP];
            uint64_t a = __builtin_ia32_rdtsc();
            // near candidates
            int32_t bar = 0, best = -1;
            for (uint32_t q = 0; q < NUP; q++) {
                const uint32_t m = row[q] >> 8;
                const int32_t s = row[q] & 0xFF, sf = state[m] == 0 ? s : -1;
                if (sf >= bar) { if (s > bar || best < 0 || (int32_t)m < best) { best = m; bar = s; } }
            }
            total += best;
            uint64_t b = __builtin_ia32_rdtsc();
            t_near += b - a;
            if (variant == 1 || variant == 3) { for (uint32_t q = 0; q < NUP; q++) { const uint32_t x = row[q] >> 8; bits[x >> 6] |= 1ull << (x & 63); } }
            else for (uint32_t q = 0; q < NUP; q++) stamp[row[q] >> 8] = step << 8 | (row[q] & 0xFFu);
            uint64_t c = __builtin_ia32_rdtsc();
            t_stamp += c - b;
            const uint32_t *bl = tr + where[k];
            uint32_t np = 0;
            if (variant == 0) {
                for (uint32_t q = 0; q < LEN; q++) {
                    const uint32_t x = bl[q] >> 8, xs = x < R1 ? x : 0;
                    picked[np] = q;
                    np += (uint32_t)((xs > k) & ((stamp[xs] >> 8) == step));
                }
            } else if (variant == 1) {
                for (uint32_t q = 0; q < LEN; q++) {
                    const uint32_t x = bl[q] >> 8, xs = x < R1 ? x : 0;
                    picked[np] = q;
                    np += (uint32_t)((xs > k) & ((bits[xs >> 6] >> (xs & 63)) & 1));
                }
            } else if (variant == 2) {
                // AVX-512: 16 entries at a time, gather of the stamps
                const __m512i vk = _mm512_set1_epi32((int)k), vstep = _mm512_set1_epi32((int)step), vr1 = _mm512_set1_epi32((int)R1);
                uint32_t q = 0;
                for (; q + 16 <= LEN; q += 16) {
                    const __m512i e = _mm512_loadu_si512(bl + q), x = _mm512_srli_epi32(e, 8);
                    const __mmask16 in = _mm512_cmplt_epu32_mask(x, vr1) & _mm512_cmpgt_epu32_mask(x, vk);
                    const __m512i sw = _mm512_mask_i32gather_epi32(_mm512_setzero_si512(), in, x, stamp.data(), 4);
                    const __mmask16 ok = in & _mm512_cmpeq_epu32_mask(_mm512_srli_epi32(sw, 8), vstep);
                    _mm512_mask_compressstoreu_epi32(&picked[np], ok, _mm512_add_epi32(_mm512_set1_epi32((int)q), _mm512_setr_epi32(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15)));
                    np += (uint32_t)__builtin_popcount(ok);
                }
                for (; q < LEN; q++) { const uint32_t x = bl[q] >> 8, xs = x < R1 ? x : 0; picked[np] = q; np += (uint32_t)((xs > k) & ((stamp[xs] >> 8) == step)); }
            } else {
                const __m512i vk = _mm512_set1_epi32((int)k), vr1 = _mm512_set1_epi32((int)R1);
                uint32_t q = 0;
                for (; q + 16 <= LEN; q += 16) {
                    const __m512i e = _mm512_loadu_si512(bl + q), x = _mm512_srli_epi32(e, 8);
                    const __mmask16 in = _mm512_cmplt_epu32_mask(x, vr1) & _mm512_cmpgt_epu32_mask(x, vk);
                    const __m512i w = _mm512_mask_i32gather_epi32(_mm512_setzero_si512(), in, _mm512_srli_epi32(x, 5), (const int *)bits.data(), 4);
                    const __mmask16 ok = in & _mm512_test_epi32_mask(_mm512_srlv_epi32(w, _mm512_and_epi32(x, _mm512_set1_epi32(31))), _mm512_set1_epi32(1));
                    _mm512_mask_compressstoreu_epi32(&picked[np], ok, _mm512_add_epi32(_mm512_set1_epi32((int)q), _mm512_setr_epi32(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15)));
                    np += (uint32_t)__builtin_popcount(ok);
                }
            }
            total += np;
            uint64_t d = __builtin_ia32_rdtsc();
            t_p1 += d - c;
            if (variant == 1 || variant == 3) for (uint32_t q = 0; q < NUP; q++) { const uint32_t x = row[q] >> 8; bits[x >> 6] = 0; }
            if (cold) { for (uint32_t i = 0; i < 64; i++) __builtin_ia32_clflush(bl + 16 * i); }
        }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("variant %d: %.2f ms total; near %.1f stamps %.1f filter %.1f Mticks (%.2f ticks per list entry) chk %llu\n", variant, ms, t_near / 1e6, t_stamp / 1e6, t_p1 / 1e6,
               (double)t_p1 / ((double)STEPS * LEN), (unsigned long long)total);
    }
    return 0;
}
