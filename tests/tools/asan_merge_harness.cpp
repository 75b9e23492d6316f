// ASAN/UBSAN harness for the host merge: random thresholded graphs -> greedy_from_edges (both code paths)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "hmk_internal.h"
int main() {
    std::mt19937_64 rng(7);
    for (int trial = 0; trial < 30; trial++) {
        const uint32_t n = 50 + (uint32_t)(rng() % 6000);
        const double dens = (trial % 3 == 0) ? 0.05 : 0.004;
        std::vector<uint64_t> edges;
        for (uint32_t x = 0; x < n; x++)
            for (uint32_t m = x + 1; m < n; m++)
                if ((rng() % 1000000) < dens * 1000000) edges.push_back(((uint64_t)x << 40) | ((uint64_t)m << 16) | (uint64_t)(20 + rng() % 30));
        std::vector<int32_t> sizes(n), cid(n), order(n), rank(n);
        for (auto &s : sizes) s = 1 + (int32_t)(rng() % 5);
        hmk_greedy_stats st;
        std::string err;
        const int mc = (int)(n / 40) + (trial % 4 == 0 ? 0 : 1);
        hmk::GreedyOptions opt;
        if (const char *t = getenv("HMK_PHASE1_THREADS")) opt.phase1_threads = atoi(t);
        if (const char *w = getenv("HMK_PHASE1_WINDOW")) opt.phase1_window = atoi(w);
        if (trial % 3 != 2) {   // phase 1 on a prepared band (built on the host): bands that end before, inside and after phase 1
            opt.host_band_rows = (int)(1 + rng() % (n + 10));
            opt.host_band_far_t = 1 + (int)(rng() % 5);
        }
        int rc = hmk::greedy_from_edges(n, trial % 2 ? sizes.data() : nullptr, edges.data(), edges.size(), trial % 5 != 4, 20, mc,
                                        cid.data(), order.data(), rank.data(), &st, &err, opt);
        printf("trial %d n %u edges %zu rc %d clusters %d result %d\n", trial, n, edges.size(), rc, st.n_multi, st.n_result_clusters);
    }
    return 0;
}
