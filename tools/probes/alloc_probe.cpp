// How long do the big allocations of a first clustering call take on a fresh process?  (hipMalloc of 12 GB buffers, a 0.8 GB
// pinned host buffer), one after the other and on parallel threads.
//   hipcc -O2 -o /tmp/alloc_probe tools/probes/alloc_probe.cpp && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const bool parallel = argc > 1 && argv[1][0] == 'p';
    double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    std::printf("init %.1f ms\n", now() - t0);
    const size_t GB = 1ull << 30;
    void *d[3] = {nullptr, nullptr, nullptr};
    void *h = nullptr;
    t0 = now();
    if (parallel) {
        std::vector<std::thread> pool;
        for (int k = 0; k < 3; k++) pool.emplace_back([&, k] { double t = now(); hipSetDevice(0); hipError_t e = hipMalloc(&d[k], 12 * GB); std::printf("  hipMalloc 12 GB [%d] %.1f ms (%d)\n", k, now() - t, (int)e); });
        pool.emplace_back([&] { double t = now(); hipError_t e = hipHostMalloc(&h, GB * 8 / 10, 0); std::printf("  hipHostMalloc 0.8 GB %.1f ms (%d)\n", now() - t, (int)e); });
        for (auto &th : pool) th.join();
    } else {
        for (int k = 0; k < 3; k++) { double t = now(); hipError_t e = hipMalloc(&d[k], 12 * GB); std::printf("  hipMalloc 12 GB [%d] %.1f ms (%d)\n", k, now() - t, (int)e); }
        double t = now(); hipError_t e = hipHostMalloc(&h, GB * 8 / 10, 0); std::printf("  hipHostMalloc 0.8 GB %.1f ms (%d)\n", now() - t, (int)e);
    }
    std::printf("all allocations %.1f ms\n", now() - t0);
    t0 = now();
    hipMemsetAsync(d[0], 0, 12 * GB, 0);
    hipStreamSynchronize(0);
    std::printf("first touch (memset 12 GB) %.1f ms\n", now() - t0);
    t0 = now();
    hipMemsetAsync(d[0], 0, 12 * GB, 0);
    hipStreamSynchronize(0);
    std::printf("second memset 12 GB %.1f ms\n", now() - t0);
    std::fflush(stdout);
    if (argc > 2) { t0 = now(); for (int k = 0; k < 3; k++) hipFree(d[k]); hipHostFree(h); std::printf("free %.1f ms\n", now() - t0); }
    std::fflush(stdout);
    _Exit(0);
}
