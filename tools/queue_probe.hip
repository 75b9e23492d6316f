// tools/queue_probe.hip -- which streams of a process share a hardware queue?  hipcc --offload-arch=gfx950 -O2 -o /tmp/queue_probe tools/queue_probe.hip && /tmp/queue_probe 10 [0|1|2]
// (0: ten fresh streams; 1: half of them destroyed and created again first; 2: three priorities).  Result on MI355X / ROCm 7.2: four queues per priority, ten fresh
// streams land on a b c d d c b a d c; GPU_MAX_HW_QUEUES=8 gives eight.  What hmk_pass.cpp make_side_streams is built on.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;   // 1: destroy and re-create half of them first; 2: priorities
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    std::vector<hipStream_t> s(N);
    for (int i = 0; i < N; i++) {
        if (mode == 2) hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i % 3 == 0 ? hi : i % 3 == 1 ? 0 : lo);
        else hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    }
    if (mode == 1) { for (int i = 0; i < N; i += 2) { hipStreamDestroy(s[i]); } for (int i = 0; i < N; i += 2) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); }
    const long long T = 20000;   // 100 MHz wall clock: 200 us
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 100); }
    hipDeviceSynchronize();
    auto pair_ms = [&](hipStream_t a, hipStream_t b) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, T);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T);
        hipStreamSynchronize(a); hipStreamSynchronize(b);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    printf("    ");
    for (int j = 0; j < N; j++) printf("%5d", j);
    printf("   (S = the pair serialises)\n");
    for (int i = 0; i < N; i++) {
        printf("%3d ", i);
        for (int j = 0; j < N; j++) { if (j <= i) { printf("    ."); continue; } const double ms = pair_ms(s[i], s[j]); printf("%5s", ms > 0.33 ? "S" : "-"); }
        printf("\n");
    }
    // null stream against each
    printf("null");
    for (int j = 0; j < N; j++) { const double ms = pair_ms(nullptr, s[j]); printf("%5s", ms > 0.33 ? "S" : "-"); }
    printf("\n");
    return 0;
}
