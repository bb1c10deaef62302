// What does a cross-stream dependency cost the chain of a tick?  Stream A: K1 (15 us) -> K2 (4 us) per iteration; stream B: K3
// (10 us) beside K1.  K2 needs K3 of the same iteration (A waits on B's event), K3 must not start before K1 of the previous
// iteration is over (B waits on A's event).  Per-iteration time for: no streams at all (K3 K1 K2 in order), no waits (unsafe),
// only B waits on A, both waits; events default / disable-timing / + release-to-device.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cross_stream.hip -o tools/micro/cross_stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long long ticks, float* buf)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float v = buf[blockIdx.x * 64 + threadIdx.x];
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
    buf[blockIdx.x * 64 + threadIdx.x] = v + 1.f;
}

int main()
{
    float *b1, *b2, *b3;
    hipMalloc(&b1, 1024 * 64 * 4); hipMalloc(&b2, 64 * 64 * 4); hipMalloc(&b3, 2048 * 64 * 4);
    hipMemset(b1, 0, 1024 * 64 * 4); hipMemset(b2, 0, 64 * 64 * 4); hipMemset(b3, 0, 2048 * 64 * 4);
    hipStream_t A, B;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    const unsigned flagsets[3] = {hipEventDefault, hipEventDisableTiming, hipEventDisableTiming | hipEventReleaseToDevice};
    const char* fname[3] = {"default", "disable-timing", "disable-timing+release-to-device"};
    const int n = 300;
    for (int fs = 0; fs < 3; ++fs) {
        hipEvent_t ea[2], eb[2];
        for (int i = 0; i < 2; ++i) { hipEventCreateWithFlags(&ea[i], flagsets[fs]); hipEventCreateWithFlags(&eb[i], flagsets[fs]); }
        for (int mode = 0; mode < 4; ++mode) {
            if (fs > 0 && mode < 2) continue;
            for (int rep = 0; rep < 2; ++rep) {
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; ++i) {
                    if (mode == 0) {
                        spin<<<2048, 64, 0, A>>>(1000, b3);
                    } else {
                        if (mode >= 2 && i > 0) hipStreamWaitEvent(B, ea[(i - 1) & 1], 0);
                        spin<<<2048, 64, 0, B>>>(1000, b3);
                        if (mode == 3) hipEventRecord(eb[i & 1], B);
                    }
                    spin<<<1024, 64, 0, A>>>(1500, b1);
                    if (mode >= 2) hipEventRecord(ea[i & 1], A);
                    if (mode == 3) hipStreamWaitEvent(A, eb[i & 1], 0);
                    spin<<<64, 64, 0, A>>>(400, b2);
                }
                const auto t1 = std::chrono::steady_clock::now();
                hipDeviceSynchronize();
                const auto t2 = std::chrono::steady_clock::now();
                if (rep == 1)
                    printf("events %-34s mode %d (%s): %.2f us per iteration (host enqueue %.2f)\n", fname[fs], mode,
                           mode == 0 ? "one stream, in order" : mode == 1 ? "two streams, no waits" : mode == 2 ? "B waits on A" : "B waits on A, A waits on B",
                           std::chrono::duration<double, std::micro>(t2 - t0).count() / n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
            }
        }
    }
    return 0;
}
