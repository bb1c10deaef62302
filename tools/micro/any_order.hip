// Does hipExtAnyOrderLaunch let a kernel start beside its predecessor in the SAME stream on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/any_order.hip -o tools/micro/any_order && tools/micro/any_order
// Three kernels on one stream: spin(A), spin(B) [flag under test], tiny(C); wall time of the trio from events and the
// start / end stamps (100 MHz counter) each kernel leaves.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(unsigned long long* out, int slot, long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = __builtin_amdgcn_s_memrealtime(); }
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int flags = 0; flags <= 1; ++flags)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d, 0, 64, s);
            hipEventRecord(e0, s);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, d, 0, 2000LL);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, d, 1, 2000LL);
            hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d, 2, 10LL);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[6];
            hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
            printf("flag %d: trio %.1f us by events; A %.1f..%.1f  B %.1f..%.1f  C %.1f..%.1f us (from A's start)\n", flags, ms * 1e3, 0.0,
                   (h[1] - h[0]) * 0.01, (double)(long long)(h[2] - h[0]) * 0.01, (double)(long long)(h[3] - h[0]) * 0.01,
                   (double)(long long)(h[4] - h[0]) * 0.01, (double)(long long)(h[5] - h[0]) * 0.01);
        }
    return 0;
}
