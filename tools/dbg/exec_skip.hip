// Diagnostic: VALU issue rate of a lone wavefront vs. instruction-level parallelism (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void probe(float* out, long long* cyc)
{
    const int lane = threadIdx.x;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = lane * 0.001f + 1.0f + i;
    const float b = 0.999f, c = 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int k = 0; k < CH; ++k) a[k] = a[k] * b + c;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[lane] = s;
    if (lane == 0) *cyc = t1 - t0;
}
template <int CH>
void run(float* out, long long* cyc)
{
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<CH>, dim3(1), dim3(64), 0, 0, out, cyc); (void)hipDeviceSynchronize(); }
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d independent chains: %lld cycles for %d VALU -> %.2f cycles/instr\n", CH, h, 256 * 16, (double)h / (256 * 16));
}
int main()
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 4); (void)hipMalloc(&cyc, 8);
    run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<8>(out, cyc);
    return 0;
}
