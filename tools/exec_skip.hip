// Diagnostic: VALU issue rate on gfx950 -- one wavefront vs. several wavefronts per SIMD, dependent vs. independent
// FMA chains.  One workgroup on one CU; waves_per_simd * 4 wavefronts (blockDim = 256 * waves_per_simd).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void probe(float* out, long long* cyc)
{
    const int lane = threadIdx.x;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = lane * 0.001f + 1.0f + i;
    const float b = 0.999f, c = 0.5f;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 1024; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int k = 0; k < CH; ++k) a[k] = a[k] * b + c;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[lane] = s;
    if ((lane & 63) == 0) cyc[lane >> 6] = t1 - t0;
}
template <int CH>
void run(float* out, long long* cyc, int waves_per_simd)
{
    const int threads = 256 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<CH>, dim3(1), dim3(threads), 0, 0, out, cyc); (void)hipDeviceSynchronize(); }
    long long h[16]; (void)hipMemcpy(h, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    const double n = 1024.0 * 16;
    printf("%d chain(s), %d wave(s)/SIMD: %.2f cycles per instruction per wave -> %.2f cycles per instruction per SIMD\n", CH,
           waves_per_simd, mx / n, mx / n / waves_per_simd);
}
int main()
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&cyc, 8 * 16);
    for (int w : {1, 2, 3, 4}) { run<1>(out, cyc, w); run<2>(out, cyc, w); run<4>(out, cyc, w); run<8>(out, cyc, w); }
    return 0;
}
