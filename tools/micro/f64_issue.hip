// Lone-wavefront issue cost of float64 / float32 vector FMAs on gfx950: cycles per instruction with 1, 2, 4, 8 independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o f64_issue f64_issue.hip && ./f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int CH>
__global__ void chains(T* out, long long* cyc, T a, T b, int iters)
{
    T v[CH];
    for (int c = 0; c < CH; ++c) v[c] = (T)threadIdx.x + (T)c;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) v[c] = __builtin_fma(v[c], a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    T s = 0;
    for (int c = 0; c < CH; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename T, int CH>
void run(const char* name, int waves_per_block)
{
    T* out; long long* cyc;
    hipMalloc(&out, 1024 * sizeof(T)); hipMalloc(&cyc, 8 * sizeof(long long));
    const int iters = 2000;
    chains<T, CH><<<1, 64 * waves_per_block>>>(out, cyc, (T)0.999, (T)0.001, iters);
    hipDeviceSynchronize();
    chains<T, CH><<<1, 64 * waves_per_block>>>(out, cyc, (T)0.999, (T)0.001, iters);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s chains=%d waves/block=%d: %.2f cycles per FMA (per wavefront)\n", name, CH, waves_per_block, (double)h / (iters * 16.0 * CH));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<double, 1>("f64", 1); run<double, 2>("f64", 1); run<double, 4>("f64", 1); run<double, 8>("f64", 1);
    run<float, 1>("f32", 1); run<float, 2>("f32", 1); run<float, 4>("f32", 1); run<float, 8>("f32", 1);
    run<double, 8>("f64", 4); run<double, 8>("f64", 8); run<float, 8>("f32", 4); run<float, 8>("f32", 8);
    run<double, 1>("f64", 8); run<float, 1>("f32", 8);
    return 0;
}
