// Lone-wavefront cost of a DEPENDENT chain of float64 wavefront sums on gfx950 (what one pair of the L-BFGS two-loop recursion
// in backend_kernels.hip pays): cycles per  p = s * d -> sum over the wavefront -> a = sum / ys -> d += -a * y  step, for several
// ways of taking the sum.
//   hipcc --offload-arch=gfx950 -O3 -o wave_sum_f64 wave_sum_f64.hip && ./wave_sum_f64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ __forceinline__ double dpp64(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// V0: four row shifts + row_bcast15 + row_bcast31, read lane 63 (the kernel's wave_sum)
__device__ __forceinline__ double sum_v0(double v)
{
    v += dpp64<0x111>(v); v += dpp64<0x112>(v); v += dpp64<0x114>(v); v += dpp64<0x118>(v);
    v += dpp64<0x142>(v); v += dpp64<0x143>(v);
    return rl(v, 63);
}
// V1: four row shifts, the four row totals through scalar registers
__device__ __forceinline__ double sum_v1(double v)
{
    v += dpp64<0x111>(v); v += dpp64<0x112>(v); v += dpp64<0x114>(v); v += dpp64<0x118>(v);
    const double r0 = rl(v, 15), r1 = rl(v, 31), r2 = rl(v, 47), r3 = rl(v, 63);
    return (r3 + r2) + (r1 + r0);
}
// V2: butterflies inside the row (quad_perm, row_half_mirror, row_mirror) then scalar combination
__device__ __forceinline__ double sum_v2(double v)
{
    v += dpp64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp64<0x141>(v); // row_half_mirror
    v += dpp64<0x140>(v); // row_mirror
    const double r0 = rl(v, 0), r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
    return (r3 + r2) + (r1 + r0);
}
// V3: only the four in-row steps (lower bound of the DPP part)
__device__ __forceinline__ double sum_v3(double v)
{
    v += dpp64<0x111>(v); v += dpp64<0x112>(v); v += dpp64<0x114>(v); v += dpp64<0x118>(v);
    return rl(v, 63);
}
// V4: through LDS: every lane writes, 64 values summed by a tree of reads (two levels of 8)
__device__ __forceinline__ double sum_v4(double v, double* sh)
{
    sh[threadIdx.x] = v;
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += sh[(threadIdx.x & 7) * 8 + i];
    // lanes 0..7 hold partial sums of 8: combine through DPP row shifts over 8 lanes
    t += dpp64<0x111>(t); t += dpp64<0x112>(t); t += dpp64<0x114>(t);
    return rl(t, 7);
}

template <int V>
__global__ void chain(double* out, long long* cyc, const double* sv, const double* yv, double ys, double rys, int iters)
{
    __shared__ double sh[64];
    double d = 1.0 + 1e-3 * threadIdx.x;
    const double s = sv[threadIdx.x], y = yv[threadIdx.x];
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double p = s * d;
            double sum;
            if (V == 0) sum = sum_v0(p);
            else if (V == 1) sum = sum_v1(p);
            else if (V == 2) sum = sum_v2(p);
            else if (V == 3) sum = sum_v3(p);
            else sum = sum_v4(p, sh);
            const double q0 = sum * rys;
            const double q1 = fma(fma(-ys, q0, sum), rys, q0);
            const double a = fma(fma(-ys, q1, sum), rys, q1);
            d += (-a) * y;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// cost of a dependent chain of one kind of DPP move + add
template <int CTRL>
__global__ void dpp_chain(double* out, long long* cyc, int iters)
{
    double v = 1.0 + threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v = v * 0.5 + dpp64<CTRL>(v);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *out, *sv, *yv; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&sv, 64 * 8); hipMalloc(&yv, 64 * 8); hipMalloc(&cyc, 8);
    double h[64];
    for (int i = 0; i < 64; ++i) h[i] = 1e-3 * (i + 1);
    hipMemcpy(sv, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(yv, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 500;
    long long c;
#define RUN(V, name)                                                                                                      \
    chain<V><<<1, 64>>>(out, cyc, sv, yv, 3.0, 1.0 / 3.0, iters); hipDeviceSynchronize();                                  \
    chain<V><<<1, 64>>>(out, cyc, sv, yv, 3.0, 1.0 / 3.0, iters); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);            \
    printf("%-60s %.1f cycles per pair step\n", name, (double)c / (iters * 8.0));
    RUN(0, "V0 row_shr x4 + row_bcast15 + row_bcast31 + readlane 63");
    RUN(1, "V1 row_shr x4 + four row totals through readlane");
    RUN(2, "V2 quad_perm x2 + half_mirror + mirror + four readlanes");
    RUN(3, "V3 row_shr x4 only (not a full sum: lower bound)");
    RUN(4, "V4 through LDS");
#define DPP(C, name)                                                                                                      \
    dpp_chain<C><<<1, 64>>>(out, cyc, iters); hipDeviceSynchronize();                                                      \
    dpp_chain<C><<<1, 64>>>(out, cyc, iters); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                \
    printf("%-60s %.1f cycles per (2 dpp moves + fma) link\n", name, (double)c / (iters * 16.0));
    DPP(0x111, "row_shr:1");
    DPP(0x118, "row_shr:8");
    DPP(0xB1, "quad_perm");
    DPP(0x140, "row_mirror");
    DPP(0x142, "row_bcast:15");
    DPP(0x143, "row_bcast:31");
    return 0;
}
