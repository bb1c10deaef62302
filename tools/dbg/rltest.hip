#include <hip/hip_runtime.h>
#include <cstdio>
// ---- scans over the lanes of a group with DPP row shifts / row broadcasts (no LDS traffic) --------
// scan_prev<L, STEP>(x, ident): the partner value of step STEP of an inclusive scan over groups of
// L lanes: steps 0..3 fetch lane-1,-2,-4,-8 inside a row of 16, step 4 the total of the previous
// row (lane 15), step 5 the total of the first half wavefront (lane 31); `ident` where there is no
// partner.  Combining cur (op) prev in this order gives an inclusive scan also for
// non-commutative associative operators.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_id(float x, float ident)
{
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(x), CTRL, ROWMASK, 0xF, false));
}
template <int L>
constexpr int scan_steps()
{
    return L == 4 ? 2 : L == 8 ? 3 : L == 16 ? 4 : L == 32 ? 5 : 6;
}
template <int L, int STEP>
__device__ __forceinline__ float scan_prev(float x, float ident, int j)
{
    if constexpr (STEP == 0) {
        const float v = dpp_id<0x111, 0xF>(x, ident);
        return (L >= 16) ? v : ((j >= 1) ? v : ident);
    } else if constexpr (STEP == 1) {
        const float v = dpp_id<0x112, 0xF>(x, ident);
        return (L >= 16) ? v : ((j >= 2) ? v : ident);
    } else if constexpr (STEP == 2) {
        const float v = dpp_id<0x114, 0xF>(x, ident);
        return (L >= 16) ? v : ((j >= 4) ? v : ident);
    } else if constexpr (STEP == 3) {
        return dpp_id<0x118, 0xF>(x, ident);
    } else if constexpr (STEP == 4) {
        return dpp_id<0x142, 0xA>(x, ident); // row_bcast:15 into rows 1 and 3
    } else {
        return dpp_id<0x143, 0xC>(x, ident); // row_bcast:31 into rows 2 and 3
    }
}
template <int L, int STEP = 0>
__device__ __forceinline__ float prefix_sum(float x, int j)
{
    if constexpr (STEP < scan_steps<L>()) {
        x += scan_prev<L, STEP>(x, 0.0f, j);
        return prefix_sum<L, STEP + 1>(x, j);
    } else {
        return x;
    }
}
// value of the previous / next lane of the group, `fill` at the group edge
template <int L>
__device__ __forceinline__ float shift_up1(float x, float fill, int j)
{
    const float v = dpp_id<0x138, 0xF>(x, fill); // wave_shr:1
    return (j == 0) ? fill : v;
}
template <int L>
__device__ __forceinline__ float shift_down1(float x, float fill, int j)
{
    const float v = dpp_id<0x130, 0xF>(x, fill); // wave_shl:1
    return (j == L - 1) ? fill : v;
}
// value held by the last lane of the group
template <int L>
__device__ __forceinline__ float group_last(float v)
{
    return __shfl(v, L - 1, L);
}
template <int L>
__device__ __forceinline__ float suffix_sum(float x, int j)
{
    const float pre = prefix_sum<L>(x, j);
    return group_last<L>(pre) - pre + x;
}

// closed-loop stage map dx+ = M dx + c, composed over the lanes of a group
struct AffineMap {
    float m00, m01, m02, m10, m11, m12, m20, m21, m22, c0, c1, c2;
};
template <int L, int STEP = 0>
__device__ __forceinline__ void scan_maps(AffineMap& f, int j)
{
    if constexpr (STEP < scan_steps<L>()) {
        AffineMap g; // the partner map (identity where there is none)
        g.m00 = scan_prev<L, STEP>(f.m00, 1.0f, j); g.m01 = scan_prev<L, STEP>(f.m01, 0.0f, j);
        g.m02 = scan_prev<L, STEP>(f.m02, 0.0f, j); g.m10 = scan_prev<L, STEP>(f.m10, 0.0f, j);
        g.m11 = scan_prev<L, STEP>(f.m11, 1.0f, j); g.m12 = scan_prev<L, STEP>(f.m12, 0.0f, j);
        g.m20 = scan_prev<L, STEP>(f.m20, 0.0f, j); g.m21 = scan_prev<L, STEP>(f.m21, 0.0f, j);
        g.m22 = scan_prev<L, STEP>(f.m22, 1.0f, j);
        g.c0 = scan_prev<L, STEP>(f.c0, 0.0f, j); g.c1 = scan_prev<L, STEP>(f.c1, 0.0f, j);
        g.c2 = scan_prev<L, STEP>(f.c2, 0.0f, j);
        AffineMap n; // f o g
        n.m00 = f.m00 * g.m00 + f.m01 * g.m10 + f.m02 * g.m20;
        n.m01 = f.m00 * g.m01 + f.m01 * g.m11 + f.m02 * g.m21;
        n.m02 = f.m00 * g.m02 + f.m01 * g.m12 + f.m02 * g.m22;
        n.m10 = f.m10 * g.m00 + f.m11 * g.m10 + f.m12 * g.m20;
        n.m11 = f.m10 * g.m01 + f.m11 * g.m11 + f.m12 * g.m21;
        n.m12 = f.m10 * g.m02 + f.m11 * g.m12 + f.m12 * g.m22;
        n.m20 = f.m20 * g.m00 + f.m21 * g.m10 + f.m22 * g.m20;
        n.m21 = f.m20 * g.m01 + f.m21 * g.m11 + f.m22 * g.m21;
        n.m22 = f.m20 * g.m02 + f.m21 * g.m12 + f.m22 * g.m22;
        n.c0 = f.m00 * g.c0 + f.m01 * g.c1 + f.m02 * g.c2 + f.c0;
        n.c1 = f.m10 * g.c0 + f.m11 * g.c1 + f.m12 * g.c2 + f.c1;
        n.c2 = f.m20 * g.c0 + f.m21 * g.c1 + f.m22 * g.c2 + f.c2;
        f = n;
        scan_maps<L, STEP + 1>(f, j);
    }
}


template <int L>
__device__ __forceinline__ float group_last_rl(float v)
{
    if constexpr (L == 64) {
        float r; asm volatile("s_nop 4\n\tv_readlane_b32 %0, %1, 63\n\ts_nop 4" : "=s"(r) : "v"(v));
        return r;
    } else if constexpr (L == 32) {
        float a, b;
        asm volatile("s_nop 4\n\tv_readlane_b32 %0, %2, 31\n\tv_readlane_b32 %1, %2, 63\n\ts_nop 4" : "=s"(a), "=s"(b) : "v"(v));
        return (threadIdx.x < 32) ? a : b;
    } else {
        return __shfl(v, L - 1, L);
    }
}
template<int L, bool RL> __global__ void k(float* o, const float* in, int N)
{
    const int t = threadIdx.x, j = t % L;
    float cx0 = in[64], cx1 = in[65], cx2 = in[66];
    for (int base = 0; base < N; base += L) {
        const int kk = base + j; const bool inn = kk < N;
        AffineMap f; f.m00=1;f.m01=0;f.m02=0;f.m10=0;f.m11=1;f.m12=0;f.m20=0;f.m21=0;f.m22=1; f.c0=0; f.c1=0; f.c2=0;
        if (inn) { f.m02 = in[t]; f.c0 = 1.f; f.c1 = in[t] * 2.f; f.c2 = 0.25f; f.m00 = 1.f + 0.01f * in[t]; }
        scan_maps<L>(f, j);
        const float o0 = f.m00 * cx0 + f.m01 * cx1 + f.m02 * cx2 + f.c0;
        const float o1 = f.m10 * cx0 + f.m11 * cx1 + f.m12 * cx2 + f.c1;
        const float o2 = f.m20 * cx0 + f.m21 * cx1 + f.m22 * cx2 + f.c2;
        const float dx0 = shift_up1<L>(o0, cx0, j);
        if (inn) o[base + t] = dx0 + o1 * 1e-3f;
        if (RL) { cx0 = group_last_rl<L>(o0); cx1 = group_last_rl<L>(o1); cx2 = group_last_rl<L>(o2); }
        else { cx0 = group_last<L>(o0); cx1 = group_last<L>(o1); cx2 = group_last<L>(o2); }
    }
    if (j == 0) { o[200 + t] = cx0; o[201 + t] = cx1; o[202+t] = cx2; }
}
int main(){
    float hin[67]; for (int i=0;i<64;++i) hin[i] = 0.01f*(i%7) - 0.02f; hin[64]=0.3f; hin[65]=-0.2f; hin[66]=0.1f;
    float *din, *d1, *d2; hipMalloc(&din, sizeof hin); hipMalloc(&d1, 512*4); hipMalloc(&d2, 512*4);
    hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    hipMemset(d1, 0, 2048); hipMemset(d2, 0, 2048);
    k<64,false><<<1,64>>>(d1, din, 20); k<64,true><<<1,64>>>(d2, din, 20);
    float h1[512], h2[512]; hipMemcpy(h1,d1,2048,hipMemcpyDeviceToHost); hipMemcpy(h2,d2,2048,hipMemcpyDeviceToHost);
    double md=0; for(int i=0;i<512;++i) md = fmax(md, fabs(h1[i]-h2[i]));
    printf("L=64 maxdiff shfl vs readlane: %g  (cx: %g %g %g | %g %g %g)\n", md, h1[200],h1[201],h1[202],h2[200],h2[201],h2[202]);
    hipMemset(d1, 0, 2048); hipMemset(d2, 0, 2048);
    k<32,false><<<1,64>>>(d1, din, 20); k<32,true><<<1,64>>>(d2, din, 20);
    hipMemcpy(h1,d1,2048,hipMemcpyDeviceToHost); hipMemcpy(h2,d2,2048,hipMemcpyDeviceToHost);
    md=0; for(int i=0;i<512;++i) md = fmax(md, fabs(h1[i]-h2[i]));
    printf("L=32 maxdiff shfl vs readlane: %g  (cx: %g %g %g | %g %g %g)\n", md, h1[200],h1[201],h1[202],h2[200],h2[201],h2[202]);
    return 0; }
