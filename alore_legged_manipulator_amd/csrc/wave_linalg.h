// wave_linalg.h -- small dense SPD inverse inside ONE wavefront, registers + cross-lane broadcasts only.
//
// Lane i (< N) holds row i of A in N registers.  In-place Gauss-Jordan without pivoting (A symmetric positive
// definite: every pivot is positive, growth factor 1): step k broadcasts pivot row k to the whole wavefront with
// v_readlane (the results are wavefront-uniform, i.e. scalar registers) and every lane updates its own row with N
// fused multiply-adds.  N steps x (N broadcasts + N FMAs): no LDS traffic, no barriers, and no loop-carried LDS
// latency -- the triangular substitutions this replaces spent their time waiting ~100 cycles per dependent LDS read.
#pragma once
#include <hip/hip_runtime.h>

namespace wavela {

__device__ __forceinline__ float bcast(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ double bcast(double x, int lane)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// a[0..N-1] = row `lane` of A  ->  row `lane` of A^-1.  Must be called by the whole wavefront (lanes >= N carry junk).
template <class T, int N>
__device__ __forceinline__ void spd_inverse_rows(T (&a)[N], int lane)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T rowk[N];
#pragma unroll
        for (int j = 0; j < N; ++j) rowk[j] = bcast(a[j], k);
        const T pk = T(1) / rowk[k];
        const bool me = lane == k;
        const T g = a[k] * pk;
        // two operations per element and lane: every other row subtracts g x (pivot row); the pivot row itself (g_eff = 0
        // leaves it untouched) is scaled by 1 / pivot
        const T g_eff = me ? T(0) : g, sc = me ? pk : T(1);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j == k) continue;
            a[j] = (a[j] - g_eff * rowk[j]) * sc;
        }
        a[k] = me ? pk : -g;
    }
}

// The same elimination with a row spread over TWO lanes: lane r (< N) holds columns 0 .. N/2 - 1 of row r, lane r + 32
// columns N/2 .. N - 1 (N even, N <= 32).  A step moves N/2 pivot-row entries per lane through the cross-lane permute
// (each half fetches its own half of the pivot row) plus the lane's pivot-column entry, and updates N/2 elements: less
// than half the instructions of the one-lane-per-row form, which spends its time issuing N broadcasts and 2 N
// arithmetic instructions per step on one wavefront.
__device__ __forceinline__ float fetch(float x, int src) { return __shfl(x, src, 64); }
__device__ __forceinline__ double fetch(double x, int src) { return __shfl(x, src, 64); }

template <class T, int N>
__device__ __forceinline__ void spd_inverse_rows_split(T (&a)[N / 2], int lane)
{
    static_assert(N % 2 == 0 && N <= 32, "row halves on lanes r and r + 32");
    constexpr int H = N / 2;
    const int r = lane & 31, half = lane >> 5;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        constexpr int dummy = 0; (void)dummy;
        const int hk = k / H, kk = k % H;                 // compile-time after unrolling
        T rowk[H];
#pragma unroll
        for (int j = 0; j < H; ++j) rowk[j] = fetch(a[j], k + 32 * half);   // my half of pivot row k
        const T piv = bcast(a[kk], k + 32 * hk);          // A[k][k], wave-uniform
        const T pk = T(1) / piv;
        const T f = fetch(a[kk], r + 32 * hk);            // A[r][k] of my row (held by the half that owns column k)
        const bool me = r == k;
        const T g = f * pk;
        const T g_eff = me ? T(0) : g, sc = me ? pk : T(1);
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const T upd = (a[j] - g_eff * rowk[j]) * sc;
            a[j] = (j == kk && half == hk) ? (me ? pk : -g) : upd;
        }
    }
}

} // namespace wavela
