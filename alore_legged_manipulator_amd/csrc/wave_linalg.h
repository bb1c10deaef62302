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
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j == k) continue;
            a[j] = me ? rowk[j] * pk : a[j] - g * rowk[j];
        }
        a[k] = me ? pk : -g;
    }
}

} // namespace wavela
