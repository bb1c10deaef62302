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

// 1 / x for a pivot: the hardware reciprocal (1 ulp) for float32 -- every caller refines or tolerates 1e-7 -- and the
// IEEE division for float64
__device__ __forceinline__ float pivot_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double pivot_rcp(double x) { return 1.0 / x; }

// a[0..N-1] = row `lane` of A  ->  row `lane` of A^-1.  Must be called by the whole wavefront (lanes >= N carry junk).
// The scaling of a pivot row by 1 / pivot is DEFERRED to the end: every later operation on that row is linear in it, so
// the row can stay unscaled (its identity entry is 1 instead of 1 / pivot) and is multiplied once, after the last step --
// one fused multiply-add per element and step instead of two operations.
template <class T, int N>
__device__ __forceinline__ void spd_inverse_rows(T (&a)[N], int lane)
{
    T mine = T(1);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T rowk[N];
#pragma unroll
        for (int j = 0; j < N; ++j) rowk[j] = bcast(a[j], k);
        const T pk = pivot_rcp(rowk[k]);
        int lk = lane;
        asm volatile("" : "+v"(lk)); // opaque: keeps the N lane masks from being computed up front and held in (spilled) scalar registers
        const bool me = lk == k;
        const T g = me ? T(0) : a[k] * pk; // the pivot row itself is left as it is
        if (me) mine = pk;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j == k) continue;
            a[j] -= g * rowk[j];
        }
        a[k] = me ? T(1) : -g;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] *= mine;
}

// Measured alternatives for the 30 x 30 float32 case inside the Riccati kernel (three workgroups per CU, the other
// wavefronts busy with LDS-fed matrix tiles), all dropped:
//   * a row on two lanes (15 columns each), pivot row by ds_bpermute: half the arithmetic, but 16 permutes per step at
//     ~34 cycles each next to the tile traffic -- 16 k cycles against 14 k for this form; the same with 2 x 2 pivot
//     blocks (15 steps): the same 480 permutes, the same 16 k cycles;
//   * the pivot row through LDS memory (16-byte stores by the owner, broadcast loads): an LDS round trip per step in the
//     dependent chain, 27 k cycles;
//   * s_setprio around the elimination: 1.5 k cycles faster itself, the other phases of the three workgroups lose more.
} // namespace wavela
