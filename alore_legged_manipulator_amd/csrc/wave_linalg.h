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

// 16 x 16 float32 on ALL 64 lanes: lane 16 p + r holds columns 4 p .. 4 p + 3 of row r.  With the rows of the matrix on
// the lanes of a DPP row, the pivot row reaches every lane as a DPP operand (row_newbcast:k, gfx90a+: lane k of each
// 16-lane row, fused into the multiply-add: no scalar registers, no LDS); the lane's entry of the pivot column sits in
// quarter k / 4 and is spread to the other three quarters by v_permlane32_swap + v_permlane16_swap (gfx950).  A step is
// ~20 instructions instead of 16 v_readlane + 16 FMA.
template <int K>
__device__ __forceinline__ float row_bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xF, 0xF, false)); // row_newbcast:K
}
// x of DPP row P (lanes 16 P .. 16 P + 15) to the same position of all four rows
template <int P>
__device__ __forceinline__ float spread_row(float x)
{
    const unsigned u = __float_as_uint(x);
    const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // [0] = (R0, R1, R0, R1), [1] = (R2, R3, R2, R3)
    const unsigned hh = h[P >> 1];
    const auto q = __builtin_amdgcn_permlane16_swap(hh, hh, false, false); // [0] = even row of hh everywhere, [1] = odd row
    return __uint_as_float(q[P & 1]);
}
__device__ __forceinline__ void spd_inverse16_quarters(float (&a)[4], int lane)
{
    const int r = lane & 15, p = lane >> 4;
    float mine = 1.f;
#define WAVELA_STEP(K)                                                                                        \
    {                                                                                                         \
        constexpr int pk = (K) / 4, kk = (K) % 4;                                                             \
        const float r0 = row_bcast<K>(a[0]), r1 = row_bcast<K>(a[1]), r2 = row_bcast<K>(a[2]), r3 = row_bcast<K>(a[3]); \
        const float piv = bcast(a[kk], 16 * pk + (K));                                                        \
        const float pki = pivot_rcp(piv);                                                                     \
        const float f = spread_row<pk>(a[kk]);                                                                \
        int lk = r;                                                                                           \
        asm volatile("" : "+v"(lk));                                                                          \
        const bool me = lk == (K);                                                                            \
        const float g = me ? 0.f : f * pki;                                                                   \
        if (me) mine = pki;                                                                                   \
        const float own = me ? 1.f : -g;                                                                      \
        const float u0 = a[0] - g * r0, u1 = a[1] - g * r1, u2 = a[2] - g * r2, u3 = a[3] - g * r3;           \
        a[0] = (kk == 0 && p == pk) ? own : u0; a[1] = (kk == 1 && p == pk) ? own : u1;                       \
        a[2] = (kk == 2 && p == pk) ? own : u2; a[3] = (kk == 3 && p == pk) ? own : u3;                       \
    }
    WAVELA_STEP(0) WAVELA_STEP(1) WAVELA_STEP(2) WAVELA_STEP(3) WAVELA_STEP(4) WAVELA_STEP(5) WAVELA_STEP(6) WAVELA_STEP(7)
    WAVELA_STEP(8) WAVELA_STEP(9) WAVELA_STEP(10) WAVELA_STEP(11) WAVELA_STEP(12) WAVELA_STEP(13) WAVELA_STEP(14) WAVELA_STEP(15)
#undef WAVELA_STEP
    a[0] *= mine; a[1] *= mine; a[2] *= mine; a[3] *= mine;
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
