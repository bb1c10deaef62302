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
    T mine = T(1); // 1 / pivot of my row, applied once at the end (see spd_inverse_rows)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int hk = k / H, kk = k % H;                 // compile-time after unrolling
        T rowk[H];
#pragma unroll
        for (int j = 0; j < H; ++j) rowk[j] = fetch(a[j], k + 32 * half);   // my half of pivot row k
        const T piv = bcast(a[kk], k + 32 * hk);          // A[k][k], wave-uniform
        const T pk = pivot_rcp(piv);
        const T f = fetch(a[kk], r + 32 * hk);            // A[r][k] of my row (held by the half that owns column k)
        const bool me = r == k;
        const T g = me ? T(0) : f * pk;
        if (me) mine = pk;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const T upd = a[j] - g * rowk[j];
            a[j] = (j == kk && half == hk) ? (me ? T(1) : -g) : upd;
        }
    }
#pragma unroll
    for (int j = 0; j < H; ++j) a[j] *= mine;
}

// The two-lanes-per-row elimination with 2 x 2 PIVOT BLOCKS: N / 2 steps instead of N.  A step of the scalar form is one
// dependent chain (cross-lane permute of the pivot row -> reciprocal -> update -> next permute, ~600 cycles next to the
// LDS traffic of the other wavefronts) whatever the amount of arithmetic in it; two pivots per step halve the number of
// chains.  The 2 x 2 diagonal block of a Schur complement of an SPD matrix is SPD (determinant > 0, no pivoting).  Pivot
// rows take the same update with their accumulator cleared: row <- D^-1 [row_a; row_b].
template <class T, int N>
__device__ __forceinline__ void spd_inverse_rows_split2(T (&a)[N / 2], int lane)
{
    static_assert(N % 2 == 0 && N <= 32, "row halves on lanes r and r + 32, pivots in pairs");
    constexpr int H = N / 2;
    const int r = lane & 31, half = lane >> 5;
#pragma unroll
    for (int m = 0; m < N / 2; ++m) {
        const int ca = 2 * m, cb = 2 * m + 1;             // compile-time after unrolling
        const int ha = ca / H, ka = ca % H, hb = cb / H, kb = cb % H;
        T rowa[H], rowb[H];
#pragma unroll
        for (int j = 0; j < H; ++j) { rowa[j] = fetch(a[j], ca + 32 * half); rowb[j] = fetch(a[j], cb + 32 * half); }
        const T daa = bcast(a[ka], ca + 32 * ha), dab = bcast(a[kb], ca + 32 * hb), dbb = bcast(a[kb], cb + 32 * hb);
        const T idet = pivot_rcp(daa * dbb - dab * dab);
        const T i00 = dbb * idet, i01 = -dab * idet, i11 = daa * idet;
        const T fa = fetch(a[ka], r + 32 * ha), fb = fetch(a[kb], r + 32 * hb); // my row's entries in the two pivot columns
        const bool mea = r == ca, meb = r == cb;
        const T ga = mea ? -i00 : (meb ? -i01 : fa * i00 + fb * i01);
        const T gb = mea ? -i01 : (meb ? -i11 : fa * i01 + fb * i11);
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const T base = (mea || meb) ? T(0) : a[j];
            const T upd = base - ga * rowa[j] - gb * rowb[j];
            a[j] = (j == ka && half == ha) ? -ga : ((j == kb && half == hb) ? -gb : upd);
        }
    }
}

} // namespace wavela
