// nmpc_kernels.hip -- gfx950 kernels of the batched NMPC real-time iteration.
//
// Mapping.  One wavefront per workgroup; a wavefront holds G = 64 / L problems, each worked on by a
// group of L consecutive lanes (L = 4..64, a power of two).  All per-stage data of a problem live in
// LDS as N+1 "stage records" of 21 sixteen-byte slots (the odd slot count spreads consecutive
// records over all banks), followed by a staging copy of the read-only inputs.
//   * load: the reference layout is contiguous per problem, so a group streams its problem in with
//     16-byte-per-lane coalesced loads, all issued before the first LDS store (one memory round trip);
//   * stage-parallel phases (linearise + Gauss-Newton cost, KKT/expand, objective): lane j of the
//     group takes stages j, j+L, ...;
//   * backward Riccati sweep: sequential in the stages; the 3x3 algebra of a stage is split by rows
//     over the lanes of each quad (lane r holds row r of the cost-to-go) with DPP quad exchanges,
//     stage data are LDS broadcast reads, and the sweep restarts below the highest stage whose
//     working set changed (the cost-to-go of every node is kept in its record);
//   * forward sweep: the closed-loop stage maps are affine, so the rollout is a prefix scan over
//     lanes (lane j <-> stage j): log2(L) steps instead of N.
// Global traffic per tick is the algorithmic minimum: every input member is read once, x/u/dual
// and four scalars are written once.
//
// Reference for what is computed: see nmpc_core.h.  One launch runs n_sqp real-time iterations
// (acado_preparationStep + acado_feedbackStep, CG/acado_solver.c:1057-1077) for the whole batch.
#include "nmpc_kernels.h"

#include <cstdlib>

#include <type_traits>

#include "nmpc_core.h"

namespace nmpc {

// Stage record (floats; slot = 4 floats).  Record N is the terminal node.
//   slot 0  B00 B01 B10 B11      slot 7  lb0 ub0 lb1 ub1       slot 12 P00 P01 P02 p0   (cost-to-go
//   slot 1  B20 B21 a   b        slot 8  st0 st1 du0 du1       slot 13 P10 P11 P12 p1    at this node,
//   slot 2  d0  d1  d2  r0       slot 9  c00 c01 c02 f0        slot 14 P20 P21 P22 p2    row-wise)
//   slot 3  R00 R01 R11 r1       slot 10 c10 c11 c12 e1        slot 15 dx0 dx1 dx2 -
//   slot 4  Q00 Q01 Q02 q0       slot 11 mu0 mu1 f1  -         slot 16 x0  x1  x2  -
//   slot 5  Q10 Q11 Q12 q1                                     slot 17 u0  u1  y0  y1
//   slot 6  Q20 Q21 Q22 q2                                     slot 18 sb0 sb1 sb2 -
//   slot 19 free0 free1 val0 val1 (working set as 1/0 flags + fixed values, write_fix)   slot 20 spare (odd count)
constexpr int SLOTS = 21; // odd: consecutive records spread over all banks for 16-byte accesses
constexpr int SR = SLOTS * 4;
enum Slot : int {
    S_B0 = 0, S_B1 = 1, S_D = 2, S_R = 3, S_Q = 4, S_BND = 7, S_STDU = 8, S_POL0 = 9, S_POL1 = 10, S_MU = 11,
    S_V = 12, S_DX = 15, S_X = 16, S_UY = 17, S_SB = 18, S_FIX = 19 /* slot 20 spare */
};

// After the records: a staging copy of the read-only inputs of the problem, filled by coalesced
// loads, each sub-array padded to 16 bytes.
__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }
struct Staging {
    int W, y, od, lb, ub, term, end; // float offsets from the start of the staging area (term: WN[9], yN[3])
};
__host__ __device__ inline Staging staging_layout(int N, bool wreg = false)
{
    Staging s;
    s.W = 0;
    s.y = s.W + (wreg ? 0 : pad4(25 * N)); // wreg: W only passes through (folded into the records), then lives in registers
    s.od = s.y + pad4(5 * N);
    s.lb = s.od + pad4(3 * (N + 1));
    s.ub = s.lb + pad4(2 * N);
    s.term = s.ub + pad4(2 * N);
    s.end = s.term + 12;
    return s;
}

int rti_row_floats(int N, bool wreg) { return SR * (N + 1) + staging_layout(N, wreg).end; }

bool rti_geometry(int B, int N, int forced_L, int lds_limit_bytes, int n_cu, LaunchGeom* g, int forced_wpb)
{
    if (B <= 0 || N <= 0) return false;
    int L = forced_L;
    if (L == 0) {
        // One chunk of the lane scans should cover the horizon (L >= N): the forward sweep is then a
        // single scan and the working-set prediction is available.  Measured on MI355X at N = 20:
        // L = 32 is the fastest choice from B = 1 to B = 32768 (the sequential sweeps cost the same
        // for any L, a wider group shortens every stage-parallel phase and needs less LDS per
        // wavefront, which is what bounds residency).
        L = 16;
        while (L < 64 && L < N) L *= 2;
    }
    if (L != 4 && L != 8 && L != 16 && L != 32 && L != 64) return false;
    for (; L <= 64; L *= 2) {
        // rows of one wavefront must start on different 16-byte bank slots: RS/4 not a multiple of 16
        // Long horizons: the staged copy of W (100 B per stage) is what keeps residency below two wavefronts per
        // SIMD; with one lane per node (N + 1 <= L) the lane keeps its W_k in registers instead.
        int RS = rti_row_floats(N, false);
        bool wreg = false;
        static const int force_wreg = getenv("ALORE_NMPC_WREG") ? atoi(getenv("ALORE_NMPC_WREG")) : -1; // diagnostic
        if (L >= 32 && N + 1 <= L && (force_wreg == 1 || (force_wreg != 0 && 8L * (64 / L) * 4 * RS > lds_limit_bytes))) {
            wreg = true;
            RS = rti_row_floats(N, true);
        }
        if ((RS & 63) == 0) RS += 4;
        const long row_bytes = 4L * RS;
        const int G = 64 / L; // one wavefront per workgroup
        if ((long)G * row_bytes > lds_limit_bytes) {
            if (forced_L) return false;
            continue; // more lanes per problem = fewer problems per wavefront
        }
        // Four wavefronts per workgroup (one per SIMD: the hardware then spreads the wavefronts of a CU evenly
        // over its SIMDs) when two such workgroups fit a CU and the batch gives every CU at least one.
        const int cus = n_cu > 0 ? n_cu : 256;
        const int wpb = (forced_wpb == 1 || forced_wpb == 4) ? forced_wpb
                        : ((2L * 4 * G * row_bytes <= lds_limit_bytes && (long)B >= 4L * G * cus) ? 4 : 1);
        g->L = L;
        g->G = G;
        g->wpb = wpb;
        g->wreg = wreg ? 1 : 0;
        g->threads = 64 * wpb;
        g->grid = (B + G * wpb - 1) / (G * wpb);
        g->RS = RS;
        g->lds_bytes = (size_t)row_bytes * G * wpb;
        g->block = 0;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------

__device__ __forceinline__ float4 lds4(const float* rec, int slot)
{
    return *reinterpret_cast<const float4*>(rec + slot * 4);
}
__device__ __forceinline__ void st4(float* rec, int slot, float a, float b, float c, float d)
{
    *reinterpret_cast<float4*>(rec + slot * 4) = make_float4(a, b, c, d);
}

// Working set of a stage as the backward sweep wants it: (free0, free1, fixed value 0, fixed value 1), flags as
// 1.0 / 0.0 and the value 0 for a free control, so that the elimination needs no compares or selects.
__device__ __forceinline__ void write_fix(float* rec, int st0, int st1, float lb0, float ub0, float lb1, float ub1)
{
    st4(rec, S_FIX, st0 == ST_FREE ? 1.0f : 0.0f, st1 == ST_FREE ? 1.0f : 0.0f,
        st0 == ST_FREE ? 0.0f : (st0 == ST_UPPER ? ub0 : lb0), st1 == ST_FREE ? 0.0f : (st1 == ST_UPPER ? ub1 : lb1));
}

// Wavefronts never share data: every problem lives in the LDS rows of its own wavefront.  With one
// wavefront per workgroup a workgroup barrier is just a scheduling fence; with several (WPB = 4: one per
// SIMD, which is what makes the hardware spread them evenly over the SIMDs of a CU) the barrier must NOT
// couple them, so the kernel orders its LDS traffic per wavefront only (LDS operations of one wavefront
// execute in program order).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int wave_or(int pred) { return __any(pred) ? 1 : 0; }

// DPP quad permutes: data of another lane of the same quad, no LDS traffic
template <int CTRL>
__device__ __forceinline__ float dpp(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
constexpr int QP_BC0 = 0x00, QP_BC1 = 0x55, QP_BC2 = 0xAA, QP_SWAP1 = 0xB1, QP_SWAP2 = 0x4E;
__device__ __forceinline__ float quad_sum(float x)
{
    x += dpp<QP_SWAP1>(x);
    x += dpp<QP_SWAP2>(x);
    return x;
}
// base + the values of lanes 0, 1, 2 of the quad (lane 3 is not read): three DPP-operand adds, the same
// summation order in every lane
__device__ __forceinline__ float rows_sum(float base, float x)
{
    float h = base + dpp<QP_BC0>(x);
    h += dpp<QP_BC1>(x);
    h += dpp<QP_BC2>(x);
    return h;
}

template <int L>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, L);
    return v;
}
template <int L>
__device__ __forceinline__ int group_or(int v)
{
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v |= __shfl_xor(v, off, L);
    return v;
}
// "any lane of my group": one ballot instead of a butterfly of LDS permutes (base = first lane of the group)
template <int L>
__device__ __forceinline__ bool group_any(bool pred, int base)
{
    const unsigned long long m = __ballot(pred);
    if constexpr (L == 64) return m != 0ull;
    else return ((m >> base) & ((1ull << L) - 1ull)) != 0ull;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

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
// value held by the last lane of the group.  v_readlane through inline asm with explicit wait
// states: the builtin form returned stale data right behind the DPP / VALU producers here.
__device__ __forceinline__ float readlane_f(float v, int lane31_or_63)
{
    float r;
    if (lane31_or_63 == 31)
        asm volatile("s_nop 4\n\tv_readlane_b32 %0, %1, 31\n\ts_nop 1" : "=s"(r) : "v"(v));
    else
        asm volatile("s_nop 4\n\tv_readlane_b32 %0, %1, 63\n\ts_nop 1" : "=s"(r) : "v"(v));
    return r;
}
template <int L>
__device__ __forceinline__ float group_last(float v)
{
    if constexpr (L == 64) {
        return readlane_f(v, 63);
    } else if constexpr (L == 32) {
        // ds_swizzle, bit-mask mode (and = 0, or = 31, xor = 0): every lane reads lane 31 of its half
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x03E0));
    } else {
        return __shfl(v, L - 1, L);
    }
}
// group-wide sum / max through the DPP scans
template <int L>
__device__ __forceinline__ float group_total(float v, int j)
{
    return group_last<L>(prefix_sum<L>(v, j));
}
template <int L, int STEP = 0>
__device__ __forceinline__ int prefix_max_i(int x, int j)
{
    if constexpr (STEP < scan_steps<L>()) {
        const int o = __float_as_int(scan_prev<L, STEP>(__int_as_float(x), __int_as_float(-2147483647 - 1), j));
        return prefix_max_i<L, STEP + 1>(max(x, o), j);
    } else {
        return x;
    }
}
template <int L>
__device__ __forceinline__ int group_max_i(int v, int j)
{
    return __float_as_int(group_last<L>(__int_as_float(prefix_max_i<L>(v, j))));
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

// Group-cooperative copies global -> LDS.  The loads of a batch are all issued before the first
// store, so a batch costs one memory round trip; 16 bytes per lane where the element count (hence
// every per-problem base address) allows it.
template <int L, int U>
__device__ __forceinline__ void ldg_vec(const float* g, int n4, int j, float4 (&q)[U])
{
    const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = u * L + j;
        if (i < n4) {
            // read once, never again: non-temporal (streaming) loads
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g4 + i));
            q[u] = make_float4(v.x, v.y, v.z, v.w);
        } else {
            q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
template <int L, int U>
__device__ __forceinline__ void sts_vec(float* l, int n4, int j, const float4 (&q)[U])
{
    float4* l4 = reinterpret_cast<float4*>(l);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = u * L + j;
        if (i < n4) l4[i] = q[u];
    }
}
template <int L, int U>
__device__ __forceinline__ void ldg_scl(const float* g, int n, int j, float (&f)[U])
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = u * L + j;
        f[u] = (i < n) ? g[i] : 0.0f;
    }
}
// plain loop for whatever the first batch did not cover (long horizons) or for unaligned arrays
template <int L>
__device__ __forceinline__ void copy_tail(const float* g, float* l, int from, int n, int j)
{
    for (int i = from + j; i < n; i += L) l[i] = g[i];
}
__device__ __forceinline__ bool vec_ok(const float* g, int n)
{
    return ((n & 3) == 0) && ((reinterpret_cast<size_t>(g) & 15) == 0);
}

// ---- backward Riccati step, rows of the 3x3 algebra spread over the lanes of a quad ------------
// Lane r (0..2) of every quad holds row r of the cost-to-go: P[r][0..2] and p[r]; lane 3 shadows
// lane 2 and contributes zero to the quad sums.  Same algebra as nmpc_core.h::riccati_step.
struct RowValue {
    float P0, P1, P2, p;
};
struct StageBcast { // wavefront-uniform-per-group stage data (LDS broadcast reads)
    float4 b0, b1, d, R, fix; // fix: see write_fix
};
struct RowPolicy {
    float c0c, c1c;     // component r of the two gain / multiplier rows
    float f0, e1, f1;   // group-uniform
};

__device__ __forceinline__ void load_bcast(const float* rec, StageBcast& s)
{
    s.b0 = lds4(rec, S_B0); s.b1 = lds4(rec, S_B1); s.d = lds4(rec, S_D); s.R = lds4(rec, S_R);
    s.fix = lds4(rec, S_FIX);
}

__device__ __forceinline__ int riccati_rows(const StageBcast& s, float4 qrow, float Br0, float Br1, bool is2,
                                            RowValue& V, RowPolicy& pol, bool need_value)
{
    const float B00 = s.b0.x, B01 = s.b0.y, B10 = s.b0.z, B11 = s.b0.w, B20 = s.b1.x, B21 = s.b1.y;
    const float a = s.b1.z, b = s.b1.w;
    const float al = is2 ? a : 0.0f, be = is2 ? b : 0.0f;
    const float f0m = s.fix.x, f1m = s.fix.y, v0 = s.fix.z, v1 = s.fix.w; // 1 / 0 flags, fixed value (0 if free)
    const float n0m = 1.0f - f0m, n1m = 1.0f - f1m;

    const float sr = V.P0 * s.d.x + V.P1 * s.d.y + V.P2 * s.d.z + V.p;     // (P d + p)[r]
    const float PB0 = V.P0 * B00 + V.P1 * B10 + V.P2 * B20;                 // (P B)[r][0]
    const float PB1 = V.P0 * B01 + V.P1 * B11 + V.P2 * B21;                 // (P B)[r][1]
    const float H00 = rows_sum(s.R.x, Br0 * PB0);                           // R + B' P B
    const float H01 = rows_sum(s.R.y, Br0 * PB1);
    const float H11 = rows_sum(s.R.z, Br1 * PB1);
    float hu0 = rows_sum(s.d.w, Br0 * sr);                                  // r + B' s
    const float hu1 = rows_sum(s.R.w, Br1 * sr);
    // component r of the rows of Hux = B' P A
    float G0c = PB0 + al * dpp<QP_BC0>(PB0) + be * dpp<QP_BC1>(PB0);
    const float G1c = PB1 + al * dpp<QP_BC0>(PB1) + be * dpp<QP_BC1>(PB1);

    // eliminate control 1, then control 0 (group-uniform scalars).  Flags as arithmetic masks: a free control
    // has w = 1/H, z = -hu/H; a fixed one w = 0, z = its value; exact either way (the masks are 0 and 1).
    const bool bad1 = (f1m > 0.0f) && !(H11 > 0.0f);
    const float w1 = __builtin_amdgcn_rcpf(H11) * f1m; // reciprocal: 1 ulp, like the division it stands for
    const float z1 = fmaf(-hu1, w1, v1);
    const float t1 = w1 * H01;
    const float g1s = n1m - w1;                  // -1/H if free, 1 if fixed
    pol.c1c = g1s * G1c;
    pol.e1 = g1s * H01;
    pol.f1 = fmaf(f1m, z1, n1m * fmaf(H11, v1, hu1)); // z1 if free, the multiplier's constant term if fixed
    const float H00r = H00 - t1 * H01;
    G0c -= t1 * G1c;
    hu0 += H01 * z1;
    const bool bad0 = (f0m > 0.0f) && !(H00r > 0.0f);
    const float w0 = __builtin_amdgcn_rcpf(H00r) * f0m;
    const float z0 = fmaf(-hu0, w0, v0);
    const float g0s = n0m - w0;
    pol.c0c = g0s * G0c;
    pol.f0 = fmaf(f0m, z0, n0m * fmaf(H00r, v0, hu0));
    if (need_value) {
        // row r of Q + A' P A and of q + A' s
        const float m = a * V.P0 + b * V.P1 + V.P2; // (P A)[r][2]
        const float m0 = dpp<QP_BC0>(m), m1 = dpp<QP_BC1>(m);
        float X0 = qrow.x + (is2 ? m0 : V.P0);
        float X1 = qrow.y + (is2 ? m1 : V.P1);
        float X2 = qrow.z + m + al * m0 + be * m1;
        float hx = qrow.w + sr + al * dpp<QP_BC0>(sr) + be * dpp<QP_BC1>(sr);
        // minus the rank-one terms of the two eliminated controls
        const float wg1 = w1 * G1c, wg0 = w0 * G0c;
        X0 -= wg1 * dpp<QP_BC0>(G1c) + wg0 * dpp<QP_BC0>(G0c);
        X1 -= wg1 * dpp<QP_BC1>(G1c) + wg0 * dpp<QP_BC1>(G0c);
        X2 -= wg1 * dpp<QP_BC2>(G1c) + wg0 * dpp<QP_BC2>(G0c);
        hx += G1c * z1 + G0c * z0;
        V.P0 = X0; V.P1 = X1; V.P2 = X2; V.p = hx;
    }
    return (bad0 || bad1) ? 0 : 1;
}

// ---- primal active-set safeguard (nmpc_core.h: AS_SWITCH) ----------------------------------------------
// Runs after the main loop for the rare problems whose primal-dual iteration has not settled, out of
// line and deliberately plain (every lane of the group walks the stages itself: no scans, no prefetch)
// so that it costs the main loop nothing.  On entry the records hold the last working-set solution
// (du in S_STDU.zw); on exit they hold the solution (du, multipliers, dx, statuses) like after a
// settled main loop.  The feasible point of the method lives in the spare floats S_SB.w / S_DX.w.
// Returns (sweep count) | bit 29: a stage Hessian pivot was not positive | bit 30: iteration cap reached.
// (Everything by value: a reference argument would give the kernel a scratch frame.)
template <int L>
__device__ __forceinline__ int active_set_rescue(float* row, int N, int j, bool mine, float Dx0, float Dx1, float Dx2,
                                              int max_it, int it)
{
    int pd_fail = 0;
    const int rq = j & 3, rr = (rq < 3) ? rq : 2;
    const bool is2 = (rr == 2);
    wave_sync();
    if (mine && j == 0) { // start: clip the last solution into the box, fix what sits on a bound
        for (int k = 0; k < N; ++k) {
            float* rec = row + k * SR;
            const float4 sd = lds4(rec, S_STDU), bnd = lds4(rec, S_BND);
            const float c0 = (bnd.x <= bnd.y) ? clampf(sd.z, bnd.x, bnd.y) : sd.z;
            const float c1 = (bnd.z <= bnd.w) ? clampf(sd.w, bnd.z, bnd.w) : sd.w;
            const int s0 = (bnd.y - bnd.x > BOUNDTOL) ? asm_status_of(c0, bnd.x, bnd.y) : ST_LOWER;
            const int s1 = (bnd.w - bnd.z > BOUNDTOL) ? asm_status_of(c1, bnd.z, bnd.w) : ST_LOWER;
            rec[S_SB * 4 + 3] = c0; rec[S_DX * 4 + 3] = c1;
            *reinterpret_cast<float2*>(rec + S_STDU * 4) = make_float2(__int_as_float(s0), __int_as_float(s1));
            write_fix(rec, s0, s1, bnd.x, bnd.y, bnd.z, bnd.w);
        }
    }
    wave_sync();
    int todo = mine ? 1 : 0;
    for (;;) {
        // backward sweep over the whole horizon (rows over the quad lanes, as in the main loop)
        {
            const float4 vN = lds4(row + N * SR, S_V + rr);
            RowValue V;
            V.P0 = vN.x; V.P1 = vN.y; V.P2 = vN.z; V.p = vN.w;
            int ok = 1;
            for (int k = N - 1; k >= 0; --k) {
                float* rec = row + k * SR;
                StageBcast sb;
                load_bcast(rec, sb);
                const float4 qrow = lds4(rec, S_Q + rr);
                const float2 brow = *reinterpret_cast<const float2*>(rec + 2 * rr);
                RowPolicy pol;
                ok &= riccati_rows(sb, qrow, brow.x, brow.y, is2, V, pol, k > 0);
                if (j < 4 && todo) {
                    rec[S_POL0 * 4 + j] = (j < 3) ? pol.c0c : pol.f0;
                    rec[S_POL1 * 4 + j] = (j < 3) ? pol.c1c : pol.e1;
                    const int slot = (j < 3) ? S_V + j : S_MU;
                    st4(rec, slot, (j < 3) ? V.P0 : 0.0f, (j < 3) ? V.P1 : 0.0f, (j < 3) ? V.P2 : pol.f1, (j < 3) ? V.p : 0.0f);
                }
            }
            if (todo) pd_fail |= (ok == 0);
        }
        wave_sync();
        // forward sweep, sequential, with the ratio test (free controls) and the multiplier test (fixed ones)
        float alpha = AS_NONE, viol = 0.0f;
        int akey = -1, vkey = -1; // (2 * stage + control) * 4 + bound hit
        if (todo) {
            float dx0 = Dx0, dx1 = Dx1, dx2 = Dx2;
            for (int k = 0; k < N; ++k) {
                float* rec = row + k * SR;
                const float4 l0 = lds4(rec, S_B0), l1 = lds4(rec, S_B1), dd = lds4(rec, S_D), bnd = lds4(rec, S_BND),
                             sd = lds4(rec, S_STDU), p0 = lds4(rec, S_POL0), p1 = lds4(rec, S_POL1), mu = lds4(rec, S_MU);
                const int st0 = __float_as_int(sd.x), st1 = __float_as_int(sd.y);
                Policy pol;
                pol.c00 = p0.x; pol.c01 = p0.y; pol.c02 = p0.z; pol.f0 = p0.w;
                pol.c10 = p1.x; pol.c11 = p1.y; pol.c12 = p1.z; pol.e1 = p1.w; pol.f1 = mu.z;
                StageStep o;
                forward_step(pol, st0, st1, dx0, dx1, dx2, bnd.x, bnd.y, bnd.z, bnd.w, o);
                const float c0 = rec[S_SB * 4 + 3], c1 = rec[S_DX * 4 + 3];
                int h;
                const float r0 = asm_ratio(st0, c0, o.du0, bnd.x, bnd.y, h);
                if (r0 < alpha) { alpha = r0; akey = (2 * k) * 4 + h; }
                const float r1 = asm_ratio(st1, c1, o.du1, bnd.z, bnd.w, h);
                if (r1 < alpha) { alpha = r1; akey = (2 * k + 1) * 4 + h; }
                const float v0 = asm_violation(st0, o.mu0, bnd.x, bnd.y), v1 = asm_violation(st1, o.mu1, bnd.z, bnd.w);
                if (v0 > viol) { viol = v0; vkey = (2 * k) * 4; }
                if (v1 > viol) { viol = v1; vkey = (2 * k + 1) * 4; }
                if (j == 0) {
                    *reinterpret_cast<float2*>(rec + S_MU * 4) = make_float2(o.mu0, o.mu1);
                    rec[S_DX * 4] = dx0; rec[S_DX * 4 + 1] = dx1; rec[S_DX * 4 + 2] = dx2;
                    *reinterpret_cast<float2*>(rec + S_STDU * 4 + 2) = make_float2(o.du0, o.du1);
                }
                const float n0 = dx0 + l1.z * dx2 + l0.x * o.du0 + l0.y * o.du1 + dd.x;
                const float n1 = dx1 + l1.w * dx2 + l0.z * o.du0 + l0.w * o.du1 + dd.y;
                const float n2 = dx2 + l1.x * o.du0 + l1.y * o.du1 + dd.z;
                dx0 = n0; dx1 = n1; dx2 = n2;
            }
            if (j == 0) {
                float* rec = row + N * SR;
                rec[S_DX * 4] = dx0; rec[S_DX * 4 + 1] = dx1; rec[S_DX * 4 + 2] = dx2;
            }
            ++it;
        }
        wave_sync();
        // one change of the working set (every lane of the group holds the same alpha / viol / keys)
        if (todo) {
            const bool blocked = akey >= 0;
            const bool release = !blocked && vkey >= 0;
            const float al = blocked ? fmaxf(alpha, 0.0f) : 1.0f;
            if (j == 0) {
                for (int k = 0; k < N; ++k) {
                    float* rec = row + k * SR;
                    const float4 sd = lds4(rec, S_STDU), bnd = lds4(rec, S_BND);
                    int st0 = __float_as_int(sd.x), st1 = __float_as_int(sd.y);
                    float c0 = rec[S_SB * 4 + 3], c1 = rec[S_DX * 4 + 3];
                    if (st0 == ST_FREE) c0 += al * (sd.z - c0);
                    if (st1 == ST_FREE) c1 += al * (sd.w - c1);
                    if (blocked && (akey >> 3) == k) {
                        const int hit = akey & 3;
                        if ((akey >> 2) & 1) { c1 = (hit == ST_UPPER) ? bnd.w : bnd.z; st1 = hit; }
                        else { c0 = (hit == ST_UPPER) ? bnd.y : bnd.x; st0 = hit; }
                    }
                    if (release && (vkey >> 3) == k) {
                        if ((vkey >> 2) & 1) st1 = ST_FREE; else st0 = ST_FREE;
                    }
                    rec[S_SB * 4 + 3] = c0; rec[S_DX * 4 + 3] = c1;
                    *reinterpret_cast<float2*>(rec + S_STDU * 4) = make_float2(__int_as_float(st0), __int_as_float(st1));
                    write_fix(rec, st0, st1, bnd.x, bnd.y, bnd.z, bnd.w);
                }
            }
            if (!blocked && !release) todo = 0;     // optimal
            else if (it >= max_it) todo = 2;        // cap reached
        }
        const int more = wave_or(todo == 1 ? 1 : 0);
        if (!more) break;
    }
    return it | (pd_fail ? (1 << 29) : 0) | (todo == 2 ? (1 << 30) : 0);
}

// one wavefront per workgroup, G = 64 / L problems per wavefront
template <int L, bool STAMP, int WPB, bool WREG, bool DIAG>
__global__ __launch_bounds__(64 * WPB) void rti_kernel(const RtiParams p)
{
    extern __shared__ float4 lds_raw[];
    float* lds = reinterpret_cast<float*>(lds_raw);
    const int N = p.N;
    constexpr int G = 64 / L;
    const int wave = (WPB == 1) ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int grp = wave * G + lane / L; // group index inside the workgroup
    const int j = lane % L;
    const bool writer = (j == 0);
    int prob = blockIdx.x * (G * WPB) + grp;
    const bool valid = prob < p.B;
    if (!valid) prob = p.B - 1; // padding groups shadow the last problem, never store
    float* row = lds + (size_t)grp * p.RS;

    long long t_stamp[6];
    if (STAMP) t_stamp[0] = __builtin_amdgcn_s_memtime();

    IrkConst K;
    K.h = p.h; K.hh = p.hh; K.c1h = p.c1h; K.c2h = p.c2h;

    const float* gx = p.b.x + (size_t)prob * (N + 1) * 3;
    const float* gu = p.b.u + (size_t)prob * N * 2;
    const float* gdual = p.b.dual + (size_t)prob * N * 2;
    const float* god = p.b.od + ((p.shared & ALORE_NMPC_SHARED_OD) ? 0 : (size_t)prob * (N + 1) * 3);
    const float* gy = p.b.y + (size_t)prob * N * 5;
    const float* gyN = p.b.yN + (size_t)prob * 3;
    const float* gW = p.b.W + ((p.shared & ALORE_NMPC_SHARED_W) ? 0 : (size_t)prob * N * 25);
    const float* gWN = p.b.WN + ((p.shared & ALORE_NMPC_SHARED_W) ? 0 : (size_t)prob * 9);
    const float* glb = p.b.lbValues + ((p.shared & ALORE_NMPC_SHARED_BOUNDS) ? 0 : (size_t)prob * N * 2);
    const float* gub = p.b.ubValues + ((p.shared & ALORE_NMPC_SHARED_BOUNDS) ? 0 : (size_t)prob * N * 2);

    // ---- phase 0: coalesced loads.  Read-only inputs -> staging, iterate (x, u, dual) -> records.
    const Staging SG = staging_layout(N, WREG);
    float* stg = row + (N + 1) * SR;
    {
        constexpr int UW = (128 + L - 1) / L; // first batch: up to 128 float4 of W (a 20-node horizon)
        constexpr int UV = (32 + L - 1) / L;  // ... 32 float4 of y / lb / ub
        constexpr int UX = (64 + L - 1) / L;  // ... 64 floats of od / x / u / dual
        const bool vW = vec_ok(gW, 25 * N), vy = vec_ok(gy, 5 * N), vb = vec_ok(glb, 2 * N) && vec_ok(gub, 2 * N);
        float4 qW[UW], qy[UV], qlb[UV], qub[UV];
        float fod[UX], fx[UX], fu[UX], fd[UX];
        if (vW) ldg_vec<L, UW>(gW, (25 * N) >> 2, j, qW);
        if (vy) ldg_vec<L, UV>(gy, (5 * N) >> 2, j, qy);
        if (vb) { ldg_vec<L, UV>(glb, (2 * N) >> 2, j, qlb); ldg_vec<L, UV>(gub, (2 * N) >> 2, j, qub); }
        ldg_scl<L, UX>(god, 3 * (N + 1), j, fod);
        ldg_scl<L, UX>(gx, 3 * (N + 1), j, fx);
        ldg_scl<L, UX>(gu, 2 * N, j, fu);
        ldg_scl<L, UX>(gdual, 2 * N, j, fd);
        const float fterm = (j < 9) ? gWN[j] : ((j < 12) ? gyN[j - 9] : 0.0f); // terminal weights and reference (L >= 12)
        auto wfold = [&](int i) { return row + (i >> 6) * SR + (i & 63); }; // WREG: float i of W in slots 0..15 of record i / 64
        if (vW) {
            if constexpr (WREG) {
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int i = u * L + j;
                    if (i < ((25 * N) >> 2)) *reinterpret_cast<float4*>(wfold(4 * i)) = qW[u];
                }
            } else {
                sts_vec<L, UW>(stg + SG.W, (25 * N) >> 2, j, qW);
            }
        }
        if (vy) sts_vec<L, UV>(stg + SG.y, (5 * N) >> 2, j, qy);
        if (vb) { sts_vec<L, UV>(stg + SG.lb, (2 * N) >> 2, j, qlb); sts_vec<L, UV>(stg + SG.ub, (2 * N) >> 2, j, qub); }
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int i = u * L + j;
            if (i < 3 * (N + 1)) {
                stg[SG.od + i] = fod[u];
                row[(i / 3) * SR + S_X * 4 + (i % 3)] = fx[u];
            }
            if (i < 2 * N) {
                float* rec = row + (i >> 1) * SR + S_UY * 4 + (i & 1);
                rec[0] = fu[u];
                rec[2] = fd[u];
            }
        }
        if (L >= 12) { if (j < 12) stg[SG.term + j] = fterm; }
        else { for (int i = j; i < 12; i += L) stg[SG.term + i] = (i < 9) ? gWN[i] : gyN[i - 9]; }
        // whatever the first batch did not cover
        if constexpr (WREG) { for (int i = (vW ? min(25 * N, 4 * UW * L) : 0) + j; i < 25 * N; i += L) *wfold(i) = gW[i]; }
        else copy_tail<L>(gW, stg + SG.W, vW ? min(25 * N, 4 * UW * L) : 0, 25 * N, j);
        copy_tail<L>(gy, stg + SG.y, vy ? min(5 * N, 4 * UV * L) : 0, 5 * N, j);
        copy_tail<L>(glb, stg + SG.lb, vb ? min(2 * N, 4 * UV * L) : 0, 2 * N, j);
        copy_tail<L>(gub, stg + SG.ub, vb ? min(2 * N, 4 * UV * L) : 0, 2 * N, j);
        copy_tail<L>(god, stg + SG.od, UX * L, 3 * (N + 1), j);
        for (int i = UX * L + j; i < 3 * (N + 1); i += L) row[(i / 3) * SR + S_X * 4 + (i % 3)] = gx[i];
        for (int i = UX * L + j; i < 2 * N; i += L) {
            float* rec = row + (i >> 1) * SR + S_UY * 4 + (i & 1);
            rec[0] = gu[i];
            rec[2] = gdual[i];
        }
    }
    float x00 = p.b.x0[(size_t)prob * 3], x01 = p.b.x0[(size_t)prob * 3 + 1], x02 = p.b.x0[(size_t)prob * 3 + 2];
    wave_sync();
    // WREG: W_j of this lane's stage from the folded pass-through area into registers, for the whole launch
    float wreg[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) wreg[i] = 0.0f;
    if constexpr (WREG) {
        if (j < N) {
#pragma unroll
            for (int i = 0; i < 25; ++i) wreg[i] = row[((25 * j + i) >> 6) * SR + ((25 * j + i) & 63)];
        }
        wave_sync(); // the records are about to overwrite the pass-through area
    }

    // problems the caller masked out (alore_nmpc_set_problem_mask: idle robots of a fleet) run along and store nothing.  Their
    // references may be stale or non-finite: they are replaced by the iterate itself (zero tracking error), so that such a
    // problem cannot keep its wavefront in the working-set loop up to max_as_iter
    const bool sit_out = valid && p.mask != nullptr && p.mask[prob] == 0;
    if (__any(sit_out)) {
        if (sit_out) {
            for (int k = j; k < N; k += L) {
                const float4 xk = lds4(row + k * SR, S_X), uy = lds4(row + k * SR, S_UY);
                float* yk = stg + SG.y + k * 5;
                yk[0] = xk.x; yk[1] = xk.y; yk[2] = xk.z; yk[3] = uy.x; yk[4] = uy.y;
            }
            if (j == 0) {
                const float4 xe = lds4(row + N * SR, S_X);
                stg[SG.term + 9] = xe.x; stg[SG.term + 10] = xe.y; stg[SG.term + 11] = xe.z;
            }
        }
        wave_sync();
        if (sit_out) { // ... and the state estimate by node 0 of the iterate
            const float4 xb = lds4(row, S_X);
            x00 = xb.x; x01 = xb.y; x02 = xb.z;
        }
    }
    const bool keep = valid && !sit_out;

    // row role of this lane inside its quad (backward sweep)
    const int rq = j & 3;
    const int rr = (rq < 3) ? rq : 2; // lane 3 shadows lane 2 (rows_sum never reads it)
    const bool is2 = (rr == 2);

    // acado_getKKT / acado_getObjective are separate calls in the reference, never made by its wrapper:
    // computed here only when the caller asks for them (non-null kkt / obj)
    // (DIAG = false: compiled out; chosen by the launcher when both pointers are null)
    constexpr bool want_kkt = DIAG, want_obj = DIAG;
    int status = RET_OK, n_iter = 0;
    float kkt = 0.0f;

    for (int sqp = 0; sqp < p.n_sqp; ++sqp) {
        // ---- phase A (stage-parallel): linearise, Gauss-Newton cost, bounds, working-set guess
        int infeasible = 0;
        for (int k = j; k <= N; k += L) {
            float* rec = row + k * SR;
            float4 xk = lds4(rec, S_X);
            // ACADO-compatibility mode, first iteration only: linearise (and evaluate the residual) at
            // the point the caller prepared on, which may differ from the iterate that is expanded
            const bool relin = (sqp == 0) && (p.lin_x != nullptr);
            if (relin) {
                const float* lx = p.lin_x + ((size_t)prob * (N + 1) + k) * 3;
                xk = make_float4(lx[0], lx[1], lx[2], 0.0f);
            }
            if (k < N) {
                const float4 uy = lds4(rec, S_UY);
                float4 xn = lds4(rec + SR, S_X);
                float vr = uy.x, vl = uy.y;
                if (relin) {
                    const float* lx = p.lin_x + ((size_t)prob * (N + 1) + k + 1) * 3;
                    const float* lu = p.lin_u + ((size_t)prob * N + k) * 2;
                    xn = make_float4(lx[0], lx[1], lx[2], 0.0f);
                    vr = lu[0]; vl = lu[1];
                }
                StageLin lin;
                const float* odk = stg + SG.od + k * 3;
                ddr_linearize(K, xk.x, xk.y, xk.z, vr, vl, odk[0], odk[1], odk[2], lin);
                const float d0 = lin.phi0 - xn.x, d1 = lin.phi1 - xn.y, d2 = lin.phi2 - xn.z;
                // Dy = h(x,u) - y ; gradient = W[rows] * Dy ; Hessian blocks of W
                const float* yk = stg + SG.y + k * 5;
                const float e0 = xk.x - yk[0], e1 = xk.y - yk[1], e2 = xk.z - yk[2], e3 = vr - yk[3], e4 = vl - yk[4];
                float w[25];
#pragma unroll
                for (int i = 0; i < 25; ++i) w[i] = WREG ? wreg[i] : stg[SG.W + k * 25 + i]; // WREG: k == j
                const float q0 = w[0] * e0 + w[1] * e1 + w[2] * e2 + w[3] * e3 + w[4] * e4;
                const float q1 = w[5] * e0 + w[6] * e1 + w[7] * e2 + w[8] * e3 + w[9] * e4;
                const float q2 = w[10] * e0 + w[11] * e1 + w[12] * e2 + w[13] * e3 + w[14] * e4;
                const float r0 = w[15] * e0 + w[16] * e1 + w[17] * e2 + w[18] * e3 + w[19] * e4;
                const float r1 = w[20] * e0 + w[21] * e1 + w[22] * e2 + w[23] * e3 + w[24] * e4;
                const float* lbk = stg + SG.lb + k * 2;
                const float* ubk = stg + SG.ub + k * 2;
                const float lb0 = lbk[0] - uy.x, lb1 = lbk[1] - uy.y; // bounds on du from the current u
                const float ub0 = ubk[0] - uy.x, ub1 = ubk[1] - uy.y;
                infeasible |= (lb0 > ub0 + 1e-6f) || (lb1 > ub1 + 1e-6f);
                st4(rec, S_B0, lin.B00, lin.B01, lin.B10, lin.B11);
                st4(rec, S_B1, lin.B20, -lin.B20, lin.a, lin.b);
                st4(rec, S_D, d0, d1, d2, r0);
                st4(rec, S_R, w[18], w[19], w[24], r1);
                st4(rec, S_Q, w[0], w[1], w[2], q0);
                st4(rec, S_Q + 1, w[5], w[6], w[7], q1);
                st4(rec, S_Q + 2, w[10], w[11], w[12], q2);
                st4(rec, S_BND, lb0, ub0, lb1, ub1);
                const int is0_ = status_from_dual(uy.z, lb0, ub0), is1_ = status_from_dual(uy.w, lb1, ub1);
                st4(rec, S_STDU, __int_as_float(is0_), __int_as_float(is1_), 0.0f, 0.0f);
                write_fix(rec, is0_, is1_, lb0, ub0, lb1, ub1);
            } else {
                const float* tn = stg + SG.term; // WN (9), yN (3): staged with the other inputs
                const float e0 = xk.x - tn[9], e1 = xk.y - tn[10], e2 = xk.z - tn[11];
                float w[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) w[i] = tn[i];
                const float q0 = w[0] * e0 + w[1] * e1 + w[2] * e2;
                const float q1 = w[3] * e0 + w[4] * e1 + w[5] * e2;
                const float q2 = w[6] * e0 + w[7] * e1 + w[8] * e2;
                st4(rec, S_Q, w[0], w[1], w[2], q0);
                st4(rec, S_Q + 1, w[3], w[4], w[5], q1);
                st4(rec, S_Q + 2, w[6], w[7], w[8], q2);
                // cost-to-go at the terminal node
                st4(rec, S_V, w[0], w[1], w[2], q0);
                st4(rec, S_V + 1, w[3], w[4], w[5], q1);
                st4(rec, S_V + 2, w[6], w[7], w[8], q2);
            }
        }
        infeasible = group_any<L>(infeasible != 0, lane - j) ? 1 : 0;
        wave_sync();
        if (STAMP && sqp == 0) t_stamp[1] = __builtin_amdgcn_s_memtime();

        const float4 xfirst = lds4(row, S_X);
        const float Dx0 = x00 - xfirst.x, Dx1 = x01 - xfirst.y, Dx2 = x02 - xfirst.z;

        // ---- working-set prediction for cold starts (no dual information, every control free): if the
        //      clipped Jacobi step of the condensed QP hits a bound, a few projected-gradient steps
        //      (Hessian applied stage-wise by prefix / suffix sums over the lanes, Barzilai-Borwein step
        //      lengths) predict which bounds are active, and that set is what the first
        //      Riccati sweep starts from.  Only a guess: the sweeps below still iterate until the
        //      working set reproduces itself, so the solution is the same with or without it.
        long long t_pg = 0;
        if (p.pg_steps > 0 && N <= L) {
            long long tp0 = 0;
            if (STAMP) tp0 = __builtin_amdgcn_s_memtime();
            const bool in = j < N;
            float* rec = row + (in ? j : 0) * SR;
            const float* recn = rec + SR; // node j+1
            const float msk = in ? 1.0f : 0.0f;
            const float4 l0 = lds4(rec, S_B0), l1 = lds4(rec, S_B1), dd = lds4(rec, S_D), RR = lds4(rec, S_R),
                         bnd = lds4(rec, S_BND), sd = lds4(rec, S_STDU);
            const float4 q0 = lds4(recn, S_Q), q1 = lds4(recn, S_Q + 1), q2 = lds4(recn, S_Q + 2), ln = lds4(recn, S_B1);
            const int nonfree = in ? (__float_as_int(sd.x) | __float_as_int(sd.y)) : 0;
            const bool cold = !group_any<L>(nonfree != 0, lane - j);
            const float B00 = l0.x * msk, B01 = l0.y * msk, B10 = l0.z * msk, B11 = l0.w * msk, B20 = l1.x * msk;
            const float a = l1.z * msk, b = l1.w * msk, d0 = dd.x * msk, d1 = dd.y * msk, d2 = dd.z * msk;
            const float r0 = dd.w * msk, r1 = RR.w * msk, R00 = RR.x * msk, R01 = RR.y * msk, R11 = RR.z * msk;
            const float Q00 = q0.x * msk, Q01 = q0.y * msk, Q02 = q0.z * msk, qv0 = q0.w * msk;
            const float Q10 = q1.x * msk, Q11 = q1.y * msk, Q12 = q1.z * msk, qv1 = q1.w * msk;
            const float Q20 = q2.x * msk, Q21 = q2.y * msk, Q22 = q2.z * msk, qv2 = q2.w * msk;
            const bool has_next = (j + 1 < N);
            const float an = has_next ? ln.z : 0.0f, bn = has_next ? ln.w : 0.0f; // shear of stage j+1
            const float lb0 = bnd.x * msk, ub0 = bnd.y * msk, lb1 = bnd.z * msk, ub1 = bnd.w * msk;
            // H du (AFF = false) or the gradient H du + g (AFF = true) of the condensed QP
            auto apply = [&](float u0, float u1, auto aff_tag, float& g0, float& g1) {
                constexpr bool AFF = decltype(aff_tag)::value;
                const float X2 = prefix_sum<L>(B20 * (u0 - u1) + (AFF ? d2 : 0.0f), j) + (AFF ? Dx2 : 0.0f); // psi at node j+1
                const float in2 = shift_up1<L>(X2, AFF ? Dx2 : 0.0f, j);
                const float X0 = prefix_sum<L>(a * in2 + B00 * u0 + B01 * u1 + (AFF ? d0 : 0.0f), j) + (AFF ? Dx0 : 0.0f);
                const float X1 = prefix_sum<L>(b * in2 + B10 * u0 + B11 * u1 + (AFF ? d1 : 0.0f), j) + (AFF ? Dx1 : 0.0f);
                const float y0 = Q00 * X0 + Q01 * X1 + Q02 * X2 + (AFF ? qv0 : 0.0f);
                const float y1 = Q10 * X0 + Q11 * X1 + Q12 * X2 + (AFF ? qv1 : 0.0f);
                const float y2 = Q20 * X0 + Q21 * X1 + Q22 * X2 + (AFF ? qv2 : 0.0f);
                const float Lx = suffix_sum<L>(y0, j), Ly = suffix_sum<L>(y1, j); // adjoint at node j+1
                const float nx = shift_down1<L>(Lx, 0.0f, j), ny = shift_down1<L>(Ly, 0.0f, j);
                const float Lp = suffix_sum<L>(y2 + an * nx + bn * ny, j);
                g0 = R00 * u0 + R01 * u1 + (AFF ? r0 : 0.0f) + B00 * Lx + B10 * Ly + B20 * Lp;
                g1 = R01 * u0 + R11 * u1 + (AFF ? r1 : 0.0f) + B01 * Lx + B11 * Ly - B20 * Lp;
            };
            // gradient at du = 0 and the clipped R-scaled (Jacobi) step
            const float is0 = in ? rcp_f(fmaxf(R00, 1e-20f)) : 0.0f, is1 = in ? rcp_f(fmaxf(R11, 1e-20f)) : 0.0f;
            float g0, g1;
            apply(0.0f, 0.0f, std::true_type{}, g0, g1);
            const float j0 = -is0 * g0, j1 = -is1 * g1;
            const int hits = (j0 < lb0) | (j0 > ub0) | (j1 < lb1) | (j1 > ub1);
            // a control weight that is not positive: no prediction, the all-free first sweep then reports the
            // indefinite Hessian exactly where the reference's initial Cholesky does (status 31)
            const int badw = in ? ((!(R00 > 0.0f)) | (!(R11 > 0.0f))) : 0;
            const bool run = cold && group_any<L>(hits != 0, lane - j) && !group_any<L>(badw != 0, lane - j);
            if (__any(run)) {
                // projected Barzilai-Borwein iteration in the R-scaled metric: the first step is the clipped
                // Jacobi step (unit step length: the scaled Hessian is I + positive semidefinite, so the BB
                // lengths s's / s'y lie in (0, 1]); no eigenvalue estimate is needed
                float u0 = clampf(j0, lb0, ub0), u1 = clampf(j1, lb1, ub1);
                float pu0 = 0.0f, pu1 = 0.0f, pg0 = g0, pg1 = g1, alpha = 1.0f;
                // pg_steps steps, then up to half as many again while the predicted set of the problem still moves
                const int max_steps = p.pg_steps + p.pg_steps / 2;
                auto at_bounds = [&](float a0_, float a1_) {
                    return (a0_ <= lb0 ? 1 : 0) | (a0_ >= ub0 ? 2 : 0) | (a1_ <= lb1 ? 4 : 0) | (a1_ >= ub1 ? 8 : 0);
                };
                int bits = at_bounds(u0, u1), still = 0;
                bool frozen = !run; // per group: its prediction has settled (so that the path of a problem never depends on its wavefront mates)
#pragma unroll 1
                for (int t = 1; t < max_steps; ++t) {
                    apply(u0, u1, std::true_type{}, g0, g1);
                    const float du0 = u0 - pu0, du1 = u1 - pu1, dg0 = g0 - pg0, dg1 = g1 - pg1;
                    const float num = group_total<L>(R00 * du0 * du0 + R11 * du1 * du1, j);
                    const float den = group_total<L>(du0 * dg0 + du1 * dg1, j);
                    alpha = (den > 1e-30f) ? fminf(fmaxf(num * __builtin_amdgcn_rcpf(den), 1e-3f), 1.0f) : alpha;
                    pu0 = u0; pu1 = u1; pg0 = g0; pg1 = g1;
                    const float n0 = __builtin_amdgcn_fmed3f(u0 - alpha * is0 * g0, lb0, ub0);
                    const float n1 = __builtin_amdgcn_fmed3f(u1 - alpha * is1 * g1, lb1, ub1);
                    u0 = frozen ? u0 : n0;
                    u1 = frozen ? u1 : n1;
                    const int nb = at_bounds(u0, u1);
                    still = group_any<L>(in && nb != bits, lane - j) ? 0 : still + 1;
                    bits = nb;
                    // never fewer than pg_steps; more only while the set of this group still moves
                    frozen = frozen || (still >= 2 && t + 1 >= p.pg_steps);
                    if (__all(frozen)) break;
                }
                if (in && run) {
                    const int n0 = (ub0 - lb0 > BOUNDTOL) ? ((u0 <= lb0) ? ST_LOWER : ((u0 >= ub0) ? ST_UPPER : ST_FREE)) : ST_LOWER;
                    const int n1 = (ub1 - lb1 > BOUNDTOL) ? ((u1 <= lb1) ? ST_LOWER : ((u1 >= ub1) ? ST_UPPER : ST_FREE)) : ST_LOWER;
                    *reinterpret_cast<float2*>(rec + S_STDU * 4) = make_float2(__int_as_float(n0), __int_as_float(n1));
                    write_fix(rec, n0, n1, lb0, ub0, lb1, ub1);
                }
            }
            wave_sync();
            if (STAMP) t_pg = __builtin_amdgcn_s_memtime() - tp0;
        }

        // ---- phase B: working-set iterations
        int pd_fail = 0;
        bool changed = true; // "this problem still needs a sweep"
        int khi = N - 1;     // highest stage whose cost-to-go is stale
        int it = 0;
        n_iter = 0;
        long long t_b = 0, t_f = 0;
        for (;;) {
            long long tb0 = 0;
            if (STAMP) tb0 = __builtin_amdgcn_s_memtime();
            // ---- backward Riccati sweep from the highest stale stage of this wavefront
            const int kk = changed ? khi : -1;
            int kmax_v; // wavefront-uniform loop bound
            if constexpr (L == 64) kmax_v = kk;
            else if constexpr (L == 32) kmax_v = max(__float_as_int(readlane_f(__int_as_float(kk), 31)), __float_as_int(readlane_f(__int_as_float(kk), 63)));
            else kmax_v = wave_max(kk);
            // The loop below handles two stages per trip with two inlined copies of the step whose roundings
            // may differ (FMA contraction); a restart keeps the parity of the full sweep so that a stage is
            // always processed by the same copy and a problem's result does not depend on where its
            // wavefront restarts (i.e. on its wavefront mates).
            const int kraw = __builtin_amdgcn_readfirstlane(kmax_v);
            const int kmax = (kraw < 0) ? -1 : kraw + ((N - 1 - kraw) & 1);
            {
                // the sequential sweep is the longest dependent chain of a wavefront: it wins the SIMD's issue
                // arbitration against a mate that is in a stage-parallel phase (measured: -0.8 us per B=4096 launch)
                __builtin_amdgcn_s_setprio(3);
                RowValue V;
                V.P0 = V.P1 = V.P2 = V.p = 0.0f;
                int ok = 1;
                // Every group that still iterates sweeps from the highest stale stage of the wavefront:
                // above its own stale stage it recomputes, bit for bit, what its records already hold.  Lanes 0..2 of the
                // first quad store their row / components, lane 3 the three group-uniform scalars, in the
                // same instructions.
                auto step = [&](int k, const StageBcast& cur, const float4& qrow, const float2& brow) {
                    RowPolicy pol;
                    ok &= riccati_rows(cur, qrow, brow.x, brow.y, is2, V, pol, k > 0);
                    if (j < 4 && changed) { // a settled group keeps its records (and its multipliers)
                        float* rec = row + k * SR;
                        rec[S_POL0 * 4 + j] = (j < 3) ? pol.c0c : pol.f0;
                        rec[S_POL1 * 4 + j] = (j < 3) ? pol.c1c : pol.e1;
                        // lane 3 parks f1 in the multiplier slot (mu is rewritten by the forward sweep)
                        const int slot = (j < 3) ? S_V + j : S_MU;
                        const float w0_ = (j < 3) ? V.P0 : 0.0f, w1_ = (j < 3) ? V.P1 : 0.0f,
                                    w2_ = (j < 3) ? V.P2 : pol.f1, w3_ = (j < 3) ? V.p : 0.0f;
                        st4(rec, slot, w0_, w1_, w2_, w3_);
                    }
                };
                auto fetch = [&](int k, StageBcast& sb, float4& qrow, float2& brow) {
                    const float* rec = row + k * SR;
                    load_bcast(rec, sb);
                    qrow = lds4(rec, S_Q + rr);
                    brow = *reinterpret_cast<const float2*>(rec + 2 * rr);
                };
                // two register sets, loads of the next record in flight while the current one is used
                StageBcast sa, sb;
                float4 qa = make_float4(0, 0, 0, 0), qb = qa;
                float2 ba = make_float2(0, 0), bb = ba;
                int k = kmax;
                if (k >= 0) {
                    const float4 v0 = lds4(row + (k + 1) * SR, S_V + rr); // cost-to-go of node kmax + 1
                    V.P0 = v0.x; V.P1 = v0.y; V.P2 = v0.z; V.p = v0.w;
                    fetch(k, sa, qa, ba);
                }
                while (k >= 1) {
                    fetch(k - 1, sb, qb, bb);
                    step(k, sa, qa, ba);
                    if (k >= 2) fetch(k - 2, sa, qa, ba);
                    step(k - 1, sb, qb, bb);
                    k -= 2;
                }
                if (k == 0) step(0, sa, qa, ba);
                pd_fail |= (ok == 0);
                __builtin_amdgcn_s_setprio(0);
            }
            wave_sync();
            long long tf0 = 0;
            if (STAMP) { tf0 = __builtin_amdgcn_s_memtime(); t_b += tf0 - tb0; }

            // ---- forward sweep as a prefix scan, stages in chunks of L (lane j <-> stage base + j)
            const bool first = (it == 0);
            const bool active = changed;
            int new_khi = -1;
            if (__any(active)) {
                float cx0 = Dx0, cx1 = Dx1, cx2 = Dx2; // state step entering the chunk
                float cs0 = Dx0, cs1 = Dx1, cs2 = Dx2; // free response entering the chunk (first sweep)
                for (int base = 0; base < N; base += L) {
                    const int k = base + j;
                    const bool in = k < N;
                    float* rec = row + (in ? k : 0) * SR;
                    const float4 l0 = lds4(rec, S_B0), l1 = lds4(rec, S_B1), dd = lds4(rec, S_D),
                                 bnd = lds4(rec, S_BND), sd = lds4(rec, S_STDU), p0 = lds4(rec, S_POL0),
                                 p1 = lds4(rec, S_POL1), mu = lds4(rec, S_MU);
                    const int st0 = __float_as_int(sd.x), st1 = __float_as_int(sd.y);
                    const bool f0 = (st0 == ST_FREE), f1 = (st1 == ST_FREE);
                    const float b0 = (st0 == ST_UPPER) ? bnd.y : bnd.x, b1 = (st1 == ST_UPPER) ? bnd.w : bnd.z;
                    // du0 = g0.dx + h0 ; du1 = g1.dx + h1  under the current working set
                    const float g00 = f0 ? p0.x : 0.0f, g01 = f0 ? p0.y : 0.0f, g02 = f0 ? p0.z : 0.0f;
                    const float h0 = f0 ? p0.w : b0;
                    const float g10 = f1 ? p1.x + p1.w * g00 : 0.0f, g11 = f1 ? p1.y + p1.w * g01 : 0.0f,
                                g12 = f1 ? p1.z + p1.w * g02 : 0.0f;
                    const float h1 = f1 ? p1.w * h0 + mu.z : b1;
                    const float B00 = l0.x, B01 = l0.y, B10 = l0.z, B11 = l0.w, B20 = l1.x, a = l1.z, b = l1.w;
                    const float d0 = dd.x, d1 = dd.y, d2 = dd.z;
                    // closed-loop stage map  dx+ = M dx + c  (identity for padding lanes)
                    AffineMap f;
                    f.m00 = 1.0f; f.m01 = 0.0f; f.m02 = 0.0f; f.m10 = 0.0f; f.m11 = 1.0f; f.m12 = 0.0f;
                    f.m20 = 0.0f; f.m21 = 0.0f; f.m22 = 1.0f; f.c0 = 0.0f; f.c1 = 0.0f; f.c2 = 0.0f;
                    // free-response map: sbar+ = A sbar + d
                    float sa_ = 0.0f, sb_ = 0.0f, sc0 = 0.0f, sc1 = 0.0f, sc2 = 0.0f;
                    if (in) {
                        const float gd0 = g00 - g10, gd1 = g01 - g11, gd2 = g02 - g12; // B2 = B20 * (1, -1)
                        f.m00 = 1.0f + B00 * g00 + B01 * g10; f.m01 = B00 * g01 + B01 * g11; f.m02 = a + B00 * g02 + B01 * g12;
                        f.m10 = B10 * g00 + B11 * g10; f.m11 = 1.0f + B10 * g01 + B11 * g11; f.m12 = b + B10 * g02 + B11 * g12;
                        f.m20 = B20 * gd0; f.m21 = B20 * gd1; f.m22 = 1.0f + B20 * gd2;
                        f.c0 = B00 * h0 + B01 * h1 + d0;
                        f.c1 = B10 * h0 + B11 * h1 + d1;
                        f.c2 = B20 * (h0 - h1) + d2;
                        sa_ = a; sb_ = b; sc0 = d0; sc1 = d1; sc2 = d2;
                    }
                    // inclusive scan of map composition: after it, lane j holds f_k o ... o f_base
                    scan_maps<L>(f, j);
                    // state step leaving stage k (= entering k+1), and the one entering stage k
                    const float o0 = f.m00 * cx0 + f.m01 * cx1 + f.m02 * cx2 + f.c0;
                    const float o1 = f.m10 * cx0 + f.m11 * cx1 + f.m12 * cx2 + f.c1;
                    const float o2 = f.m20 * cx0 + f.m21 * cx1 + f.m22 * cx2 + f.c2;
                    const float dx0 = shift_up1<L>(o0, cx0, j), dx1 = shift_up1<L>(o1, cx1, j),
                                dx2 = shift_up1<L>(o2, cx2, j);
                    Policy pol;
                    pol.c00 = p0.x; pol.c01 = p0.y; pol.c02 = p0.z; pol.f0 = p0.w;
                    pol.c10 = p1.x; pol.c11 = p1.y; pol.c12 = p1.z; pol.e1 = p1.w; pol.f1 = mu.z;
                    StageStep o;
                    forward_step(pol, st0, st1, dx0, dx1, dx2, bnd.x, bnd.y, bnd.z, bnd.w, o);
                    if (in && ((o.nst0 != st0) || (o.nst1 != st1))) new_khi = max(new_khi, k);
                    if (in && active) {
                        *reinterpret_cast<float2*>(rec + S_MU * 4) = make_float2(o.mu0, o.mu1);
                        rec[S_DX * 4] = dx0; rec[S_DX * 4 + 1] = dx1; rec[S_DX * 4 + 2] = dx2;
                        st4(rec, S_STDU, __int_as_float(o.nst0), __int_as_float(o.nst1), o.du0, o.du1);
                        write_fix(rec, o.nst0, o.nst1, bnd.x, bnd.y, bnd.z, bnd.w);
                    }
                    if (first) { // free response (du = 0): A is a shear, so two chained prefix sums do it
                        const float e2 = prefix_sum<L>(sc2, j);            // sum_{i<=k} d2_i
                        const float in2 = cs2 + shift_up1<L>(e2, 0.0f, j); // sbar2 entering stage k
                        const float e0 = prefix_sum<L>(sa_ * in2 + sc0, j), e1 = prefix_sum<L>(sb_ * in2 + sc1, j);
                        const float in0 = cs0 + shift_up1<L>(e0, 0.0f, j), in1 = cs1 + shift_up1<L>(e1, 0.0f, j);
                        if (in && active) { rec[S_SB * 4] = in0; rec[S_SB * 4 + 1] = in1; rec[S_SB * 4 + 2] = in2; }
                        // carry to the next chunk: values leaving the last lane
                        cs0 += group_last<L>(e0); cs1 += group_last<L>(e1); cs2 += group_last<L>(e2);
                    }
                    cx0 = group_last<L>(o0); cx1 = group_last<L>(o1); cx2 = group_last<L>(o2);
                }
                if (writer && active) { // node N
                    float* rec = row + N * SR;
                    rec[S_DX * 4] = cx0; rec[S_DX * 4 + 1] = cx1; rec[S_DX * 4 + 2] = cx2;
                    if (first) { rec[S_SB * 4] = cs0; rec[S_SB * 4 + 1] = cs1; rec[S_SB * 4 + 2] = cs2; }
                }
                // highest changed stage of the group
                new_khi = group_max_i<L>(new_khi, j);
            }
            ++it;
            if (active) {
                changed = (new_khi >= 0);
                khi = new_khi;
                if (changed) n_iter = it;
            }
            if (STAMP) t_f += __builtin_amdgcn_s_memtime() - tf0;
            // wavefront-uniform continuation: any problem of this wavefront still changing?
            const int more = wave_or((changed && it < min(p.max_as_iter, AS_SWITCH)) ? 1 : 0);
            if (!more) break;
        }
        n_iter = (n_iter == 0) ? 1 : (changed ? n_iter : n_iter + 1); // + the confirming sweep
        // not settled after AS_SWITCH sweeps (the primal-dual iteration can cycle): finish with the primal
        // active-set method, one change of the working set per sweep
        {
            const bool rescue = changed && it >= AS_SWITCH && it < p.max_as_iter;
            if (__builtin_expect(__any(rescue) ? 1 : 0, 0)) { // cold: register allocation should not pay for it
                const int r = active_set_rescue<L>(row, N, j, rescue, Dx0, Dx1, Dx2, p.max_as_iter, it);
                if (rescue) {
                    changed = (r >> 30) & 1;
                    pd_fail |= (r >> 29) & 1;
                    it = r & 0xffffff;
                    n_iter = it;
                }
            }
        }
        status = infeasible ? RET_INIT_FAILED_INFEASIBILITY
                            : (pd_fail ? RET_INIT_FAILED_CHOLESKY : (changed ? RET_MAX_NWSR_REACHED : RET_OK));
        if (STAMP && sqp == 0) { t_stamp[2] = t_b; t_stamp[3] = t_f; t_stamp[4] = __builtin_amdgcn_s_memtime(); }
        if (STAMP && sqp == 0 && lane == 0 && p.stamps) p.stamps[((size_t)blockIdx.x * WPB + wave) * 8 + 6] = t_pg;

        // ---- phase C (stage-parallel): KKT value (acado_getKKT), expand (acado_expand), carry the dual
        float gd = 0.0f, comp = 0.0f;
        for (int k = j; k <= N; k += L) {
            float* rec = row + k * SR;
            const float4 dxp = lds4(rec, S_DX), sb = lds4(rec, S_SB), xk = lds4(rec, S_X);
            if (want_kkt && k > 0) { // (Q_k sbar_k + q_k)' (dx_k - sbar_k)
                const float4 q0 = lds4(rec, S_Q), q1 = lds4(rec, S_Q + 1), q2 = lds4(rec, S_Q + 2);
                const float t0 = dxp.x - sb.x, t1 = dxp.y - sb.y, t2 = dxp.z - sb.z;
                gd += (q0.x * sb.x + q0.y * sb.y + q0.z * sb.z + q0.w) * t0 +
                      (q1.x * sb.x + q1.y * sb.y + q1.z * sb.z + q1.w) * t1 +
                      (q2.x * sb.x + q2.y * sb.y + q2.z * sb.z + q2.w) * t2;
            }
            st4(rec, S_X, xk.x + dxp.x, xk.y + dxp.y, xk.z + dxp.z, 0.0f);
            if (k < N) {
                const float4 dd = lds4(rec, S_D), R = lds4(rec, S_R), bnd = lds4(rec, S_BND), sd = lds4(rec, S_STDU),
                             mu = lds4(rec, S_MU), uy = lds4(rec, S_UY);
                if (want_kkt) {
                    gd += dd.w * sd.z + R.w * sd.w; // r' du
                    comp += (mu.x > 1e-12f) ? fabsf(bnd.x * mu.x) : ((mu.x < -1e-12f) ? fabsf(bnd.y * mu.x) : 0.0f);
                    comp += (mu.y > 1e-12f) ? fabsf(bnd.z * mu.y) : ((mu.y < -1e-12f) ? fabsf(bnd.w * mu.y) : 0.0f);
                }
                // a free control may sit up to TOL_PRIMAL outside its box: keep the iterate feasible
                const float du0 = (bnd.x <= bnd.y) ? clampf(sd.z, bnd.x, bnd.y) : sd.z;
                const float du1 = (bnd.z <= bnd.w) ? clampf(sd.w, bnd.z, bnd.w) : sd.w;
                st4(rec, S_UY, uy.x + du0, uy.y + du1, mu.x, mu.y);
            }
        }
        if (want_kkt) kkt = fabsf(group_total<L>(gd, j)) + group_total<L>(comp, j);
        wave_sync();
    }
    if (STAMP) t_stamp[5] = __builtin_amdgcn_s_memtime();

    // ---- objective at the returned iterate (acado_getObjective) + write-back
    float part = 0.0f;
    for (int k = j; k <= N; k += L) {
        const float* rec = row + k * SR;
        const float4 xk = lds4(rec, S_X);
        if (k < N) {
            const float4 uy = lds4(rec, S_UY);
            if (want_obj) {
                const float* yk = stg + SG.y + k * 5;
                float Wk[25];
#pragma unroll
                for (int i = 0; i < 25; ++i) Wk[i] = WREG ? wreg[i] : stg[SG.W + k * 25 + i];
                float e[5] = {xk.x - yk[0], xk.y - yk[1], xk.z - yk[2], uy.x - yk[3], uy.y - yk[4]};
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const float t = e[0] * Wk[c] + e[1] * Wk[5 + c] + e[2] * Wk[10 + c] + e[3] * Wk[15 + c] +
                                    e[4] * Wk[20 + c];
                    acc += e[c] * t;
                }
                part += acc;
            }
            if (keep) {
                float* ou = p.b.u + (size_t)prob * N * 2;
                float* od = p.b.dual + (size_t)prob * N * 2;
                ou[k * 2] = uy.x; ou[k * 2 + 1] = uy.y;
                od[k * 2] = uy.z; od[k * 2 + 1] = uy.w;
            }
        } else { // the reference uses only the diagonal of WN here (acado_solver.c:1442-1444)
            if (want_obj) {
                const float* tn = stg + SG.term;
                const float e0 = xk.x - tn[9], e1 = xk.y - tn[10], e2 = xk.z - tn[11];
                part += e0 * e0 * tn[0] + e1 * e1 * tn[4] + e2 * e2 * tn[8];
            }
        }
        if (keep) {
            float* ox = p.b.x + (size_t)prob * (N + 1) * 3;
            ox[k * 3] = xk.x; ox[k * 3 + 1] = xk.y; ox[k * 3 + 2] = xk.z;
        }
    }
    float obj = 0.0f;
    if (want_obj) obj = 0.5f * group_total<L>(part, j);
    if (keep && writer) {
        p.b.status[prob] = status;
        p.b.n_iter[prob] = n_iter;
        if (want_kkt && p.b.kkt) p.b.kkt[prob] = kkt;
        if (want_obj && p.b.obj) p.b.obj[prob] = obj;
    }
    if (STAMP && lane == 0 && p.stamps) {
        long long* o = p.stamps + ((size_t)blockIdx.x * WPB + wave) * 8;
        const long long t_end = __builtin_amdgcn_s_memtime();
        o[0] = t_stamp[1] - t_stamp[0]; // load + phase A
        o[1] = t_stamp[2];              // backward sweeps
        o[2] = t_stamp[3];              // forward sweeps
        o[3] = t_stamp[5] - t_stamp[4]; // phase C
        o[4] = t_end - t_stamp[5];      // objective + write-back
        o[5] = t_end - t_stamp[0];      // total
    }
}

hipError_t launch_rti(const RtiParams& p, const LaunchGeom& g, hipStream_t s)
{
    dim3 grid(g.grid), block(g.threads);
    hipError_t e = hipSuccess;
    const bool stamp = p.stamps != nullptr;
    int dev = 0; // hipFuncSetAttribute applies to the current device: the cache is per device
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 15;
    switch (g.L) {
#define CASE(LL)                                                                                              \
    case LL: {                                                                                                \
        static size_t configured[16][16] = {{0}}; /* raise the dynamic-LDS cap once per (device, variant) */  \
        constexpr bool WR = (LL >= 32);                                                                       \
        const bool diag = stamp || p.b.kkt != nullptr || p.b.obj != nullptr;                                  \
        const int v = (stamp ? 1 : 0) + (g.wpb == 4 ? 2 : 0) + ((g.wreg && WR) ? 4 : 0) + (diag ? 0 : 8);    \
        const void* fn = nullptr;                                                                             \
        switch (v) {                                                                                          \
        case 0: fn = (const void*)rti_kernel<LL, false, 1, false, true>; break;                               \
        case 1: fn = (const void*)rti_kernel<LL, true, 1, false, true>; break;                                \
        case 2: fn = (const void*)rti_kernel<LL, false, 4, false, true>; break;                               \
        case 3: fn = (const void*)rti_kernel<LL, true, 4, false, true>; break;                                \
        case 4: fn = (const void*)rti_kernel<LL, false, 1, WR, true>; break;                                  \
        case 5: fn = (const void*)rti_kernel<LL, true, 1, WR, true>; break;                                   \
        case 6: fn = (const void*)rti_kernel<LL, false, 4, WR, true>; break;                                  \
        case 7: fn = (const void*)rti_kernel<LL, true, 4, WR, true>; break;                                   \
        case 8: fn = (const void*)rti_kernel<LL, false, 1, false, false>; break;                              \
        case 10: fn = (const void*)rti_kernel<LL, false, 4, false, false>; break;                             \
        case 12: fn = (const void*)rti_kernel<LL, false, 1, WR, false>; break;                                \
        default: fn = (const void*)rti_kernel<LL, false, 4, WR, false>; break;                                \
        }                                                                                                     \
        if (g.lds_bytes > configured[dev][v]) {                                                               \
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);        \
            if (e != hipSuccess) return e;                                                                    \
            configured[dev][v] = g.lds_bytes;                                                                 \
        }                                                                                                     \
        void* args[] = {const_cast<RtiParams*>(&p)};                                                          \
        e = hipLaunchKernel(fn, grid, block, args, g.lds_bytes, s);                                           \
        if (e != hipSuccess) return e;                                                                        \
        break;                                                                                                \
    }
        CASE(4)
        CASE(8)
        CASE(16)
        CASE(32)
        CASE(64)
#undef CASE
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// small stage-parallel kernels around the hot path (one thread per (problem, stage))

__global__ void linearize_kernel(alore_nmpc_batch b, int B, int N, IrkConst K, alore_nmpc_lin_out o)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * N) return;
    const int prob = (int)(t / N), k = (int)(t % N);
    const float* x = b.x + ((size_t)prob * (N + 1) + k) * 3;
    const float* u = b.u + ((size_t)prob * N + k) * 2;
    const float* od = b.od + ((size_t)prob * (N + 1) + k) * 3;
    StageLin lin;
    ddr_linearize(K, x[0], x[1], x[2], u[0], u[1], od[0], od[1], od[2], lin);
    if (o.d) {
        float* d = o.d + (size_t)t * 3;
        d[0] = lin.phi0 - x[3]; d[1] = lin.phi1 - x[4]; d[2] = lin.phi2 - x[5];
    }
    if (o.evGx) {
        float* g = o.evGx + (size_t)t * 9;
        g[0] = 1.f; g[1] = 0.f; g[2] = lin.a;
        g[3] = 0.f; g[4] = 1.f; g[5] = lin.b;
        g[6] = 0.f; g[7] = 0.f; g[8] = 1.f;
    }
    if (o.evGu) {
        float* g = o.evGu + (size_t)t * 6;
        g[0] = lin.B00; g[1] = lin.B01; g[2] = lin.B10; g[3] = lin.B11; g[4] = lin.B20; g[5] = -lin.B20;
    }
}

hipError_t launch_linearize(const alore_nmpc_batch& b, int B, int N, float dt, const alore_nmpc_lin_out& o,
                            hipStream_t s)
{
    const long total = (long)B * N;
    const int threads = 256;
    hipLaunchKernelGGL(linearize_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, s, b,
                       B, N, make_irk(dt), o);
    return hipGetLastError();
}

// sequential in k, one thread per problem (acado_initializeNodesByForwardSimulation)
__global__ void forward_sim_kernel(alore_nmpc_batch b, int B, int N, IrkConst K)
{
    const int prob = blockIdx.x * blockDim.x + threadIdx.x;
    if (prob >= B) return;
    float* x = b.x + (size_t)prob * (N + 1) * 3;
    const float* u = b.u + (size_t)prob * N * 2;
    const float* od = b.od + (size_t)prob * (N + 1) * 3;
    float px = x[0], py = x[1], ps = x[2];
    for (int k = 0; k < N; ++k) {
        StageLin lin;
        ddr_linearize(K, px, py, ps, u[k * 2], u[k * 2 + 1], od[k * 3], od[k * 3 + 1], od[k * 3 + 2], lin);
        px = lin.phi0; py = lin.phi1; ps = lin.phi2;
        x[(k + 1) * 3] = px; x[(k + 1) * 3 + 1] = py; x[(k + 1) * 3 + 2] = ps;
    }
}

hipError_t launch_forward_simulate(const alore_nmpc_batch& b, int B, int N, float dt, hipStream_t s)
{
    hipLaunchKernelGGL(forward_sim_kernel, dim3((B + 63) / 64), dim3(64), 0, s, b, B, N, make_irk(dt));
    return hipGetLastError();
}

// acado_shiftStates + acado_shiftControls, one thread per problem
__global__ void shift_kernel(alore_nmpc_batch b, int B, int N, IrkConst K, int strategy, const float* xEnd,
                             const float* uEnd)
{
    const int prob = blockIdx.x * blockDim.x + threadIdx.x;
    if (prob >= B) return;
    float* x = b.x + (size_t)prob * (N + 1) * 3;
    float* u = b.u + (size_t)prob * N * 2;
    const float* od = b.od + (size_t)prob * (N + 1) * 3;
    for (int i = 0; i < 3 * N; ++i) x[i] = x[i + 3];
    if (strategy == 1 && xEnd) {
        for (int i = 0; i < 3; ++i) x[3 * N + i] = xEnd[(size_t)prob * 3 + i];
    } else if (strategy == 2) {
        const float vr = uEnd ? uEnd[(size_t)prob * 2] : u[(N - 1) * 2];
        const float vl = uEnd ? uEnd[(size_t)prob * 2 + 1] : u[(N - 1) * 2 + 1];
        StageLin lin;
        ddr_linearize(K, x[3 * N], x[3 * N + 1], x[3 * N + 2], vr, vl, od[3 * N], od[3 * N + 1], od[3 * N + 2], lin);
        x[3 * N] = lin.phi0; x[3 * N + 1] = lin.phi1; x[3 * N + 2] = lin.phi2;
    }
    for (int i = 0; i < 2 * (N - 1); ++i) u[i] = u[i + 2];
    if (uEnd) {
        u[2 * (N - 1)] = uEnd[(size_t)prob * 2];
        u[2 * (N - 1) + 1] = uEnd[(size_t)prob * 2 + 1];
    }
}

hipError_t launch_shift(const alore_nmpc_batch& b, int B, int N, float dt, int strategy, const float* xEnd,
                        const float* uEnd, hipStream_t s)
{
    hipLaunchKernelGGL(shift_kernel, dim3((B + 63) / 64), dim3(64), 0, s, b, B, N, make_irk(dt), strategy, xEnd, uEnd);
    return hipGetLastError();
}

__global__ void fill_kernel(float* p, float v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

hipError_t launch_fill(float* p, float v, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}

} // namespace nmpc
