// nmpc_block_kernel.hip -- the "stage-block" form of the batched NMPC real-time iteration (gfx950).
//
// Why a second mapping.  nmpc::rti_kernel (nmpc_kernels.hip) gives a problem 32 lanes, one lane per stage, and runs the
// sequential Riccati sweep row-split over quads, 8-fold redundantly: 2240 vector instructions per problem, almost all
// of them links of dependent chains (DPP scans, the sweep), which a gfx950 SIMD issues at one per ~4.6 cycles however
// many wavefronts share it (profiles/r01_d_sq_counters.txt).  Here a problem gets L = 4, 8, 16 or 32 lanes and every lane
// OWNS S consecutive stages (L * S >= N) with all of their data in registers for the whole launch:
//   * stage-parallel phases (linearise + Gauss-Newton cost, working-set prediction, KKT/expand, objective) are plain
//     scalar code over the S stages of the lane -- S independent instruction streams per lane, so the in-order issue
//     always has independent work (4 cycles per instruction for a lone wavefront, 2 when two share a SIMD);
//   * the sequential sweeps go lane by lane: lane t runs the scalar Riccati recursion (nmpc_core.h, the same functions
//     the CPU harness strings together) over its S stages while the other lanes of the group are masked, then hands
//     the cost-to-go (9 floats) to lane t-1 with one DPP row shift per value; the forward sweep runs the other way.
//     A sweep costs N scalar stage steps per wavefront whatever L is, and a wavefront carries 64 / L problems: 16 at
//     L = 4 -- an eighth of the sweep instructions per problem of the wave-per-two-problems kernel;
//   * prefix / suffix sums of the prediction: serial inside a lane, log2(L) DPP steps across the group.
// LDS holds only what is read again at the end (W and y of the wavefront's problems, for acado_getObjective) and is
// the transposition buffer between the reference's per-problem layout in HBM (contiguous per problem, so a wavefront
// streams 64 / L consecutive problems with 16-byte-per-lane coalesced loads) and the lane-owns-stages registers.
// Global traffic is the algorithmic minimum as before.
//
// Builds: DIAG (KKT value and objective wanted), STAMP (phase stamps, diagnostic), ONCE (n_sqp = 1, the control tick: no
// iteration loop, so od / the raw bounds die after the linearisation and the iterate, WN, yN are read again at the end
// instead of held -- (4, 5) 435 registers instead of 512 + scratch, (16, 2) 252: two wavefronts per SIMD).
// Which (L, S) runs: block_geometry() below -- the widest L whose wavefronts, over ALL launches in flight
// (alore_nmpc_rti_many, nmpc_capi.hip), still fit one per SIMD.
//
// Numerics are those of nmpc_kernels.hip / nmpc_core.h (see there for the reference citations); per-problem results
// do not depend on the batch or on wavefront mates (masked lanes never feed a problem).  Different (L, S) agree to
// rounding only (summation order of the diagnostics and of the prediction).
#include "nmpc_kernels.h"

#include <cstdlib>

#include <type_traits>

#include "nmpc_core.h"

// -DALORE_PHASE_MARKERS (tools/asm_phases.py, never the shipped build): every phase boundary leaves `; @phase <name>` in the assembly
#ifdef ALORE_PHASE_MARKERS
#define PHASE(name) asm volatile("; @phase " name)
#else
#define PHASE(name)
#endif

namespace nmpc {
namespace {

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP move inside a row of 16 lanes; lanes without a source keep `old`
template <int CTRL>
__device__ __forceinline__ float dppk(float old, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dppk_i(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, 0xF, false);
}
// DPP move with a row mask: rows outside the mask keep `old`
template <int CTRL, int ROWS>
__device__ __forceinline__ float dppr(float old, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROWS, 0xF, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ int dppr_i(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWS, 0xF, false);
}
// lane i <- lane i + 1 / lane i - 1.  Groups of up to 16 lanes sit inside a DPP row (row_shl / row_shr: the lane at the
// end of a row keeps its value); a group of 32 spans two rows and uses the wavefront shifts of gfx9 (the lane at the end
// of a group then sees its neighbour group's value: every caller treats the edge lanes separately).
template <int L>
__device__ __forceinline__ float lane_next(float x)
{
    if constexpr (L == 32) return dppk<0x130>(x, x); // wave_shl:1
    else return dppk<0x101>(x, x);                    // row_shl:1
}
template <int L>
__device__ __forceinline__ float lane_prev(float x)
{
    if constexpr (L == 32) return dppk<0x138>(x, x); // wave_shr:1
    else return dppk<0x111>(x, x);                    // row_shr:1
}

// value of the first / last lane of the group of L lanes (L = 4: quad_perm; 8, 16: row_newbcast, gfx90a+; 32: the two
// group ends through scalar registers)
template <int L>
__device__ __forceinline__ float gfirst(float x, int lane)
{
    if constexpr (L == 4) return dppk<0x00>(x, x);
    else if constexpr (L == 16) return dppk<0x150>(x, x);
    else if constexpr (L == 32) {
        const int lo = __builtin_amdgcn_readlane(__float_as_int(x), 0), hi = __builtin_amdgcn_readlane(__float_as_int(x), 32);
        return __int_as_float((lane & 32) ? hi : lo);
    } else { const float lo = dppk<0x150>(x, x), hi = dppk<0x158>(x, x); return (lane & 8) ? hi : lo; }
}
template <int L>
__device__ __forceinline__ float glast(float x, int lane)
{
    if constexpr (L == 4) return dppk<0xFF>(x, x);
    else if constexpr (L == 16) return dppk<0x15F>(x, x);
    else if constexpr (L == 32) {
        const int lo = __builtin_amdgcn_readlane(__float_as_int(x), 31), hi = __builtin_amdgcn_readlane(__float_as_int(x), 63);
        return __int_as_float((lane & 32) ? hi : lo);
    } else { const float lo = dppk<0x157>(x, x), hi = dppk<0x15F>(x, x); return (lane & 8) ? hi : lo; }
}
// inclusive prefix sum over the lanes of a group (row_shr:1, 2, 4, 8; groups are aligned inside DPP rows; a group of 32
// adds the total of its first row to its second: row_bcast:15 into rows 1 and 3)
template <int L>
__device__ __forceinline__ float gprefix(float x, int j)
{
    float v;
    v = dppk<0x111>(0.0f, x); if (L < 16) v = (j >= 1) ? v : 0.0f; x += v;
    v = dppk<0x112>(0.0f, x); if (L < 16) v = (j >= 2) ? v : 0.0f; x += v;
    if constexpr (L >= 8) { v = dppk<0x114>(0.0f, x); if (L < 16) v = (j >= 4) ? v : 0.0f; x += v; }
    if constexpr (L >= 16) { v = dppk<0x118>(0.0f, x); x += v; }
    if constexpr (L >= 32) { v = dppr<0x142, 0xA>(0.0f, x); x += v; }
    return x;
}
template <int L>
__device__ __forceinline__ int gprefix_max(int x, int j)
{
    constexpr int NEG = -2147483647 - 1;
    int v;
    v = dppk_i<0x111>(NEG, x); if (L < 16) v = (j >= 1) ? v : NEG; x = max(x, v);
    v = dppk_i<0x112>(NEG, x); if (L < 16) v = (j >= 2) ? v : NEG; x = max(x, v);
    if constexpr (L >= 8) { v = dppk_i<0x114>(NEG, x); if (L < 16) v = (j >= 4) ? v : NEG; x = max(x, v); }
    if constexpr (L >= 16) { v = dppk_i<0x118>(NEG, x); x = max(x, v); }
    if constexpr (L >= 32) { v = dppr_i<0x142, 0xA>(NEG, x); x = max(x, v); }
    return x;
}
template <int L>
__device__ __forceinline__ float gtotal(float x, int j, int lane) { return glast<L>(gprefix<L>(x, j), lane); }
template <int L>
__device__ __forceinline__ int gmax(int x, int j, int lane)
{
    return __float_as_int(glast<L>(__int_as_float(gprefix_max<L>(x, j)), lane));
}
// lexicographic minimum of (v, key) over the lanes of a group, in every lane
template <int L>
__device__ __forceinline__ void gmin_pair(float& v, int& key, int j, int lane)
{
    auto step = [&](auto tag, auto rows, int dist) {
        constexpr int CTRL = decltype(tag)::value;
        constexpr int ROWS = decltype(rows)::value;
        const float pv = dppr<CTRL, ROWS>(v, v);
        const int pk = dppr_i<CTRL, ROWS>(key, key);
        const bool has = (L >= 16) || (j >= dist);
        const bool take = has && ((pv < v) || (pv == v && pk < key));
        v = take ? pv : v;
        key = take ? pk : key;
    };
    using all = std::integral_constant<int, 0xF>;
    step(std::integral_constant<int, 0x111>{}, all{}, 1);
    step(std::integral_constant<int, 0x112>{}, all{}, 2);
    if constexpr (L >= 8) step(std::integral_constant<int, 0x114>{}, all{}, 4);
    if constexpr (L >= 16) step(std::integral_constant<int, 0x118>{}, all{}, 8);
    if constexpr (L >= 32) step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{}, 16);
    v = glast<L>(v, lane);
    key = __float_as_int(glast<L>(__int_as_float(key), lane));
}
template <int L>
__device__ __forceinline__ bool gany(bool pred, int base)
{
    const unsigned long long m = __ballot(pred);
    constexpr unsigned long long MASK = (L >= 64) ? ~0ull : ((1ull << (L & 63)) - 1ull);
    return ((m >> base) & MASK) != 0ull;
}

// ---- global <-> LDS, coalesced: all loads of a launch are issued before the first LDS store ---------------------
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U>
__device__ __forceinline__ void g_issue(const float* g, int total, int lane, float4 (&q)[U], float& tail)
{
    const int n4 = total >> 2, rem = total & 3;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = u * 64 + lane;
        if (i < n4) {
            const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g) + i); // read once
            q[u] = make_float4(v.x, v.y, v.z, v.w);
        } else {
            q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    tail = (lane < rem) ? g[n4 * 4 + lane] : 0.0f; // only the ragged last wavefront has one
}
template <int U>
__device__ __forceinline__ void l_commit(float* l, int total, int lane, const float4 (&q)[U], float tail)
{
    const int n4 = total >> 2, rem = total & 3;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = u * 64 + lane;
        if (i < n4) reinterpret_cast<float4*>(l)[i] = q[u];
    }
    if (lane < rem) l[n4 * 4 + lane] = tail;
}
template <int U>
__device__ __forceinline__ void g_store(float* g, const float* l, int total, int lane)
{
    const int n4 = total >> 2, rem = total & 3;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = u * 64 + lane;
        if (i < n4) reinterpret_cast<float4*>(g)[i] = reinterpret_cast<const float4*>(l)[i];
    }
    if (lane < rem) g[n4 * 4 + lane] = l[n4 * 4 + lane];
}

// ---- packed float32 (v_pk_fma_f32 / v_pk_mul_f32, gfx90a+: two multiply-adds per issue slot of a wavefront) ------------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f bc2(float a) { v2f r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// lane-local transposes of register pairs (v_pk_mov_b32: low half of the result from src0, high half from src1, each taken
// from the half op_sel names; tools/micro/pk_mov_check.hip prints them)
__device__ __forceinline__ v2f hi_hi(v2f a, v2f b) { v2f r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f lo_lo(v2f a, v2f b) { v2f r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f hi_lo(v2f a, v2f b) { v2f r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }

// nmpc_core.h: riccati_step on register pairs.  The cost-to-go travels as Pa = (P00, P01), Pb = (P02, P12), Pc = (P11, P22),
// pa = (p0, p1), p2; the stage as B0 = (B00, B01), B1 = (B10, B11), B2 = (B20, -B20), Rd = (R00, R11), r = (r0, r1),
// QA = (Q00, Q01), QB = (Q02, Q12), QC = (Q11, Q22), qa = (q0, q1).  P d, P B, B' P B, B' P A, B' s, A' P A and the rank-one
// updates run as packed multiply-adds over the two inputs / over pairs of entries; the handful of lane-local transposes are
// single v_pk_mov_b32.  Same algebra as the scalar step, sums associate differently in places (float32 rounding).
struct ValuePk {
    v2f Pa, Pb, Pc, pa;
    float p2;
};
struct StagePk {
    v2f B0, B1, B2, Rd, r, QA, QB, QC, qa;
    float R01, a, b, d0, d1, d2, q2;
    int st0, st1;
    float v0, v1;
};
__device__ __forceinline__ bool riccati_step_pk(const StagePk& s, ValuePk& V, Policy& pol)
{
    const v2f Pa = V.Pa, Pb = V.Pb, Pc = V.Pc;
    const v2f Pd = hi_lo(Pa, Pc); // (P01, P11)
    // s = P d + p
    const v2f s01 = pfma(Pa, bc2(s.d0), pfma(Pd, bc2(s.d1), pfma(Pb, bc2(s.d2), V.pa)));
    const float s2 = Pb.x * s.d0 + Pb.y * s.d1 + Pc.y * s.d2 + V.p2;
    // rows of P B over the two inputs
    const v2f PB0 = pfma(bc2(Pa.x), s.B0, pfma(bc2(Pa.y), s.B1, bc2(Pb.x) * s.B2));
    const v2f PB1 = pfma(bc2(Pa.y), s.B0, pfma(bc2(Pc.x), s.B1, bc2(Pb.y) * s.B2));
    const v2f PB2 = pfma(bc2(Pb.x), s.B0, pfma(bc2(Pb.y), s.B1, bc2(Pc.y) * s.B2));
    const v2f Hd = pfma(s.B0, PB0, pfma(s.B1, PB1, pfma(s.B2, PB2, s.Rd)));  // (H00, H11)
    const float H01 = s.R01 + s.B0.x * PB0.y + s.B1.x * PB1.y + s.B2.x * PB2.y;
    const v2f G2 = pfma(bc2(s.a), PB0, pfma(bc2(s.b), PB1, PB2));            // (G02, G12)
    const v2f hu = pfma(s.B0, bc2(s01.x), pfma(s.B1, bc2(s01.y), pfma(s.B2, bc2(s2), s.r)));
    const v2f G1 = hi_hi(PB0, PB1);                                            // (G10, G11)
    v2f G0 = lo_lo(PB0, PB1);                                                  // (G00, G01)
    // ---- eliminate input 1
    const float H11 = Hd.y, hu1 = hu.y;
    const bool free1 = (s.st1 == ST_FREE);
    const bool bad1 = free1 && !(H11 > 0.0f);
    const float inv11 = pivot_rcp(H11);
    const float w1 = free1 ? inv11 : 0.0f;
    const float z1 = free1 ? -hu1 * inv11 : s.v1;
    const float t1 = w1 * H01;
    const float g1s = free1 ? -w1 : 1.0f;
    const v2f c1 = bc2(g1s) * G1;
    pol.c10 = c1.x; pol.c11 = c1.y; pol.c12 = g1s * G2.y; pol.e1 = g1s * H01;
    pol.f1 = free1 ? z1 : hu1 + H11 * s.v1;
    const float H00r = Hd.x - t1 * H01;
    G0 = pfma(-bc2(t1), G1, G0);                       // reduced (G00, G01)
    const v2f G2r = pfma(-bc2(t1), bc2(G2.y), G2);     // .x = reduced G02
    const float hu0 = hu.x + H01 * z1;
    // ---- eliminate input 0
    const bool free0 = (s.st0 == ST_FREE);
    const bool bad0 = free0 && !(H00r > 0.0f);
    const float inv00 = pivot_rcp(H00r);
    const float w0 = free0 ? inv00 : 0.0f;
    const float z0 = free0 ? -hu0 * inv00 : s.v0;
    const float g0s = free0 ? -w0 : 1.0f;
    const v2f c0 = bc2(g0s) * G0;
    pol.c00 = c0.x; pol.c01 = c0.y; pol.c02 = g0s * G2r.x;
    pol.f0 = free0 ? z0 : hu0 + H00r * s.v0;
    // ---- Hxx = Q + A' P A, hx = q + A' s
    const v2f mb = pfma(bc2(s.a), Pa, pfma(bc2(s.b), Pd, Pb));  // (P A)(0..1, 2): a P00 + b P01 + P02, a P01 + b P11 + P12
    v2f Xa = s.QA + Pa;
    v2f Xb = s.QB + mb;
    v2f Xc = s.QC + Pc;
    Xc.y += s.a * mb.x + s.b * mb.y + s.a * Pb.x + s.b * Pb.y;  // + a (PA)02 + b (PA)12 + (a P02 + b P12): the rest of (A' P A)22
    v2f hxa = s.qa + s01;
    float hx2 = s.q2 + (s.a * s01.x + s.b * s01.y + s2);
    { // input 1 out:  Hxx -= w1 G1' G1,  hx += G1' z1   (G1 = (G10, G11, G12))
        const v2f Gb = hi_hi(PB1, G2);                 // (G11, G12)
        const v2f wga = bc2(w1) * G1, wgb = bc2(w1) * Gb;
        Xa = pfma(-bc2(wga.x), G1, Xa);
        Xb = pfma(-wga, bc2(G2.y), Xb);
        Xc = pfma(-wgb, Gb, Xc);
        hxa = pfma(G1, bc2(z1), hxa);
        hx2 += G2.y * z1;
    }
    { // input 0 out (the reduced row)
        const v2f Gb = hi_lo(G0, G2r);                 // (G01, G02)
        const v2f wga = bc2(w0) * G0, wgb = bc2(w0) * Gb;
        Xa = pfma(-bc2(wga.x), G0, Xa);
        Xb = pfma(-wga, bc2(G2r.x), Xb);
        Xc = pfma(-wgb, Gb, Xc);
        hxa = pfma(G0, bc2(z0), hxa);
        hx2 += G2r.x * z0;
    }
    V.Pa = Xa; V.Pb = Xb; V.Pc = Xc; V.pa = hxa; V.p2 = hx2;
    return !(bad0 || bad1);
}

} // namespace

// LDS floats of one wavefront: W and y of its 64 / L problems, each area padded to whole 256-float DMA pieces
int block_lds_floats(int N, int L)
{
    const int G = 64 / L;
    return ((G * 25 * N + 255) & ~255) + ((G * 5 * N + 255) & ~255);
}

// One grid serves up to GROUP_MAX independent batches (alore_nmpc_rti_many): the descriptors travel by value in the kernel
// arguments, block -> (batch, block of the batch) by one division.  The pointers of a workgroup's batch are
// wavefront-uniform (scalar loads from the kernel-argument segment); a single batch is the group of one.
// FULLN: the horizon is exactly L * S (N = 20 on (4, 5)): every slot of every lane is a stage, the horizon is a compile-time
// constant and all the masking of neutral slots folds away (a twentieth of the instructions of the (4, 5) build).
// TRACE (diagnostic instantiation of the FULLN build, ALORE_NMPC_TRACE=<file>): every workgroup leaves the 100 MHz real-time
// counter at its start, after the staggered wait, when its inputs have landed, at its last store and when the stores are
// acknowledged, with the SIMD it ran on (tools/trace_timeline.py turns that into per-SIMD timelines of a grid).
// PERSIST (grid builds): the grid is one workgroup per SIMD slot and every workgroup takes blocks of 64 / L problems ("items")
// from a ticket counter in device memory until none is left -- the hardware deals the workgroups of a plain grid to the XCDs and
// their shader engines in fixed shares, so the grid lasts as long as its slowest XCD (they differ by ~5 % in clock under this
// load: profiles/r05_a_timeline_*.txt); with tickets a faster XCD takes more items.  The ticket of the next item is requested
// when the current one starts, so its latency is never waited for.
template <int L, int S, bool DIAG, bool STAMP, bool ONCE, bool FULLN = false, bool TRACE = false, bool PERSIST = false>
__global__ __launch_bounds__(64) void rti_block_kernel(const RtiParams p_arg, const RtiGroup grp_arg)
{
    extern __shared__ float4 lds_raw[];
    float* lds = reinterpret_cast<float*>(lds_raw);
    constexpr int G = 64 / L;
    constexpr int NMAX = L * S;
    constexpr bool MASKED = (L == 16 && S == 2) || (L == 32 && S == 1); // see backward_sweep
    const int N = FULLN ? L * S : p_arg.N;
    int item = (int)blockIdx.x; // block of G problems this workgroup works on
    // XCD shares (grid builds, RtiGroup::xcd_on).  The hardware deals the workgroups of a grid to the eight XCDs round-robin -- workgroup w
    // runs on XCD w mod 8, each XCD gets the same number -- and the XCDs of a part differ by several per cent in speed under this load,
    // so a grid lasts as long as its slowest XCD.  With shares, XCD x works on xcd_share[x] consecutive blocks from xcd_base[x]: the grid
    // is launched with 8 x max(share) workgroups, those beyond their XCD's share leave at once, and the shares follow the finishing times
    // the last workgroups of every XCD left in host memory at the previous launches (nmpc_capi.hip: rti_group).  Results do not depend
    // on which workgroup solves a block.
    if constexpr (FULLN && !PERSIST) {
        const auto& g0 = *reinterpret_cast<const __attribute__((address_space(4))) RtiGroup*>(
            (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(RtiParams) + 7) & ~(size_t)7));
        if (g0.xcd_on) {
            const int x = (int)blockIdx.x & 7, xcd_k = (int)blockIdx.x >> 3;
            if (blockIdx.x == 0 && threadIdx.x == 0 && g0.xcd_end != nullptr) g0.xcd_end[32] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); // the grid's start
            if (xcd_k >= g0.xcd_share[x]) return;
            item = g0.xcd_base[x] + xcd_k;
        }
    }
next_item:
    int lane = threadIdx.x;
    // the kernel arguments, read through the kernel-argument segment pointer (what `p_arg`, `grp_arg` are)
    typedef const __attribute__((address_space(4))) char* karg_ptr;
    karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    // persistent loop: everything derived from the lane number and from the kernel arguments would be hoisted out of the loop and
    // kept live across the whole body (hundreds of bytes of scratch, a register full of spilled scalars); opaque to the
    // optimiser they are recomputed / loaded again per item like in the plain grid
    if constexpr (PERSIST) {
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(ka));
    }
    const auto& p = *reinterpret_cast<const __attribute__((address_space(4))) RtiParams*>(ka);
    const auto& grp = *reinterpret_cast<const __attribute__((address_space(4))) RtiGroup*>(ka + ((sizeof(RtiParams) + 7) & ~(size_t)7));
    (void)p_arg; (void)grp_arg;
    const int g = lane / L, j = lane % L;
    const int gbase = lane - j;
    int ticket = 0;
    if constexpr (PERSIST) {
        if (lane == 0) ticket = atomicAdd(grp.counter, 1);
    }
    const int bi = (grp.count > 1) ? item / grp.blocks_per_batch : 0;
    // the workgroup's batch: entry bi of the table, or (batches laid out at constant strides, any number of them) the first
    // batch with every member pointer advanced by bi strides -- wavefront-uniform either way
    // Straight-line scalar code: the fifteen pointers and strides come in a few wide scalar loads with ONE wait (a test per member
    // made every load its own round trip: two dozen of them in a row, ~1 us before a wavefront issued its first load).  A member
    // that is absent has stride 0 (the host sets it), so null stays null without a test.
    alore_nmpc_batch pb;
    {
        char** q = reinterpret_cast<char**>(&pb);
        const auto* src = reinterpret_cast<char* const __attribute__((address_space(4)))*>(&grp.b[grp.strided ? 0 : bi]);
        const long long sbi = grp.strided ? (long long)bi : 0ll;
#pragma unroll
        for (int i = 0; i < 15; ++i) q[i] = src[i] + sbi * grp.stride[i];
    }
    const int prob0 = (item - bi * grp.blocks_per_batch) * G;
    const int np_ = min(G, p.B - prob0);
    const bool valid = g < np_;
    const int ge = valid ? g : np_ - 1; // padding groups shadow the last problem, never store
    const int prob = prob0 + ge;
    const int nx = 3 * (N + 1), nu = 2 * N;
    const int gw = (p.shared & ALORE_NMPC_SHARED_W) ? 0 : ge; // whose copy of W this group reads
    const int top = (N - 1) / S;        // lane that owns stage N - 1 (and the terminal node)

    // Staggered start.  A grid of several residencies (alore_nmpc_rti_many) begins with every SIMD loading at once: 1024
    // wavefronts x 67 KB is a 14 us burst during which nothing computes, and wavefronts that start together finish together,
    // so the next residency bursts again (20 batches: five rounds of 35 us against 25 us per round in a long run whose
    // wavefronts have drifted apart).  The wavefronts of the FIRST residency therefore start spread over the time HBM needs
    // to feed them -- they would have waited for their data that long anyway -- and the rounds never line up.
    if constexpr (TRACE) {
        if (grp.trace && threadIdx.x == 0) {
            long long* o = grp.trace + (size_t)item * 8;
            o[0] = (long long)__builtin_amdgcn_s_memrealtime();
            o[5] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); // HW_ID, XCC_ID
        }
    }
    if (grp.stagger_x1024 > 0 && (PERSIST ? item == (int)blockIdx.x : true) && (int)blockIdx.x < grp.stagger_blocks) {
        const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = ((unsigned long long)blockIdx.x * (unsigned)grp.stagger_x1024) >> 10;
        while (__builtin_amdgcn_s_memrealtime() - ts < wait) __builtin_amdgcn_s_sleep(2);
    }
    long long t0 = 0, t1 = 0, t4 = 0, t5 = 0, t_b = 0, t_f = 0, t_pg = 0;
    if (STAMP) t0 = __builtin_amdgcn_s_memtime();
    if constexpr (TRACE) {
        if (grp.trace && threadIdx.x == 0) grp.trace[(size_t)item * 8 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }

    IrkConst K;
    K.h = p.h; K.hh = p.hh; K.c1h = p.c1h; K.c2h = p.c2h;

    // LDS map (floats): [W | y] of the wavefront's problems, each area a whole number of 256-float DMA pieces; after
    // the objective the W area is the staging buffer of the outputs
    const int WA = (G * 25 * N + 255) & ~255, oW = 0, oY = WA;
    const int oX = 0, oU = G * nx, oDL = oU + G * nu;

    PHASE("load");
    // ---- phase 0.  W and y (read again for the objective) go HBM -> LDS by DMA, no registers: the wavefront's
    //      problems are contiguous, one instruction moves 1 KB.  The iterate, od, the bounds and the dual are read by
    //      the lane that owns the stage straight into its registers (12- and 8-byte pieces of one contiguous span).
    {
        constexpr int UW = (G * 25 * NMAX + 255) / 256, UY = (G * 5 * NMAX + 255) / 256;
        auto dma = [&](const float* gsrc, int total, int lds_off, auto utag) {
            constexpr int U = decltype(utag)::value;
            const int n4 = total >> 2, rem = total & 3;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u * 64 < n4) { // wavefront-uniform
                    const int i = min(u * 64 + lane, n4 - 1); // lanes past the end re-read the last piece (their LDS slot is padding)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gsrc + 4 * i),
                                                     (void __attribute__((address_space(3)))*)(lds + lds_off + u * 256), 16, 0, 0);
                }
            }
            if (rem && lane < rem) lds[lds_off + n4 * 4 + lane] = gsrc[n4 * 4 + lane]; // ragged last wavefront only
        };
        if (p.shared & ALORE_NMPC_SHARED_W) dma(pb.W, 25 * N, oW, std::integral_constant<int, UW>{}); // one copy for the batch
        else dma(pb.W + (size_t)prob0 * 25 * N, np_ * 25 * N, oW, std::integral_constant<int, UW>{});
        dma(pb.y + (size_t)prob0 * 5 * N, np_ * 5 * N, oY, std::integral_constant<int, UY>{});
    }
    float x[S][3], u[S][2], od[S][3], lbv[S][2], ubv[S][2], xN[3];
    float mu0[S], mu1[S]; // bound multipliers: the incoming dual until the first forward sweep overwrites it
    float x00, x01, x02, WN[9], yN[3];
    {
        typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        const float* gx = pb.x + (size_t)prob * nx;
        const float* god = pb.od + ((p.shared & ALORE_NMPC_SHARED_OD) ? 0 : (size_t)prob * nx);
        const float* gu = pb.u + (size_t)prob * nu;
        const float* gdl = pb.dual + (size_t)prob * nu;
        const float* glb = pb.lbValues + ((p.shared & ALORE_NMPC_SHARED_BOUNDS) ? 0 : (size_t)prob * nu);
        const float* gub = pb.ubValues + ((p.shared & ALORE_NMPC_SHARED_BOUNDS) ? 0 : (size_t)prob * nu);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s;
            const int kn = min(k, N), kc = min(k, N - 1);
            const f3u vx = *reinterpret_cast<const f3u*>(gx + 3 * kn), vo = *reinterpret_cast<const f3u*>(god + 3 * kn);
            const f2u vu = *reinterpret_cast<const f2u*>(gu + 2 * kc), vd = *reinterpret_cast<const f2u*>(gdl + 2 * kc),
                      vl = *reinterpret_cast<const f2u*>(glb + 2 * kc), vh = *reinterpret_cast<const f2u*>(gub + 2 * kc);
            x[s][0] = vx.x; x[s][1] = vx.y; x[s][2] = vx.z; od[s][0] = vo.x; od[s][1] = vo.y; od[s][2] = vo.z;
            u[s][0] = vu.x; u[s][1] = vu.y; mu0[s] = vd.x; mu1[s] = vd.y;
            lbv[s][0] = vl.x; lbv[s][1] = vl.y; ubv[s][0] = vh.x; ubv[s][1] = vh.y;
        }
        const f3u vn = *reinterpret_cast<const f3u*>(gx + 3 * N);
        xN[0] = vn.x; xN[1] = vn.y; xN[2] = vn.z;
        x00 = pb.x0[(size_t)prob * 3]; x01 = pb.x0[(size_t)prob * 3 + 1]; x02 = pb.x0[(size_t)prob * 3 + 2];
#pragma unroll
        for (int i = 0; i < 9; ++i) WN[i] = pb.WN[((p.shared & ALORE_NMPC_SHARED_W) ? 0 : (size_t)prob * 9) + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) yN[i] = pb.yN[(size_t)prob * 3 + i];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the DMA pieces have landed
    wave_sync();
    if constexpr (PERSIST) ticket = __builtin_amdgcn_readfirstlane(ticket); // the ticket has come back with the loads: into a scalar register
    if constexpr (TRACE) {
        if (grp.trace && threadIdx.x == 0) grp.trace[(size_t)item * 8 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    }

    // problems the caller masked out (alore_nmpc_set_problem_mask) run along on whatever their members hold and write nothing
    // (see the end of the kernel).  Their references may be stale or non-finite: they are replaced by the iterate itself (zero
    // tracking error), so that such a problem cannot keep its wavefront in the working-set loop up to max_as_iter.  Cold path.
    if (p.mask != nullptr) {
        const bool sit_out = valid && p.mask[prob] == 0;
        if (__any(sit_out)) {
            if (sit_out) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int k = j * S + s;
                    if (k < N) {
                        float* yk = lds + oY + ge * 5 * N + 5 * k;
                        yk[0] = x[s][0]; yk[1] = x[s][1]; yk[2] = x[s][2]; yk[3] = u[s][0]; yk[4] = u[s][1];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) yN[c] = sit_out ? xN[c] : yN[c];
            const float f0 = gfirst<L>(x[0][0], lane), f1 = gfirst<L>(x[0][1], lane), f2 = gfirst<L>(x[0][2], lane);
            x00 = sit_out ? f0 : x00; x01 = sit_out ? f1 : x01; x02 = sit_out ? f2 : x02; // ... and the state estimate by node 0
            wave_sync();
        }
    }

    // ---- diagonal weights (FULLN build): the reference's controller sets W = diag(Q, R), WN = diag(QN) (mpc_wrapper.cpp: setCosts),
    //      and with literal zeros off the diagonal the Gauss-Newton blocks, the Hessian application of the prediction, the KKT
    //      value and the objective lose a third of their multiply-adds.  Decided per wavefront from the data (every off-diagonal
    //      entry of the W, WN of its problems is +-0), so the results are those of the general path; the rest of the kernel
    //      exists twice, as the two instantiations of one generic lambda.
    bool wdiag = false;
    if constexpr (FULLN && !STAMP) {
        unsigned nz = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float* Wk = lds + oW + gw * 25 * N + 25 * (j * S + s);
#pragma unroll
            for (int i = 0; i < 25; ++i)
                if (i % 6 != 0) nz |= __float_as_uint(Wk[i]);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i % 4 != 0) nz |= __float_as_uint(WN[i]);
        wdiag = __all((nz & 0x7fffffffu) == 0u);
    }
    auto body = [&](auto diagw_tag) {
    constexpr bool DIAGW = decltype(diagw_tag)::value;
    // the two instantiations begin with the same instructions; merged in front of the branch they would stay live through both
    if constexpr (DIAGW) asm volatile("; diagonal weights" ::: "memory");
    else asm volatile("; general weights" ::: "memory");
    // per-stage data of the S stages this lane owns (stage k = j * S + s; slots with k >= N are neutral)
    // Q, q, a, b carry one more element: the node after the lane's block (the next lane's first node, or the terminal
    // node, which also sits in its own slot when it falls inside the block) -- what the prediction's adjoint reads
    // the input map, the input weights and gradients live as PAIRS over the two inputs (even-aligned register pairs): the
    // backward step multiplies them with packed float32 instructions (v_pk_fma_f32: both inputs in one issue slot)
    v2f BP0[S], BP1[S], BP2[S]; // rows of B: (B00, B01), (B10, B11), (B20, -B20)
    v2f RD[S], RP[S];           // (R00, R11), (r0, r1)
    float R01[S];
    float sa[S + 1], sb[S + 1], d0[S], d1[S], d2[S];
    v2f QA[S + 1], QB[S + 1], QC[S + 1], qA[S + 1]; // (Q00, Q01), (Q02, Q12), (Q11, Q22), (q0, q1): the pairs the backward step adds up
    float q2[S + 1];
    float lb0[S], ub0[S], lb1[S], ub1[S];
    int st0[S], st1[S];
    float c00[S], c01[S], c02[S], pf0[S], c10[S], c11[S], c12[S], pe1[S], pf1[S];
    float du0[S], du1[S], dxs[S][3], sbs[S][3];
    // FULLN: every slot of every lane is a stage, the first backward sweep (every lane runs its block, see backward_sweep) writes all
    // policy records and the first forward sweep (every problem starts `changed`) all steps before anything reads them: no initial
    // value is needed, and "no value" costs no instruction where 0 costs ninety moves into registers and AGPRs per wavefront
    auto blank = [] { float v; if constexpr (FULLN) asm("" : "=v"(v)); else v = 0.0f; return v; };
#pragma unroll
    for (int s = 0; s < S; ++s) {
        du0[s] = blank(); du1[s] = blank();
        c00[s] = blank(); c01[s] = blank(); c02[s] = blank(); pf0[s] = blank(); c10[s] = blank(); c11[s] = blank(); c12[s] = blank();
        pe1[s] = blank(); pf1[s] = blank();
#pragma unroll
        for (int c = 0; c < 3; ++c) { dxs[s][c] = blank(); sbs[s][c] = blank(); }
    }
    float dxo[3] = {blank(), blank(), blank()}, sbo[3] = {blank(), blank(), blank()}; // state step / free response leaving the lane's block
    float QN[6], qN[3];                                       // terminal node (meaningful in lane `top`)

    int status = RET_OK, n_iter = 0;
    float kkt = 0.0f;

    // ONCE: one real-time iteration per launch (the control tick): no loop, so the members that only feed the linearisation
    // (od, the raw bounds) are dead after phase A instead of live across the whole body
    const int n_sqp = ONCE ? 1 : p.n_sqp;
    for (int sqp = 0; sqp < n_sqp; ++sqp) {
    PHASE("linearise+cost");
        // ---- phase A: linearise, Gauss-Newton cost, bounds on the step, working-set guess from the dual
        int infeasible = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s;
            const bool vs = k < N;
            const int kc = min(k, N - 1);
            // node k + 1: the next slot, the first slot of the next lane, or the terminal node
            float xn[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float nb = (s + 1 < S) ? x[(s + 1 < S) ? s + 1 : s][c] : lane_next<L>(x[0][c]);
                // slots past the horizon take 0, not the shifted value: for the last lane of a group that is the NEXT problem's
                // node 0, and 0 * NaN of a broken neighbour must not enter this problem's defects
                xn[c] = (k + 1 == N) ? xN[c] : ((k + 1 < N) ? nb : 0.0f);
            }
            StageLin lin;
            ddr_linearize(K, x[s][0], x[s][1], x[s][2], u[s][0], u[s][1], od[s][0], od[s][1], od[s][2], lin);
            const float* yk = lds + oY + ge * 5 * N + 5 * kc;
            const float* Wk = lds + oW + gw * 25 * N + 25 * kc;
            float w[25];
#pragma unroll
            for (int i = 0; i < 25; ++i) w[i] = (DIAGW && i % 6 != 0) ? 0.0f : Wk[i]; // diagonal path: literal zeros fold everything that follows
            const float e0 = x[s][0] - yk[0], e1 = x[s][1] - yk[1], e2 = x[s][2] - yk[2], e3 = u[s][0] - yk[3],
                        e4 = u[s][1] - yk[4];
            const float m = vs ? 1.0f : 0.0f;
            qA[s].x = m * (w[0] * e0 + w[1] * e1 + w[2] * e2 + w[3] * e3 + w[4] * e4);
            qA[s].y = m * (w[5] * e0 + w[6] * e1 + w[7] * e2 + w[8] * e3 + w[9] * e4);
            q2[s] = m * (w[10] * e0 + w[11] * e1 + w[12] * e2 + w[13] * e3 + w[14] * e4);
            RP[s].x = m * (w[15] * e0 + w[16] * e1 + w[17] * e2 + w[18] * e3 + w[19] * e4);
            RP[s].y = m * (w[20] * e0 + w[21] * e1 + w[22] * e2 + w[23] * e3 + w[24] * e4);
            QA[s].x = m * w[0]; QA[s].y = m * w[1]; QB[s].x = m * w[2]; QC[s].x = m * w[6]; QB[s].y = m * w[7]; QC[s].y = m * w[12];
            RD[s].x = vs ? w[18] : 1.0f; R01[s] = m * w[19]; RD[s].y = vs ? w[24] : 1.0f;
            BP0[s].x = m * lin.B00; BP0[s].y = m * lin.B01; BP1[s].x = m * lin.B10; BP1[s].y = m * lin.B11; BP2[s].x = m * lin.B20; BP2[s].y = -(m * lin.B20);
            sa[s] = m * lin.a; sb[s] = m * lin.b;
            d0[s] = m * (lin.phi0 - xn[0]); d1[s] = m * (lin.phi1 - xn[1]); d2[s] = m * (lin.phi2 - xn[2]);
            const float l0 = m * (lbv[s][0] - u[s][0]), l1 = m * (lbv[s][1] - u[s][1]);
            const float h0 = m * (ubv[s][0] - u[s][0]), h1 = m * (ubv[s][1] - u[s][1]);
            lb0[s] = l0; ub0[s] = h0; lb1[s] = l1; ub1[s] = h1;
            infeasible |= (vs && ((l0 > h0 + 1e-6f) || (l1 > h1 + 1e-6f))) ? 1 : 0;
            st0[s] = vs ? status_from_dual(mu0[s], l0, h0) : ST_LOWER;
            st1[s] = vs ? status_from_dual(mu1[s], l1, h1) : ST_LOWER;
        }
        {
            const float e0 = xN[0] - yN[0], e1 = xN[1] - yN[1], e2 = xN[2] - yN[2];
            float wn[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) wn[i] = (DIAGW && i % 4 != 0) ? 0.0f : WN[i];
            QN[0] = wn[0]; QN[1] = wn[1]; QN[2] = wn[2]; QN[3] = wn[4]; QN[4] = wn[5]; QN[5] = wn[8];
            qN[0] = wn[0] * e0 + wn[1] * e1 + wn[2] * e2;
            qN[1] = wn[3] * e0 + wn[4] * e1 + wn[5] * e2;
            qN[2] = wn[6] * e0 + wn[7] * e1 + wn[8] * e2;
        }
        { // element [S]: first node of the next lane; the terminal node goes where node N falls
            QA[S].x = lane_next<L>(QA[0].x); QA[S].y = lane_next<L>(QA[0].y); QB[S].x = lane_next<L>(QB[0].x); QC[S].x = lane_next<L>(QC[0].x);
            QB[S].y = lane_next<L>(QB[0].y); QC[S].y = lane_next<L>(QC[0].y); qA[S].x = lane_next<L>(qA[0].x); qA[S].y = lane_next<L>(qA[0].y);
            q2[S] = lane_next<L>(q2[0]); sa[S] = lane_next<L>(sa[0]); sb[S] = lane_next<L>(sb[0]);
            if (j == L - 1) { QA[S].x = QA[S].y = QB[S].x = QC[S].x = QB[S].y = QC[S].y = qA[S].x = qA[S].y = q2[S] = sa[S] = sb[S] = 0.0f; }
#pragma unroll
            for (int s = 0; s <= S; ++s) {
                if (j * S + s == N) {
                    QA[s].x = QN[0]; QA[s].y = QN[1]; QB[s].x = QN[2]; QC[s].x = QN[3]; QB[s].y = QN[4]; QC[s].y = QN[5];
                    qA[s].x = qN[0]; qA[s].y = qN[1]; q2[s] = qN[2]; sa[s] = 0.0f; sb[s] = 0.0f;
                }
            }
        }
        infeasible = gany<L>(infeasible != 0, gbase) ? 1 : 0;
        if (STAMP && sqp == 0) t1 = __builtin_amdgcn_s_memtime();

        const float Dx0 = x00 - gfirst<L>(x[0][0], lane), Dx1 = x01 - gfirst<L>(x[0][1], lane),
                    Dx2 = x02 - gfirst<L>(x[0][2], lane);

    PHASE("prediction");
        // ---- working-set prediction for cold starts (see nmpc_kernels.hip): projected Barzilai-Borwein steps on
        //      the condensed QP, its Hessian applied stage-wise by prefix / suffix sums (serial inside the lane's
        //      block, DPP scan across the group).  Only a guess: the sweeps below iterate to a fixed point.
        if (p.pg_steps > 0) {
            long long tp0 = 0;
            if (STAMP) tp0 = __builtin_amdgcn_s_memtime();
            int nonfree = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) nonfree |= (j * S + s < N) ? (st0[s] | st1[s]) : 0;
            const bool cold = !gany<L>(nonfree != 0, gbase);
            // gradient H du + g of the condensed QP at du
            float g0[S], g1[S];
            auto apply = [&](const float (&v0)[S], const float (&v1)[S], float (&g0)[S], float (&g1)[S]) {
                float X0[S], X1[S], X2[S], in2[S];
                float acc = 0.0f;
#pragma unroll
                for (int s = 0; s < S; ++s) { acc += BP2[s].x * (v0[s] - v1[s]) + d2[s]; X2[s] = acc; }
                const float ex2 = gprefix<L>(acc, j) - acc + Dx2; // psi entering the block
#pragma unroll
                for (int s = 0; s < S; ++s) { in2[s] = (s == 0) ? ex2 : X2[(s > 0) ? s - 1 : 0] + ex2; }
#pragma unroll
                for (int s = 0; s < S; ++s) X2[s] += ex2;
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    a0 += sa[s] * in2[s] + BP0[s].x * v0[s] + BP0[s].y * v1[s] + d0[s]; X0[s] = a0;
                    a1 += sb[s] * in2[s] + BP1[s].x * v0[s] + BP1[s].y * v1[s] + d1[s]; X1[s] = a1;
                }
                const float ex0 = gprefix<L>(a0, j) - a0 + Dx0, ex1 = gprefix<L>(a1, j) - a1 + Dx1;
                float y0[S], y1[S], y2[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float X0s = X0[s] + ex0, X1s = X1[s] + ex1;
                    y0[s] = QA[s + 1].x * X0s + QA[s + 1].y * X1s + QB[s + 1].x * X2[s] + qA[s + 1].x;
                    y1[s] = QA[s + 1].y * X0s + QC[s + 1].x * X1s + QB[s + 1].y * X2[s] + qA[s + 1].y;
                    y2[s] = QB[s + 1].x * X0s + QB[s + 1].y * X1s + QC[s + 1].y * X2[s] + q2[s + 1];
                }
                // adjoint at node k + 1: suffix sums
                float Lx[S], Ly[S], Lp[S];
                float b0 = 0.0f, b1 = 0.0f;
#pragma unroll
                for (int s = S - 1; s >= 0; --s) { b0 += y0[s]; Lx[s] = b0; b1 += y1[s]; Ly[s] = b1; }
                const float pr0 = gprefix<L>(b0, j), pr1 = gprefix<L>(b1, j);
                const float es0 = glast<L>(pr0, lane) - pr0, es1 = glast<L>(pr1, lane) - pr1; // sum over the lanes above
#pragma unroll
                for (int s = 0; s < S; ++s) { Lx[s] += es0; Ly[s] += es1; }
                // adjoint at node k + 2: the next slot's, the next lane's first, zero past the end of the group
                const float nx_edge = (j == L - 1) ? 0.0f : lane_next<L>(Lx[0]);
                const float ny_edge = (j == L - 1) ? 0.0f : lane_next<L>(Ly[0]);
                float b2 = 0.0f;
#pragma unroll
                for (int s = S - 1; s >= 0; --s) {
                    const float nxs = (s + 1 < S) ? Lx[(s + 1 < S) ? s + 1 : s] : nx_edge;
                    const float nys = (s + 1 < S) ? Ly[(s + 1 < S) ? s + 1 : s] : ny_edge;
                    b2 += y2[s] + sa[s + 1] * nxs + sb[s + 1] * nys;
                    Lp[s] = b2;
                }
                const float pr2 = gprefix<L>(b2, j);
                const float es2 = glast<L>(pr2, lane) - pr2;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float Lps = Lp[s] + es2;
                    g0[s] = RD[s].x * v0[s] + R01[s] * v1[s] + RP[s].x + BP0[s].x * Lx[s] + BP1[s].x * Ly[s] + BP2[s].x * Lps;
                    g1[s] = R01[s] * v0[s] + RD[s].y * v1[s] + RP[s].y + BP0[s].y * Lx[s] + BP1[s].y * Ly[s] - BP2[s].x * Lps;
                }
            };
            float is0[S], is1[S], w0[S], w1[S];
            int hits = 0, badw = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) { w0[s] = 0.0f; w1[s] = 0.0f; }
            apply(w0, w1, g0, g1);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bool in = j * S + s < N;
                is0[s] = in ? rcp_f(fmaxf(RD[s].x, 1e-20f)) : 0.0f;
                is1[s] = in ? rcp_f(fmaxf(RD[s].y, 1e-20f)) : 0.0f;
                const float j0 = -is0[s] * g0[s], j1 = -is1[s] * g1[s];
                hits |= (j0 < lb0[s]) | (j0 > ub0[s]) | (j1 < lb1[s]) | (j1 > ub1[s]);
                badw |= in ? ((!(RD[s].x > 0.0f)) | (!(RD[s].y > 0.0f))) : 0;
                w0[s] = clampf(j0, lb0[s], ub0[s]); w1[s] = clampf(j1, lb1[s], ub1[s]);
            }
            const bool run = cold && gany<L>(hits != 0, gbase) && !gany<L>(badw != 0, gbase);
            if (__any(run)) {
                // Two buffers take turns as (iterate, gradient) of the current and of the previous step: a step reads both and writes the
                // next iterate over the previous one, so nothing is copied from step to step.
                // FULLN (the grid build, 3 .. 4 steps): plain steps, no bookkeeping.  Elsewhere a problem whose predicted set has
                // not moved for two steps stops updating and the loop ends when all have (the fewest sweeps for a launch on its own).
                float v0[S], v1[S], h0[S], h1[S]; // second buffer: starts as the previous point (0, gradient at 0)
                int bits[S];
                auto at_bounds = [&](float a0, float a1, int s) {
                    return (a0 <= lb0[s] ? 1 : 0) | (a0 >= ub0[s] ? 2 : 0) | (a1 <= lb1[s] ? 4 : 0) | (a1 >= ub1[s] ? 8 : 0);
                };
#pragma unroll
                for (int s = 0; s < S; ++s) { v0[s] = 0.0f; v1[s] = 0.0f; h0[s] = g0[s]; h1[s] = g1[s]; bits[s] = FULLN ? 0 : at_bounds(w0[s], w1[s], s); }
                float alpha = 1.0f;
                const int max_steps = FULLN ? p.pg_steps + 1 : p.pg_steps + p.pg_steps / 2; // FULLN: exactly pg_steps steps
                int still = 0;
                bool frozen = !run;
                // one step: gradient at the current iterate (c*) into cg*, Barzilai-Borwein length from the differences to the previous point
                // (q*, qg*), next iterate over q*; returns true when every problem of the wavefront has stopped
                auto bb_step = [&](int t, const float (&c0)[S], const float (&c1)[S], float (&cg0)[S], float (&cg1)[S], float (&q0)[S], float (&q1)[S],
                                   const float (&qg0)[S], const float (&qg1)[S]) -> bool {
                    apply(c0, c1, cg0, cg1);
                    float num = 0.0f, den = 0.0f;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float e0 = c0[s] - q0[s], e1 = c1[s] - q1[s];
                        num += RD[s].x * e0 * e0 + RD[s].y * e1 * e1;
                        den += e0 * (cg0[s] - qg0[s]) + e1 * (cg1[s] - qg1[s]);
                    }
                    num = gtotal<L>(num, j, lane);
                    den = gtotal<L>(den, j, lane);
                    alpha = (den > 1e-30f) ? fminf(fmaxf(num * __builtin_amdgcn_rcpf(den), 1e-3f), 1.0f) : alpha;
                    int moved = 0;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float n0 = __builtin_amdgcn_fmed3f(c0[s] - alpha * is0[s] * cg0[s], lb0[s], ub0[s]);
                        const float n1 = __builtin_amdgcn_fmed3f(c1[s] - alpha * is1[s] * cg1[s], lb1[s], ub1[s]);
                        if constexpr (FULLN) { // a problem that does not take the prediction runs along: its result is not used
                            q0[s] = n0; q1[s] = n1;
                        } else {
                            q0[s] = frozen ? c0[s] : n0;
                            q1[s] = frozen ? c1[s] : n1;
                            const int nb = at_bounds(q0[s], q1[s], s);
                            moved |= (j * S + s < N && nb != bits[s]) ? 1 : 0;
                            bits[s] = nb;
                        }
                    }
                    if constexpr (FULLN) return false;
                    still = gany<L>(moved != 0, gbase) ? 0 : still + 1;
                    frozen = frozen || (still >= 2 && t + 1 >= p.pg_steps);
                    return __all(frozen);
                };
                auto take_set = [&](const float (&a0)[S], const float (&a1)[S]) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const bool set = run && (j * S + s < N);
                        const int n0 = (ub0[s] - lb0[s] > BOUNDTOL) ? ((a0[s] <= lb0[s]) ? ST_LOWER : ((a0[s] >= ub0[s]) ? ST_UPPER : ST_FREE)) : ST_LOWER;
                        const int n1 = (ub1[s] - lb1[s] > BOUNDTOL) ? ((a1[s] <= lb1[s]) ? ST_LOWER : ((a1[s] >= ub1[s]) ? ST_UPPER : ST_FREE)) : ST_LOWER;
                        st0[s] = set ? n0 : st0[s];
                        st1[s] = set ? n1 : st1[s];
                    }
                };
                bool in_w = true; // the latest iterate is in (w0, w1), else in (v0, v1)
#pragma unroll 1
                for (int t = 1; t < max_steps; t += 2) {
                    const bool done = bb_step(t, w0, w1, g0, g1, v0, v1, h0, h1);
                    in_w = false;
                    if (done || t + 1 >= max_steps) break;
                    const bool done2 = bb_step(t + 1, v0, v1, h0, h1, w0, w1, g0, g1);
                    in_w = true;
                    if (done2) break;
                }
                if (in_w) take_set(w0, w1);
                else take_set(v0, v1);
            }
            if (STAMP) t_pg = __builtin_amdgcn_s_memtime() - tp0;
        }

    PHASE("sweep_setup");
        // ---- phase B: working-set iterations; the sweeps go lane by lane through the group
        float V[9];    // cost-to-go travelling down the lanes: P00 P01 P02 P11 P12 P22 p0 p1 p2
        float Vin[9];  // cost-to-go at the upper end of this lane's block (kept for restarts)
#pragma unroll
        for (int i = 0; i < 6; ++i) { V[i] = QN[i]; Vin[i] = QN[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { V[6 + i] = qN[i]; Vin[6 + i] = qN[i]; }
        int pd_fail = 0;
        bool changed = true;
        int khi = N - 1;
        int it = 0;
        n_iter = 0;

        // state step entering this lane's block: compose the closed-loop maps dx+ = (A + B G) dx + (B h + d) of the
        // lane's stages (G, h from the policy records and the working set; slots past the horizon are identities
        // by construction), scan over the lanes, apply to x0 - x[0]
        auto block_entry = [&](float& o0, float& o1, float& o2) {
            float m00 = 1.f, m01 = 0.f, m02 = 0.f, m10 = 0.f, m11 = 1.f, m12 = 0.f, m20 = 0.f, m21 = 0.f, m22 = 1.f;
            float k0 = 0.f, k1 = 0.f, k2 = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bool f0 = (st0[s] == ST_FREE), f1 = (st1[s] == ST_FREE);
                const float b0 = (st0[s] == ST_UPPER) ? ub0[s] : lb0[s], b1 = (st1[s] == ST_UPPER) ? ub1[s] : lb1[s];
                const float g00 = f0 ? c00[s] : 0.0f, g01 = f0 ? c01[s] : 0.0f, g02 = f0 ? c02[s] : 0.0f;
                const float h0 = f0 ? pf0[s] : b0;
                const float g10 = f1 ? c10[s] + pe1[s] * g00 : 0.0f, g11 = f1 ? c11[s] + pe1[s] * g01 : 0.0f,
                            g12 = f1 ? c12[s] + pe1[s] * g02 : 0.0f;
                const float h1 = f1 ? pe1[s] * h0 + pf1[s] : b1;
                const float gd0 = g00 - g10, gd1 = g01 - g11, gd2 = g02 - g12;
                const float a00 = 1.0f + BP0[s].x * g00 + BP0[s].y * g10, a01 = BP0[s].x * g01 + BP0[s].y * g11,
                            a02 = sa[s] + BP0[s].x * g02 + BP0[s].y * g12;
                const float a10 = BP1[s].x * g00 + BP1[s].y * g10, a11 = 1.0f + BP1[s].x * g01 + BP1[s].y * g11,
                            a12 = sb[s] + BP1[s].x * g02 + BP1[s].y * g12;
                const float a20 = BP2[s].x * gd0, a21 = BP2[s].x * gd1, a22 = 1.0f + BP2[s].x * gd2;
                const float cc0 = BP0[s].x * h0 + BP0[s].y * h1 + d0[s], cc1 = BP1[s].x * h0 + BP1[s].y * h1 + d1[s],
                            cc2 = BP2[s].x * (h0 - h1) + d2[s];
                if (s == 0) {
                    m00 = a00; m01 = a01; m02 = a02; m10 = a10; m11 = a11; m12 = a12; m20 = a20; m21 = a21; m22 = a22;
                    k0 = cc0; k1 = cc1; k2 = cc2;
                } else { // stage map after what is composed so far
                    const float n00 = a00 * m00 + a01 * m10 + a02 * m20, n01 = a00 * m01 + a01 * m11 + a02 * m21,
                                n02 = a00 * m02 + a01 * m12 + a02 * m22;
                    const float n10 = a10 * m00 + a11 * m10 + a12 * m20, n11 = a10 * m01 + a11 * m11 + a12 * m21,
                                n12 = a10 * m02 + a11 * m12 + a12 * m22;
                    const float n20 = a20 * m00 + a21 * m10 + a22 * m20, n21 = a20 * m01 + a21 * m11 + a22 * m21,
                                n22 = a20 * m02 + a21 * m12 + a22 * m22;
                    const float l0 = a00 * k0 + a01 * k1 + a02 * k2 + cc0, l1 = a10 * k0 + a11 * k1 + a12 * k2 + cc1,
                                l2 = a20 * k0 + a21 * k1 + a22 * k2 + cc2;
                    m00 = n00; m01 = n01; m02 = n02; m10 = n10; m11 = n11; m12 = n12; m20 = n20; m21 = n21; m22 = n22;
                    k0 = l0; k1 = l1; k2 = l2;
                }
            }
            // inclusive scan over the lanes: lane j <- f_j o f_{j-1} o ... o f_0
            auto level = [&](auto tag, auto rows, int dist) {
                constexpr int CTRL = decltype(tag)::value;
                constexpr int ROWS = decltype(rows)::value;
                const bool has = (L >= 16) || (j >= dist);
                auto fetch = [&](float v, float ident) { const float r = dppr<CTRL, ROWS>(ident, v); return has ? r : ident; };
                const float g00 = fetch(m00, 1.f), g01 = fetch(m01, 0.f), g02 = fetch(m02, 0.f), g10 = fetch(m10, 0.f),
                            g11 = fetch(m11, 1.f), g12 = fetch(m12, 0.f), g20 = fetch(m20, 0.f), g21 = fetch(m21, 0.f),
                            g22 = fetch(m22, 1.f), gc0 = fetch(k0, 0.f), gc1 = fetch(k1, 0.f), gc2 = fetch(k2, 0.f);
                const float n00 = m00 * g00 + m01 * g10 + m02 * g20, n01 = m00 * g01 + m01 * g11 + m02 * g21,
                            n02 = m00 * g02 + m01 * g12 + m02 * g22;
                const float n10 = m10 * g00 + m11 * g10 + m12 * g20, n11 = m10 * g01 + m11 * g11 + m12 * g21,
                            n12 = m10 * g02 + m11 * g12 + m12 * g22;
                const float n20 = m20 * g00 + m21 * g10 + m22 * g20, n21 = m20 * g01 + m21 * g11 + m22 * g21,
                            n22 = m20 * g02 + m21 * g12 + m22 * g22;
                const float l0 = m00 * gc0 + m01 * gc1 + m02 * gc2 + k0, l1 = m10 * gc0 + m11 * gc1 + m12 * gc2 + k1,
                            l2 = m20 * gc0 + m21 * gc1 + m22 * gc2 + k2;
                m00 = n00; m01 = n01; m02 = n02; m10 = n10; m11 = n11; m12 = n12; m20 = n20; m21 = n21; m22 = n22;
                k0 = l0; k1 = l1; k2 = l2;
            };
            using all_rows = std::integral_constant<int, 0xF>;
            level(std::integral_constant<int, 0x111>{}, all_rows{}, 1);
            level(std::integral_constant<int, 0x112>{}, all_rows{}, 2);
            if constexpr (L >= 8) level(std::integral_constant<int, 0x114>{}, all_rows{}, 4);
            if constexpr (L >= 16) level(std::integral_constant<int, 0x118>{}, all_rows{}, 8);
            if constexpr (L >= 32) level(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{}, 16); // row_bcast:15 into the group's second row
            // state step leaving this lane's block; the one entering it is the previous lane's
            const float l0 = m00 * Dx0 + m01 * Dx1 + m02 * Dx2 + k0, l1 = m10 * Dx0 + m11 * Dx1 + m12 * Dx2 + k1,
                        l2 = m20 * Dx0 + m21 * Dx1 + m22 * Dx2 + k2;
            const float p0 = lane_prev<L>(l0), p1 = lane_prev<L>(l1), p2 = lane_prev<L>(l2);
            o0 = (j == 0) ? Dx0 : p0; o1 = (j == 0) ? Dx1 : p1; o2 = (j == 0) ? Dx2 : p2;
        };
        // One lane's block of the backward recursion.  EVERY lane of the wavefront runs it on its own slots with whatever
        // cost-to-go it holds -- only lane t of a sweeping group holds the real one, and only it keeps the policy records
        // (`mine`, by selects).  No heavy code runs under a partial EXEC mask: values that are live across such a region in
        // the lanes that sit it out are not safe from the register allocator's spill / reload pairs there (seen in
        // ltv_mpc.hip).  `t` is wavefront-uniform, so the slots past the horizon are skipped with a uniform branch.
        auto backward_block = [&](int t, bool mine, auto full_tag) -> int {
            constexpr bool FULL = decltype(full_tag)::value; // every slot of lane t is a stage: one basic block
            ValuePk val; // V = P00 P01 P02 P11 P12 P22 p0 p1 p2
            val.Pa = mk2(V[0], V[1]); val.Pb = mk2(V[2], V[4]); val.Pc = mk2(V[3], V[5]); val.pa = mk2(V[6], V[7]); val.p2 = V[8];
            int ok = 1;
#pragma unroll
            for (int s = S - 1; s >= 0; --s) {
                if (FULL || t * S + s < N) { // wavefront-uniform: the stage of lane t exists
                    StagePk q;
                    q.a = sa[s]; q.b = sb[s]; q.B0 = BP0[s]; q.B1 = BP1[s]; q.B2 = BP2[s];
                    q.d0 = d0[s]; q.d1 = d1[s]; q.d2 = d2[s];
                    q.QA = QA[s]; q.QB = QB[s]; q.QC = QC[s]; q.qa = qA[s]; q.q2 = q2[s];
                    q.Rd = RD[s]; q.R01 = R01[s]; q.r = RP[s];
                    q.st0 = st0[s]; q.st1 = st1[s];
                    q.v0 = (st0[s] == ST_UPPER) ? ub0[s] : lb0[s];
                    q.v1 = (st1[s] == ST_UPPER) ? ub1[s] : lb1[s];
                    Policy pol;
                    ok &= riccati_step_pk(q, val, pol) ? 1 : 0;
                    c00[s] = mine ? pol.c00 : c00[s]; c01[s] = mine ? pol.c01 : c01[s]; c02[s] = mine ? pol.c02 : c02[s];
                    pf0[s] = mine ? pol.f0 : pf0[s];
                    c10[s] = mine ? pol.c10 : c10[s]; c11[s] = mine ? pol.c11 : c11[s]; c12[s] = mine ? pol.c12 : c12[s];
                    pe1[s] = mine ? pol.e1 : pe1[s]; pf1[s] = mine ? pol.f1 : pf1[s];
                }
            }
            V[0] = val.Pa.x; V[1] = val.Pa.y; V[2] = val.Pb.x; V[3] = val.Pc.x; V[4] = val.Pb.y; V[5] = val.Pc.y;
            V[6] = val.pa.x; V[7] = val.pa.y; V[8] = val.p2;
            return ok;
        };
        // backward sweep of the groups flagged `act`, each from the lane that owns its highest stale stage `from`
        auto backward_sweep = [&](bool act, int from) {
            __builtin_amdgcn_s_setprio(3);
            for (int t = top; t >= 0; --t) {
                const bool mine = act && (j == t) && (t <= from);
                if (__any(act && t <= from)) {
                    // lane t of a group that starts here takes the cost-to-go it kept; below, the one handed down is kept
                    const bool start = mine && (t == from);
                    if constexpr (MASKED) {
                        // (16, 2) and (32, 1) only: the block of lane t under an EXEC mask (a tenth fewer instructions per sweep:
                        // no selects).  This instantiation keeps its whole state in VGPRs, and tests/test_masked_regions.py
                        // checks on every build that no spill / reload / AGPR traffic sits inside the masked region.
                        if (mine) {
#pragma unroll
                            for (int i = 0; i < 9; ++i) { V[i] = start ? Vin[i] : V[i]; Vin[i] = V[i]; }
                            const int ok = ((t + 1) * S <= N) ? backward_block(t, true, std::true_type{}) : backward_block(t, true, std::false_type{});
                            pd_fail |= ok ? 0 : 1;
                        }
                    } else {
                        // EVERY lane runs its block from the cost-to-go that enters IT (Vin) and keeps what comes out: a lane whose turn
                        // has been (j > t, or its group sits this sweep out, or starts below) reproduces its own records bit for bit
                        // -- same instructions, same inputs -- and a lane whose turn is still to come writes records that its turn
                        // replaces.  So the nine policy values of a stage need no select; only the entering cost-to-go of the lane
                        // whose turn it is does (it takes what was handed down, unless its group starts the sweep here).
                        // (Only where the horizon fills the mapping.  Elsewhere the lane that owns stage N - 1 skips its slots past the
                        // horizon, one of which carries the terminal weights for the prediction: run again from Vin over all S slots
                        // it would add them a second time.  There the records are committed by selects, on lane t's turn only.)
                        if constexpr (FULLN) {
                            const bool take = mine && !start;
#pragma unroll
                            for (int i = 0; i < 9; ++i) { Vin[i] = take ? V[i] : Vin[i]; V[i] = Vin[i]; }
                            const int ok = backward_block(t, true, std::true_type{});
                            pd_fail |= (mine && !ok) ? 1 : 0;
                        } else {
                            if (__any(start)) {
#pragma unroll
                                for (int i = 0; i < 9; ++i) V[i] = start ? Vin[i] : V[i];
                            }
#pragma unroll
                            for (int i = 0; i < 9; ++i) Vin[i] = mine ? V[i] : Vin[i];
                            const int ok = ((t + 1) * S <= N) ? backward_block(t, mine, std::true_type{}) : backward_block(t, mine, std::false_type{});
                            pd_fail |= (mine && !ok) ? 1 : 0;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 9; ++i) V[i] = lane_next<L>(V[i]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };

        for (;;) {
            long long tb0 = 0;
            if (STAMP) tb0 = __builtin_amdgcn_s_memtime();
    PHASE("backward");
            backward_sweep(changed, khi / S);
            long long tf0 = 0;
            if (STAMP) { tf0 = __builtin_amdgcn_s_memtime(); t_b += tf0 - tb0; }

    PHASE("forward");
            // ---- forward sweep: every lane condenses its block into one affine map under the current working set,
            //      a DPP scan over the lanes gives the state step entering each block, then the lanes walk their own
            //      stages in parallel (nmpc_core.h: forward_step)
            const bool active = changed;
            int new_khi = -1;
            if (__any(active)) {
                float e0, e1, e2;
                block_entry(e0, e1, e2);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    Policy pol;
                    pol.c00 = c00[s]; pol.c01 = c01[s]; pol.c02 = c02[s]; pol.f0 = pf0[s];
                    pol.c10 = c10[s]; pol.c11 = c11[s]; pol.c12 = c12[s]; pol.e1 = pe1[s]; pol.f1 = pf1[s];
                    StageStep o;
                    forward_step(pol, st0[s], st1[s], e0, e1, e2, lb0[s], ub0[s], lb1[s], ub1[s], o);
                    const bool moved = (o.nst0 != st0[s]) || (o.nst1 != st1[s]);
                    if (moved) new_khi = j * S + s; // ascending: the last one is the highest
                    if (active) {
                        st0[s] = o.nst0; st1[s] = o.nst1;
                        du0[s] = o.du0; du1[s] = o.du1; mu0[s] = o.mu0; mu1[s] = o.mu1;
                        dxs[s][0] = e0; dxs[s][1] = e1; dxs[s][2] = e2;
                    }
                    const float n0 = e0 + sa[s] * e2 + BP0[s].x * o.du0 + BP0[s].y * o.du1 + d0[s];
                    const float n1 = e1 + sb[s] * e2 + BP1[s].x * o.du0 + BP1[s].y * o.du1 + d1[s];
                    const float n2 = e2 + BP2[s].x * (o.du0 - o.du1) + d2[s];
                    e0 = n0; e1 = n1; e2 = n2;
                }
                if (active) { dxo[0] = e0; dxo[1] = e1; dxo[2] = e2; }
                new_khi = gmax<L>(new_khi, j, lane);
            }
            ++it;
            if (active) {
                changed = (new_khi >= 0);
                khi = changed ? new_khi : 0;
                if (changed) n_iter = it;
            }
            if (STAMP) t_f += __builtin_amdgcn_s_memtime() - tf0;
            if (!__any(changed && it < min(p.max_as_iter, AS_SWITCH))) break;
        }
        n_iter = (n_iter == 0) ? 1 : (changed ? n_iter : n_iter + 1); // + the confirming sweep

    PHASE("safeguard");
        // ---- safeguard (nmpc_core.h: AS_SWITCH): primal active-set iteration for the rare problem whose
        //      primal-dual update has not settled; one change of the working set per sweep.  Cold path.
        {
            const bool rescue = changed && it >= AS_SWITCH && it < p.max_as_iter;
            if (__builtin_expect(__any(rescue) ? 1 : 0, 0)) {
                float cur0[S], cur1[S];
                int todo = rescue ? 1 : 0;
#pragma unroll
                for (int s = 0; s < S; ++s) { // start: clip the last solution into the box, fix what sits on a bound
                    const float a0 = (lb0[s] <= ub0[s]) ? clampf(du0[s], lb0[s], ub0[s]) : du0[s];
                    const float a1 = (lb1[s] <= ub1[s]) ? clampf(du1[s], lb1[s], ub1[s]) : du1[s];
                    cur0[s] = a0; cur1[s] = a1;
                    if (rescue && j * S + s < N) {
                        st0[s] = (ub0[s] - lb0[s] > BOUNDTOL) ? asm_status_of(a0, lb0[s], ub0[s]) : ST_LOWER;
                        st1[s] = (ub1[s] - lb1[s] > BOUNDTOL) ? asm_status_of(a1, lb1[s], ub1[s]) : ST_LOWER;
                    }
                }
                for (;;) {
                    backward_sweep(todo == 1, top);
                    // forward sweep with the ratio test (free controls) and the multiplier test (fixed ones)
                    float alpha = AS_NONE, viol = 0.0f;
                    int akey = 0x7fffffff, vkey = 0x7fffffff; // (2 * stage + control) * 4 + bound hit
                    {
                        float e0, e1, e2;
                        block_entry(e0, e1, e2);
#pragma unroll
                        for (int s = 0; s < S; ++s) { // slots past the horizon are equality-bounded: they never block or release
                            const int k = j * S + s;
                            Policy pol;
                            pol.c00 = c00[s]; pol.c01 = c01[s]; pol.c02 = c02[s]; pol.f0 = pf0[s];
                            pol.c10 = c10[s]; pol.c11 = c11[s]; pol.c12 = c12[s]; pol.e1 = pe1[s]; pol.f1 = pf1[s];
                            StageStep o;
                            forward_step(pol, st0[s], st1[s], e0, e1, e2, lb0[s], ub0[s], lb1[s], ub1[s], o);
                            int h;
                            const float ra = asm_ratio(st0[s], cur0[s], o.du0, lb0[s], ub0[s], h);
                            if (ra < alpha) { alpha = ra; akey = (2 * k) * 4 + h; }
                            const float rb = asm_ratio(st1[s], cur1[s], o.du1, lb1[s], ub1[s], h);
                            if (rb < alpha) { alpha = rb; akey = (2 * k + 1) * 4 + h; }
                            const float va = asm_violation(st0[s], o.mu0, lb0[s], ub0[s]),
                                        vb = asm_violation(st1[s], o.mu1, lb1[s], ub1[s]);
                            if (va > viol) { viol = va; vkey = (2 * k) * 4; }
                            if (vb > viol) { viol = vb; vkey = (2 * k + 1) * 4; }
                            if (todo == 1) {
                                du0[s] = o.du0; du1[s] = o.du1; mu0[s] = o.mu0; mu1[s] = o.mu1;
                                dxs[s][0] = e0; dxs[s][1] = e1; dxs[s][2] = e2;
                            }
                            const float n0 = e0 + sa[s] * e2 + BP0[s].x * o.du0 + BP0[s].y * o.du1 + d0[s];
                            const float n1 = e1 + sb[s] * e2 + BP1[s].x * o.du0 + BP1[s].y * o.du1 + d1[s];
                            const float n2 = e2 + BP2[s].x * (o.du0 - o.du1) + d2[s];
                            e0 = n0; e1 = n1; e2 = n2;
                        }
                        if (todo == 1) { dxo[0] = e0; dxo[1] = e1; dxo[2] = e2; }
                    }
                    if (todo == 1) ++it;
                    // the group's first blocking bound / worst multiplier (ties: lowest stage): lexicographic minima
                    gmin_pair<L>(alpha, akey, j, lane);
                    float nviol = -viol;
                    gmin_pair<L>(nviol, vkey, j, lane);
                    viol = -nviol;
                    if (todo == 1) {
                        const bool blocked = akey != 0x7fffffff;
                        const bool release = !blocked && vkey != 0x7fffffff;
                        const float al = blocked ? fmaxf(alpha, 0.0f) : 1.0f;
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const int k = j * S + s;
                            if (k < N) {
                                if (st0[s] == ST_FREE) cur0[s] += al * (du0[s] - cur0[s]);
                                if (st1[s] == ST_FREE) cur1[s] += al * (du1[s] - cur1[s]);
                                if (blocked && (akey >> 3) == k) {
                                    const int hit = akey & 3;
                                    if ((akey >> 2) & 1) { cur1[s] = (hit == ST_UPPER) ? ub1[s] : lb1[s]; st1[s] = hit; }
                                    else { cur0[s] = (hit == ST_UPPER) ? ub0[s] : lb0[s]; st0[s] = hit; }
                                }
                                if (release && (vkey >> 3) == k) {
                                    if ((vkey >> 2) & 1) st1[s] = ST_FREE; else st0[s] = ST_FREE;
                                }
                            }
                        }
                        if (!blocked && !release) todo = 0;  // optimal
                        else if (it >= p.max_as_iter) todo = 2; // cap reached
                    }
                    if (!__any(todo == 1)) break;
                }
                if (rescue) {
                    changed = (todo == 2);
                    n_iter = it;
                }
            }
        }
        status = infeasible ? RET_INIT_FAILED_INFEASIBILITY
                            : (pd_fail ? RET_INIT_FAILED_CHOLESKY : (changed ? RET_MAX_NWSR_REACHED : RET_OK));
        if (STAMP && sqp == 0) t4 = __builtin_amdgcn_s_memtime();

    PHASE("kkt+expand");
        // ---- phase C: KKT value (acado_getKKT), expand (acado_expand), carry the dual
        if (DIAG) { // free response (du = 0) entering every stage, for acado_getKKT: A is a shear, so prefix sums do it
                    // (here and not before the sweeps: nine registers per lane less across them)
            float loc[S], acc = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) { loc[s] = acc; acc += d2[s]; }
            const float ex2 = gprefix<L>(acc, j) - acc + Dx2;
#pragma unroll
            for (int s = 0; s < S; ++s) sbs[s][2] = loc[s] + ex2;
            sbo[2] = acc + ex2;
            float l0[S], l1[S], a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                l0[s] = a0; a0 += sa[s] * sbs[s][2] + d0[s];
                l1[s] = a1; a1 += sb[s] * sbs[s][2] + d1[s];
            }
            const float ex0 = gprefix<L>(a0, j) - a0 + Dx0, ex1 = gprefix<L>(a1, j) - a1 + Dx1;
#pragma unroll
            for (int s = 0; s < S; ++s) { sbs[s][0] = l0[s] + ex0; sbs[s][1] = l1[s] + ex1; }
            sbo[0] = a0 + ex0; sbo[1] = a1 + ex1;
        }

        float gd = 0.0f, comp = 0.0f;
        if constexpr (ONCE) {
            // single-iteration build: the iterate was only needed for the linearisation; it is read again here (L2 / MALL) so
            // that x, u do not occupy 5 S + 3 registers per lane through the prediction and the sweeps
            typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
            const float* gx = pb.x + (size_t)prob * nx;
            const float* gu = pb.u + (size_t)prob * nu;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = j * S + s;
                const f3u vx = *reinterpret_cast<const f3u*>(gx + 3 * min(k, N));
                const f2u vu = *reinterpret_cast<const f2u*>(gu + 2 * min(k, N - 1));
                x[s][0] = vx.x; x[s][1] = vx.y; x[s][2] = vx.z; u[s][0] = vu.x; u[s][1] = vu.y;
            }
            const f3u vn = *reinterpret_cast<const f3u*>(gx + 3 * N);
            xN[0] = vn.x; xN[1] = vn.y; xN[2] = vn.z;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) { // no predicate: slots past the horizon hold zero steps / multipliers (their x, u are never stored)
            const int k = j * S + s;
            if (DIAG) {
                // (Q_k sbar_k + q_k)' (dx_k - sbar_k), stages 1 .. N - 1 (the slot of the terminal node, if the lane has it,
                // carries QN for the prediction: it is added below)
                const float b0 = sbs[s][0], b1 = sbs[s][1], b2 = sbs[s][2];
                const float e0 = dxs[s][0] - b0, e1 = dxs[s][1] - b1, e2 = dxs[s][2] - b2;
                const float tq = (QA[s].x * b0 + QA[s].y * b1 + QB[s].x * b2 + qA[s].x) * e0 +
                                 (QA[s].y * b0 + QC[s].x * b1 + QB[s].y * b2 + qA[s].y) * e1 +
                                 (QB[s].x * b0 + QB[s].y * b1 + QC[s].y * b2 + q2[s]) * e2;
                gd += (k > 0 && k < N) ? tq : 0.0f;
                gd += RP[s].x * du0[s] + RP[s].y * du1[s];
                comp += (mu0[s] > 1e-12f) ? fabsf(lb0[s] * mu0[s]) : ((mu0[s] < -1e-12f) ? fabsf(ub0[s] * mu0[s]) : 0.0f);
                comp += (mu1[s] > 1e-12f) ? fabsf(lb1[s] * mu1[s]) : ((mu1[s] < -1e-12f) ? fabsf(ub1[s] * mu1[s]) : 0.0f);
            }
            x[s][0] += dxs[s][0]; x[s][1] += dxs[s][1]; x[s][2] += dxs[s][2];
            // a free control may sit up to TOL_PRIMAL outside its box: keep the iterate feasible
            const float e0 = (lb0[s] <= ub0[s]) ? clampf(du0[s], lb0[s], ub0[s]) : du0[s];
            const float e1 = (lb1[s] <= ub1[s]) ? clampf(du1[s], lb1[s], ub1[s]) : du1[s];
            u[s][0] += e0; u[s][1] += e1;
        }
        if (j == top) { // terminal node
            if (DIAG) {
                const float b0 = sbo[0], b1 = sbo[1], b2 = sbo[2];
                const float e0 = dxo[0] - b0, e1 = dxo[1] - b1, e2 = dxo[2] - b2;
                gd += (QN[0] * b0 + QN[1] * b1 + QN[2] * b2 + qN[0]) * e0 + (QN[1] * b0 + QN[3] * b1 + QN[4] * b2 + qN[1]) * e1 +
                      (QN[2] * b0 + QN[4] * b1 + QN[5] * b2 + qN[2]) * e2;
            }
            xN[0] += dxo[0]; xN[1] += dxo[1]; xN[2] += dxo[2];
        }
        if (DIAG) kkt = fabsf(gtotal<L>(gd, j, lane)) + gtotal<L>(comp, j, lane);
    }
    if (STAMP) t5 = __builtin_amdgcn_s_memtime();

    PHASE("objective+store");
    // ---- objective at the returned iterate (acado_getObjective), then the iterate back through LDS
    float obj = 0.0f;
    if (DIAG) {
        float part = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s, kc = min(k, N - 1);
            const float* yk = lds + oY + ge * 5 * N + 5 * kc;
            const float* Wk = lds + oW + gw * 25 * N + 25 * kc;
            const float e[5] = {x[s][0] - yk[0], x[s][1] - yk[1], x[s][2] - yk[2], u[s][0] - yk[3], u[s][1] - yk[4]};
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                if constexpr (DIAGW) {
                    acc += e[c] * (e[c] * Wk[6 * c]);
                } else {
                    const float tt = e[0] * Wk[c] + e[1] * Wk[5 + c] + e[2] * Wk[10 + c] + e[3] * Wk[15 + c] + e[4] * Wk[20 + c];
                    acc += e[c] * tt;
                }
            }
            part += (k < N) ? acc : 0.0f;
        }
        if (j == top) { // the reference uses only the diagonal of WN here (acado_solver.c:1442-1444)
            float w0 = WN[0], w4 = WN[4], w8 = WN[8], y0 = yN[0], y1 = yN[1], y2 = yN[2];
            if constexpr (ONCE) { // read again instead of held since phase A
                const float* gW = pb.WN + ((p.shared & ALORE_NMPC_SHARED_W) ? 0 : (size_t)prob * 9);
                const float* gy = pb.yN + (size_t)prob * 3;
                w0 = gW[0]; w4 = gW[4]; w8 = gW[8]; y0 = gy[0]; y1 = gy[1]; y2 = gy[2];
            }
            const float e0 = xN[0] - y0, e1 = xN[1] - y1, e2 = xN[2] - y2;
            part += e0 * e0 * w0 + e1 * e1 * w4 + e2 * e2 * w8;
        }
        obj = 0.5f * gtotal<L>(part, j, lane);
    }
    // problems the caller masked out (alore_nmpc_set_problem_mask: idle robots of a fleet) go back exactly as they came: their
    // lanes ran along on whatever the members hold (results of a group never reach another group), now they fetch the
    // iterate and the dual again and write nothing else
    const bool skip = valid && p.mask != nullptr && p.mask[prob] == 0;
    // The grid build stores the iterate and the dual straight from the registers of the lane that owns the stage -- 12- and 8-byte
    // pieces of a span the wavefront writes completely, like the loads of phase 0; L2 merges them into whole lines -- instead of
    // transposing them through LDS into 16-byte pieces (76 LDS writes, 26 reads and their waits per wavefront)
    constexpr bool DIRECT = FULLN; // measured on the (16, 2) single-iteration build too (one batch at a time): 22.2 us per launch either way
    if constexpr (DIRECT) {
        if (valid && !skip) {
            typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
            float* gx = pb.x + (size_t)prob * nx;
            float* gu = pb.u + (size_t)prob * nu;
            float* gdl = pb.dual + (size_t)prob * nu;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = j * S + s;
                if (FULLN || k < N) {
                    f3u vx; vx.x = x[s][0]; vx.y = x[s][1]; vx.z = x[s][2];
                    f2u vu; vu.x = u[s][0]; vu.y = u[s][1];
                    f2u vd; vd.x = mu0[s]; vd.y = mu1[s];
                    *reinterpret_cast<f3u*>(gx + 3 * k) = vx;
                    *reinterpret_cast<f2u*>(gu + 2 * k) = vu;
                    *reinterpret_cast<f2u*>(gdl + 2 * k) = vd;
                }
            }
            if (j == top) {
                f3u vn; vn.x = xN[0]; vn.y = xN[1]; vn.z = xN[2];
                *reinterpret_cast<f3u*>(gx + 3 * N) = vn;
            }
        }
    } else {
    if (__any(skip)) {
        typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        const float* gx = pb.x + (size_t)prob * nx;
        const float* gu = pb.u + (size_t)prob * nu;
        const float* gdl = pb.dual + (size_t)prob * nu;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s;
            const f3u vx = *reinterpret_cast<const f3u*>(gx + 3 * min(k, N));
            const f2u vu = *reinterpret_cast<const f2u*>(gu + 2 * min(k, N - 1)), vd = *reinterpret_cast<const f2u*>(gdl + 2 * min(k, N - 1));
            x[s][0] = skip ? vx.x : x[s][0]; x[s][1] = skip ? vx.y : x[s][1]; x[s][2] = skip ? vx.z : x[s][2];
            u[s][0] = skip ? vu.x : u[s][0]; u[s][1] = skip ? vu.y : u[s][1];
            mu0[s] = skip ? vd.x : mu0[s]; mu1[s] = skip ? vd.y : mu1[s];
        }
        const f3u vn = *reinterpret_cast<const f3u*>(gx + 3 * N);
        xN[0] = skip ? vn.x : xN[0]; xN[1] = skip ? vn.y : xN[1]; xN[2] = skip ? vn.z : xN[2];
    }
    wave_sync(); // W / y are dead from here: their area becomes the output staging buffer
    if (valid) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s;
            if (k < N) {
#pragma unroll
                for (int c = 0; c < 3; ++c) lds[oX + g * nx + 3 * k + c] = x[s][c];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    lds[oU + g * nu + 2 * k + c] = u[s][c];
                }
                lds[oDL + g * nu + 2 * k] = mu0[s];
                lds[oDL + g * nu + 2 * k + 1] = mu1[s];
            }
        }
        if (j == top) {
#pragma unroll
            for (int c = 0; c < 3; ++c) lds[oX + g * nx + 3 * N + c] = xN[c];
        }
    }
    wave_sync();
    {
        constexpr int UX = (G * 3 * (NMAX + 1) / 4 + 63) / 64;
        constexpr int UU = (G * 2 * NMAX / 4 + 63) / 64;
        g_store<UX>(pb.x + (size_t)prob0 * nx, lds + oX, np_ * nx, lane);
        g_store<UU>(pb.u + (size_t)prob0 * nu, lds + oU, np_ * nu, lane);
        g_store<UU>(pb.dual + (size_t)prob0 * nu, lds + oDL, np_ * nu, lane);
    }
    }
    if (valid && !skip && j == 0) {
        pb.status[prob] = status;
        pb.n_iter[prob] = n_iter;
        if (DIAG && pb.kkt) pb.kkt[prob] = kkt;
        if (DIAG && pb.obj) pb.obj[prob] = obj;
    }
    if constexpr (FULLN && !PERSIST) {
        if (grp.xcd_on && grp.xcd_end != nullptr) { // the last four workgroups of an XCD leave the time they finished at (host memory: read without a copy)
            const int x = (int)blockIdx.x & 7, left = grp.xcd_share[x] - 1 - ((int)blockIdx.x >> 3); // recomputed: nothing is kept live across the body for it
            if (left < 4 && lane == 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70); // the results are on their way out: the stamp follows them
                grp.xcd_end[x * 4 + left] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
            }
        }
    }
    if constexpr (TRACE) {
        if (grp.trace) {
            int worst = 0; // sweeps of the slowest problem of the wavefront
            while (worst < 64 && __any(n_iter > worst)) ++worst;
            const long long te = (long long)__builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the stores are acknowledged
            if (threadIdx.x == 0) {
                long long* o = grp.trace + (size_t)item * 8;
                o[3] = te;
                o[4] = (long long)__builtin_amdgcn_s_memrealtime();
                o[6] = worst;
                o[7] = DIAGW ? 1 : 0;
            }
        }
    }
    };
    if constexpr (FULLN && !STAMP) {
#ifdef ALORE_ONLY_DIAGW // analysis builds (tools/asm_phases.py): one path in the assembly
        body(std::true_type{});
#else
        if (__builtin_expect(wdiag, 1)) body(std::true_type{}); // block frequencies steer the register allocator: the copies go to the rare path
        else body(std::false_type{});
#endif
    } else {
        body(std::false_type{});
    }
    if constexpr (PERSIST) {
        item = (int)gridDim.x + ticket;
        if (item < grp.blocks_per_batch * grp.count) {
            wave_sync(); // the staging area has been read: the next item's W / y may land
            goto next_item;
        }
        // the last workgroup out puts the counters back for the next launch
        if (lane == 0) {
            if (atomicAdd(grp.counter + 1, 1) == (int)gridDim.x - 1) {
                grp.counter[0] = 0;
                grp.counter[1] = 0;
            }
        }
    }
    if (STAMP && lane == 0 && p.stamps) {
        long long* o = p.stamps + (size_t)blockIdx.x * 8;
        const long long t_end = __builtin_amdgcn_s_memtime();
        o[0] = t1 - t0;    // load + phase A
        o[1] = t_b;        // backward sweeps
        o[2] = t_f;        // forward sweeps
        o[3] = t5 - t4;    // phase C
        o[4] = t_end - t5; // objective + store
        o[5] = t_end - t0; // total
        o[6] = t_pg;       // working-set prediction
    }
}

// (L, S) instantiated: (4, 5) (8, 3) (16, 2) (32, 1) for horizons up to 20 / 24 / 32 / 32, (16, 4) up to 64
bool block_geometry(int B, int N, int forced_L, int lds_limit_bytes, int n_cu, LaunchGeom* g, int B_in_flight)
{
    if (B <= 0 || N <= 0) return false;
    const int cus = n_cu > 0 ? n_cu : 256;
    int L = forced_L;
    if (L == 0) {
        // the sweeps cost N scalar stage steps per wavefront whatever L is: spread a small batch over all SIMDs
        // (one wavefront each), pack a large one
        // (measured, profiles/r03_c_block_kernel_experiments.txt: two wavefronts on a SIMD do not issue faster than one,
        // so L = 32 only pays while every wavefront still has a CU to itself)
        // B_in_flight: problems of all launches that run concurrently with this one (alore_nmpc_rti_many) -- what fills
        // the chip is their sum
        const long Bo = (B_in_flight > B) ? B_in_flight : B;
        L = 16;
        while (L > 4 && (Bo + 64 / L - 1) / (64 / L) > 4L * cus) L >>= 1;
        while (L < 16 && N > L * (L == 4 ? 5 : 3)) L <<= 1;
        if (L == 16 && N <= 32 && (Bo + 1) / 2 <= (long)cus) L = 32;
    }
    int S = 0;
    if (L == 4 && N <= 20) S = 5;
    else if (L == 8 && N <= 24) S = 3;
    else if (L == 16 && N <= 32) S = 2;
    else if (L == 16 && N <= 64) S = 4;
    else if (L == 32 && N <= 32) S = 1;
    if (S == 0) return false;
    const size_t lds = (size_t)block_lds_floats(N, L) * 4;
    if ((long)lds > lds_limit_bytes) return false;
    g->L = L;
    g->G = 64 / L;
    g->wpb = 1;
    g->wreg = 0;
    g->threads = 64;
    g->grid = (B + g->G - 1) / g->G;
    g->RS = S; // stages per lane
    g->lds_bytes = lds;
    g->block = 1;
    return true;
}

hipError_t launch_rti_block(const RtiParams& p, const LaunchGeom& g, hipStream_t s)
{
    RtiGroup grp;
    grp.count = 1;
    grp.blocks_per_batch = g.grid;
    grp.strided = 0;
    grp.stagger_blocks = 0;
    grp.stagger_x1024 = 0;
    grp.trace = nullptr;
    grp.counter = nullptr;
    grp.persist_blocks = 0;
    grp.xcd_on = 0;
    grp.xcd_end = nullptr;
    grp.b[0] = p.b;
    return launch_rti_block_group(p, grp, g, s);
}

// `grp.count` batches of p.B problems each in one grid (g.grid = blocks of ONE batch); every batch of the group has the
// same set of diagnostics pointers (checked by the caller)
hipError_t launch_rti_block_group(const RtiParams& p, const RtiGroup& grp, const LaunchGeom& g, hipStream_t s)
{
    if (grp.count < 1 || (!grp.strided && grp.count > GROUP_MAX) || grp.blocks_per_batch != g.grid) return hipErrorInvalidValue;
    if ((long long)g.grid * grp.count > 0x7fffffffLL) return hipErrorInvalidValue;
    const bool stamp = p.stamps != nullptr;
    const bool diag = stamp || grp.b[0].kkt != nullptr || grp.b[0].obj != nullptr;
    const bool once = p.n_sqp == 1;
    const void* fn = nullptr;
    int v = -1;
#define PICK(LL, SS, idx)                                                                                 \
    if (g.L == LL && g.RS == SS) {                                                                        \
        v = idx * 5 + (stamp ? 4 : ((diag ? 0 : 1) + (once ? 2 : 0)));                                    \
        fn = stamp ? (const void*)rti_block_kernel<LL, SS, true, true, false>                             \
                   : (diag ? (once ? (const void*)rti_block_kernel<LL, SS, true, false, true>             \
                                   : (const void*)rti_block_kernel<LL, SS, true, false, false>)           \
                           : (once ? (const void*)rti_block_kernel<LL, SS, false, false, true>            \
                                   : (const void*)rti_block_kernel<LL, SS, false, false, false>));        \
    }
    PICK(4, 5, 0)
    PICK(8, 3, 1)
    PICK(16, 2, 2)
    PICK(16, 4, 3)
    PICK(32, 1, 4)
#undef PICK
    if (g.L == 4 && g.RS == 5 && p.N == 20 && once && !stamp) { // the control tick at the horizon that fills the (4, 5) mapping
        v = 25 + (diag ? 0 : 1);
        fn = diag ? (const void*)rti_block_kernel<4, 5, true, false, true, true> : (const void*)rti_block_kernel<4, 5, false, false, true, true>;
    }
    const bool persist = grp.counter != nullptr;
    if (persist) {
        if (!(g.L == 4 && g.RS == 5 && p.N == 20 && once && !stamp)) return hipErrorInvalidValue;
        v = 28 + (diag ? 0 : 1);
        fn = diag ? (const void*)rti_block_kernel<4, 5, true, false, true, true, false, true> : (const void*)rti_block_kernel<4, 5, false, false, true, true, false, true>;
    }
    if (grp.trace) { // diagnostic: only the grid builds have an instrumented twin
        if (!(g.L == 4 && g.RS == 5 && p.N == 20 && once && !stamp && diag)) return hipErrorInvalidValue;
        v = persist ? 30 : 27;
        fn = persist ? (const void*)rti_block_kernel<4, 5, true, false, true, true, true, true> : (const void*)rti_block_kernel<4, 5, true, false, true, true, true>;
    }
    if (!fn) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 15;
    static size_t configured[16][31] = {{0}};
    if (g.lds_bytes > configured[dev][v]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
        configured[dev][v] = g.lds_bytes;
    }
    void* args[] = {const_cast<RtiParams*>(&p), const_cast<RtiGroup*>(&grp)};
    unsigned blocks = (unsigned)g.grid * (unsigned)grp.count;
    if (persist && blocks > (unsigned)grp.persist_blocks) blocks = (unsigned)grp.persist_blocks;
    if (!persist && grp.xcd_on) {
        int mx = 0;
        for (int x = 0; x < 8; ++x) mx = grp.xcd_share[x] > mx ? grp.xcd_share[x] : mx;
        blocks = 8u * (unsigned)mx;
    }
    e = hipLaunchKernel(fn, dim3(blocks), dim3(64), args, g.lds_bytes, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

} // namespace nmpc
