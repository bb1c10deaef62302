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
#include "ref_sampler_device.h"

#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "nmpc_core.h"
#include "nmpc_scan.h"

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
// TWOPH (grid builds): the batches of the grid are solved in two passes -- see the top of nmpc_block_body.inc.
// -DALORE_EXP_WPE=n (experiments, never the shipped build): at least n wavefronts per SIMD, i.e. at most 512 / n registers
#ifdef ALORE_EXP_WPE
#define ALORE_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(ALORE_EXP_WPE, ALORE_EXP_WPE)))
#else
#define ALORE_KERNEL_ATTR
#endif
template <int L, int S, bool DIAG, bool STAMP, bool ONCE, bool FULLN = false, bool TRACE = false, bool PERSIST = false, bool TWOPH = false>
__global__ __launch_bounds__(64) ALORE_KERNEL_ATTR void rti_block_kernel(const RtiParams p_arg, const RtiGroup grp_arg)
{
    constexpr bool SCAN_BUILD = true;
#include "nmpc_block_body.inc"
}

// One grid for the solve of a tick AND the pose-independent sampling of the next tick's references (alore_nmpc_closed_loop_run:
// AheadSampler, ref_sampler_device.h): workgroups [0, first_block) are the solver's, the rest walk two robots each.  The
// sampler's wavefronts run beside the solver's on the same SIMDs -- the solver's chain leaves four issue slots in ten idle --
// where a kernel of its own would stand in the tick's chain (kernels of one stream run one after the other; a second stream costs
// two cross-stream events per tick, ~5 us each on this stack: tools/micro/cross_stream.hip; hipExtAnyOrderLaunch does not overlap
// kernels of one stream on gfx950: tools/micro/any_order.hip).
// `pl.on`: the plant step of the PREVIOUS tick runs in front of the solve, on the first lane of each problem's group (it produces the
// pose the solve starts from: x0, and the turns its references are shifted by) -- as a kernel of its own it cost 4.7 us of the
// tick's chain for three memory round trips and a microsecond of arithmetic.
template <int L, int S, bool DIAG>
__global__ __launch_bounds__(64) void rti_block_sampler_kernel(const RtiParams p_arg, const RtiGroup grp_arg, const AheadSampler sa, const PlantAhead pl)
{
    if ((int)blockIdx.x < sa.first_block && pl.on) {
        const int r = (int)blockIdx.x * (64 / L) + (int)threadIdx.x / L;
        if ((int)threadIdx.x % L == 0 && r < pl.B) plant_ahead_one(pl, r);
        // the solve's loads below (vector loads and LDS-DMA, this wavefront's) read what these lanes stored
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if ((int)blockIdx.x >= sa.first_block) {
        alore_nmpc_batch nb{};
        nb.y = sa.y;
        nb.yN = sa.yN;
        ref_sample_smooth_body<32, true>(sa.store, nb, sa.B, sa.N, sa.dt, sa.now, nullptr, sa.icr, nullptr, sa.psi_rel, (int)blockIdx.x - sa.first_block,
                                         (int)threadIdx.x, 64);
        return;
    }
    constexpr bool STAMP = false, ONCE = true, FULLN = false, TRACE = false, PERSIST = false, TWOPH = false;
    // the (16, 2) build keeps the sequential backward sweep here: with the scan over the lanes (nmpc_scan.h) it takes 284 registers, and
    // the sampler's wavefronts run beside the solver's only while a SIMD holds one of each (256): the tick of 4096 robots went from 19
    // to 27 us; capped at 256 registers (16 - 22 spilled) 21.0 against 19.5 us.  (32, 1) has room: 256 robots 14.3 -> 11.9 us per tick.
    constexpr bool SCAN_BUILD = (L == 32);
#include "nmpc_block_body.inc"
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
        // (long horizons -- the reference's N = 50 -- stay on (16, 4): 8 lanes x 7 stages with W_k in registers instead of LDS was built in
        // round 6 and spills 880 registers: 49.6 against 35 us per batch in flight; what took (16, 4) to 26 us is the scan of nmpc_scan.h)
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
    grp.tp_count2 = 0; grp.tp_tail = 0; grp.tp_lag = 0; grp.tp_timeout = 0; grp.tp_exits = nullptr; grp.tp_cnt = nullptr; grp.tp_entries = nullptr; grp.tp_trace = nullptr; grp.tp_rec = nullptr;
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
    if (g.L == 4 && g.RS == 5 && p.N == 20 && !once && !stamp) { // several iterations per launch (converged solves) at the horizon that fills the mapping
        v = 33 + (diag ? 0 : 1);
        fn = diag ? (const void*)rti_block_kernel<4, 5, true, false, false, true> : (const void*)rti_block_kernel<4, 5, false, false, false, true>;
    }
    const bool persist = grp.counter != nullptr;
    if (persist) {
        if (!(g.L == 4 && g.RS == 5 && p.N == 20 && once && !stamp)) return hipErrorInvalidValue;
        v = 28 + (diag ? 0 : 1);
        fn = diag ? (const void*)rti_block_kernel<4, 5, true, false, true, true, false, true> : (const void*)rti_block_kernel<4, 5, false, false, true, true, false, true>;
    }
    const bool twoph = grp.tp_count2 > 0;
    if (twoph) {
        if (!(g.L == 4 && g.RS == 5 && p.N == 20 && once && !stamp) || persist || grp.trace || grp.xcd_on) return hipErrorInvalidValue;
        v = 31 + (diag ? 0 : 1);
        fn = diag ? (const void*)rti_block_kernel<4, 5, true, false, true, true, false, false, true> : (const void*)rti_block_kernel<4, 5, false, false, true, true, false, false, true>;
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
    static size_t configured[16][35] = {{0}};
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
    if (twoph) { // units of (blocks of a batch, tail of the batch tp_lag units before): see nmpc_block_body.inc
        const long long units = std::max((long long)grp.count, (long long)grp.tp_count2 + grp.tp_lag);
        const long long nb = units * ((long long)g.grid + grp.tp_tail);
        if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
        blocks = (unsigned)nb;
    }
    e = hipLaunchKernel(fn, dim3(blocks), dim3(64), args, g.lds_bytes, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

bool rti_block_two_phase_supported(const RtiParams& p, const LaunchGeom& g)
{
    return g.block && g.L == 4 && g.RS == 5 && p.N == 20 && p.n_sqp == 1 && p.stamps == nullptr && g.grid <= 64 * TP_KMAX;
}

// The solve of one batch with the sampler's workgroups behind it in the same grid; false when this (mapping, mode) has no such build
// (the caller then launches the two kernels one after the other)
// 2: the sampler's workgroups run beside the solver's (builds of at most 256 registers: a SIMD holds one wavefront of each); 1: the
// grid carries the plant step only (the (8, 3) build takes 303 registers: the sampler's wavefronts would queue behind the solver's
// with one slot per SIMD -- as a kernel of its own the sampler has eight); 0: neither
int rti_block_sampler_supported(const RtiParams& p, const LaunchGeom& g)
{
    if (!(g.block && p.n_sqp == 1 && p.stamps == nullptr && p.N + 1 <= 32)) return 0;
    if ((g.L == 16 && g.RS == 2) || (g.L == 32 && g.RS == 1)) return 2;
    if (g.L == 8 && g.RS == 3) return 1;
    return 0;
}
hipError_t launch_rti_block_sampler(const RtiParams& p, const LaunchGeom& g, const AheadSampler& sa_in, const PlantAhead* plant, hipStream_t s)
{
    const int kind = rti_block_sampler_supported(p, g);
    if (kind == 0 || (kind == 1 && sa_in.B > 0)) return hipErrorInvalidValue;
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
    grp.tp_count2 = 0; grp.tp_tail = 0; grp.tp_lag = 0; grp.tp_timeout = 0; grp.tp_exits = nullptr; grp.tp_cnt = nullptr; grp.tp_entries = nullptr; grp.tp_trace = nullptr; grp.tp_rec = nullptr;
    for (int m = 0; m < 15; ++m) grp.stride[m] = 0;
    grp.b[0] = p.b;
    const bool diag = p.b.kkt != nullptr || p.b.obj != nullptr;
    const int v = (g.L == 16 ? 0 : (g.L == 32 ? 2 : 4)) + (diag ? 0 : 1);
    const void* fn = g.L == 16 ? (diag ? (const void*)rti_block_sampler_kernel<16, 2, true> : (const void*)rti_block_sampler_kernel<16, 2, false>)
                   : g.L == 32 ? (diag ? (const void*)rti_block_sampler_kernel<32, 1, true> : (const void*)rti_block_sampler_kernel<32, 1, false>)
                               : (diag ? (const void*)rti_block_sampler_kernel<8, 3, true> : (const void*)rti_block_sampler_kernel<8, 3, false>);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 15;
    static size_t configured[16][6] = {{0}};
    if (g.lds_bytes > configured[dev][v]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
        configured[dev][v] = g.lds_bytes;
    }
    AheadSampler sa = sa_in;
    sa.first_block = g.grid;
    PlantAhead pl{};
    if (plant) { pl = *plant; pl.on = 1; }
    static_assert(sizeof(RtiParams) + sizeof(RtiGroup) + sizeof(AheadSampler) + sizeof(PlantAhead) + 32 <= 4096, "kernel arguments: 4 KB");
    void* args[] = {const_cast<RtiParams*>(&p), &grp, &sa, &pl};
    const unsigned blocks = (unsigned)g.grid + (unsigned)((sa.B + 1) / 2);
    e = hipLaunchKernel(fn, dim3(blocks), dim3(64), args, g.lds_bytes, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

} // namespace nmpc
