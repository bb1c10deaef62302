// backend_kernels.hip -- the reference's back_end optimiser (MSPlanner, P/back_end/src/optimizer.cpp) for a batch
// of independent FlatTrajData problems: ONE WAVEFRONT PER PROBLEM, the whole optimisation inside one launch.
//
// What one wavefront does is MSPlanner::minco_plan (optimizer.cpp:169-220): up to safeReplanMaxTime passes of
//   stage 1  lbfgs_optimize(costFunctionCallbackPath)                      :303-309, 1272-1591
//   stage 2  ALM loop { lbfgs_optimize(costFunctionCallback); lambda, rho update }   :376-418, 631-1067
//   final collision check on a dense Simpson resample                       :474-571
// in float64.  Inside a cost evaluation the 64 lanes work on
//   - the knot system of the minimum-jerk spline (csrc/minco_spline.h: SPD 2 x 2 block-tridiagonal instead
//     of the reference's 6M x 6M band LU), solved by parallel cyclic reduction, lane = knot (knot_pcr below),
//   - (piece, dimension) for the Hermite coefficients, the energy and the adjoint,
//   - the M x 17 Simpson nodes for everything in attachPenaltyFunctional (lane = node, 64 per round),
//   - wave prefix / suffix scans for the pose integration (every pose depends on all earlier Simpson panels)
//     and for the chain rule back through it (optimizer.cpp:944-947, 1054-1066),
//   - (piece, power, dimension) for the reduction of node terms into the coefficient gradient.
// All sums are taken in a fixed order (butterfly reductions, ascending node order): results are bit-reproducible.
// L-BFGS (lbfgs.hpp:440-756, Lewis-Overton line search :276-396) keeps x, g, d in LDS, lane = variable; the
// (s, y) history (mem_size = 256 pairs, 3M - 1 doubles each) lives in HBM, one contiguous slab per problem.
// Control flow is wave-uniform by construction (every decision is taken on a butterfly-reduced value).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "backend_kernels.h"
#include "minco_spline.h"

namespace backend {

namespace {

// Address spaces are spelled out.  Inside the (noinline) device functions below a plain pointer is a generic one: the
// compiler then reaches LDS through flat instructions and looks the base of the dynamic LDS block up in a table at every
// access site, and reads the parameter block with vector loads.  With typed pointers the same accesses are ds_* instructions
// on a base handed down from the kernel, scalar loads of the (read-only) parameter block and global_* loads / stores.
#define LDSQ __attribute__((address_space(3)))
#define GLBQ __attribute__((address_space(1)))
#define CSTQ __attribute__((address_space(4)))
typedef const CSTQ Params CParams;
typedef const CSTQ Config CConfig;
typedef const CSTQ MapView CMapView;
typedef const CSTQ LbfgsParam CLbfgsParam;

// diagnostic phase stamps: workgroup 0 adds the cycles since its previous stamp to slot i (global stores: a pending FLAT
// access would make every later wait on the vector-memory counter a wait for everything)
#define BE_STAMP(i)                                                                              \
    if (gp->stamps && blockIdx.x == 0 && threadIdx.x == 0) {                                     \
        GLBQ long long* st_ = (GLBQ long long*)gp->stamps;                                       \
        const long long now_ = (long long)__builtin_readcyclecounter();                          \
        st_[i] += now_ - st_[63];                                                                \
        st_[63] = now_;                                                                          \
    }

constexpr int NS = 17; // Simpson nodes per piece (sparseResolution 8)
constexpr int RES = 8;
static_assert(RES * 6 == 48, "div_res6 divides by the literal 48");

// Wavefront reductions on the DPP data path (no LDS crossbar): inclusive scan inside each row of 16 lanes
// (row_shr 1, 2, 4, 8), row totals passed on with row_bcast15 / row_bcast31, result read from lane 63 and returned
// wave-uniform.  A double moves as two 32-bit halves.  The summation order is fixed, so results are reproducible.
// Callers are in wave-uniform control flow (all 64 lanes active).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double x)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// The total only: the value of lane 63 of the scan below, computed by the same additions in the same order -- but nothing
// else has to be right, so the moves need no `old` operand (row shifts fill with zeros through bound_ctrl, the broadcasts go
// to every row they reach: what lands in lanes that do not feed lane 63 is never looked at): three instructions per level
// instead of six.
template <int CTRL>
__device__ __forceinline__ double dpp_f64_nofill(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64_nofill<0x111>(v); // row_shr:1
    v += dpp_f64_nofill<0x112>(v); // row_shr:2
    v += dpp_f64_nofill<0x114>(v); // row_shr:4
    v += dpp_f64_nofill<0x118>(v); // row_shr:8: lanes 15, 31, 47, 63 hold their rows' sums
    v += dpp_f64_nofill<0x142>(v); // row_bcast15: lane 31 = rows 0 + 1, lane 63 = rows 2 + 3
    v += dpp_f64_nofill<0x143>(v); // row_bcast31: lane 63 += lane 31
    return lane63(v);
}
// inclusive prefix sum over the 64 lanes (the scan wave_sum reads its total from); suffix sums use it on lanes that hold
// their items in reverse order (eval_cost: chain coefficients)
__device__ __forceinline__ double wave_prefix(double v)
{
    v += dpp_f64<0x111, 0xF>(0.0, v);
    v += dpp_f64<0x112, 0xF>(0.0, v);
    v += dpp_f64<0x114, 0xF>(0.0, v);
    v += dpp_f64<0x118, 0xF>(0.0, v);
    v += dpp_f64<0x142, 0xA>(0.0, v);
    v += dpp_f64<0x143, 0xC>(0.0, v);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_f64<0x111, 0xF>(v, v)); // lanes without a source keep `old` = their own value
    v = fmax(v, dpp_f64<0x112, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x114, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x118, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x142, 0xA>(v, v));
    v = fmax(v, dpp_f64<0x143, 0xC>(v, v));
    return lane63(v);
}
__device__ __forceinline__ double uni(double v) // value known to be wave-uniform -> SGPR pair
{
    const unsigned lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Arguments of the (noinline) device functions below arrive in vector registers and count as divergent: every loop bound,
// branch and address formed from them would run on the vector unit under EXEC masks, with conservative waits at the block
// boundaries.  They are wave-uniform by construction -- the functions move them to scalar registers first.
template <class T>
__device__ __forceinline__ T* uni_ptr(T* q)
{
    const unsigned long long b = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ LbfgsParam uni(const LbfgsParam& q)
{
    LbfgsParam r;
    r.mem_size = uni(q.mem_size); r.past = uni(q.past); r.max_iterations = uni(q.max_iterations); r.max_linesearch = uni(q.max_linesearch);
    r.g_epsilon = uni(q.g_epsilon); r.delta = uni(q.delta); r.min_step = uni(q.min_step); r.max_step = uni(q.max_step);
    r.f_dec_coeff = uni(q.f_dec_coeff); r.s_curv_coeff = uni(q.s_curv_coeff); r.cautious_factor = uni(q.cautious_factor);
    r.machine_prec = uni(q.machine_prec);
    return r;
}

// the two coefficients of the cubic branch depend on eps alone: formed once per evaluation (two float64 divisions that every
// active penalty term used to repeat on its own chain)
struct SmoothL1 {
    double eps, f3, f4;
    __device__ __forceinline__ explicit SmoothL1(double e) : eps(e), f3(1.0 / (e * e)), f4(-0.5 * (1.0 / (e * e)) / e) {}
};
__device__ __forceinline__ void smoothed_l1(const SmoothL1& k, double x, double& f, double& df)
{
    if (x < k.eps) {
        f = (k.f4 * x + k.f3) * x * x * x;
        df = (4.0 * k.f4 * x + 3.0 * k.f3) * x * x;
    } else {
        f = x - 0.5 * k.eps;
        df = 1.0;
    }
}
// x / (RES * 6), the Simpson weight of the reference, correctly rounded like the division it replaces (div_by_known)
__device__ __forceinline__ double div_res6(double a)
{
    constexpr double b = 48.0, r = 1.0 / 48.0;
    const double q0 = a * r;
    const double q1 = fma(fma(-b, q0, a), r, q0);
    return fma(fma(-b, q1, a), r, q1);
}
__device__ __forceinline__ double t_of_tau(double v) { return v > 0.0 ? ((0.5 * v + 1.0) * v + 1.0) : 1.0 / ((0.5 * v - 1.0) * v + 1.0); }
__device__ __forceinline__ double tau_of_t(double t) { return t > 1.0 ? (sqrt(2.0 * t - 1.0) - 1.0) : (1.0 - sqrt(2.0 / t - 1.0)); }
__device__ __forceinline__ double dt_dtau(double v)
{
    if (v > 0) return v + 1.0;
    const double den = (0.5 * v - 1.0) * v + 1.0;
    return (1.0 - v) / (den * den);
}

// SDFmap::getDistWithGradBilinear(pos, grad, mindis) (sdf_map.cpp:796-834) and (pos) (:836-861, want_grad = false), in two halves:
// esdf_fetch finds the cell and issues the four loads, esdf_eval interpolates -- so that a caller can have the loads of several
// queries in flight before it needs the first value (a query is one trip to L2 / HBM).
// (inv = 1 / m.res, formed once by the caller: a float64 division on the chain of every query otherwise)
struct EsdfCell {
    double v00, v01, v10, v11, fx, fy;
    bool inside;
};
__device__ __forceinline__ EsdfCell esdf_fetch(CMapView& m, double inv, double x, double y)
{
    EsdfCell r{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, false};
    if (x < m.x_lo || y < m.y_lo || x > m.x_hi || y > m.y_hi) return r;
    int ix = (int)((x - m.x_lo) * inv - 0.5), iy = (int)((y - m.y_lo) * inv - 0.5);
    ix = min(max(ix, 0), m.nx - 1);
    iy = min(max(iy, 0), m.ny - 1);
    if (ix >= m.nx - 1 || iy >= m.ny - 1) return r;
    r.fx = (x - ((ix + 0.5) * m.res + m.x_lo)) * inv;
    r.fy = (y - ((iy + 0.5) * m.res + m.y_lo)) * inv;
    const GLBQ double* c = (const GLBQ double*)m.dist + (size_t)ix * m.ny + iy;
    r.v00 = c[0]; r.v01 = c[1]; r.v10 = c[m.ny]; r.v11 = c[m.ny + 1];
    r.inside = true;
    return r;
}
__device__ __forceinline__ double esdf_eval(const EsdfCell& q, double inv, bool want_grad, double mindis, double& gx, double& gy)
{
    gx = 0.0; gy = 0.0;
    if (!q.inside) return 1e10;
    const double fx = q.fx, fy = q.fy, v00 = q.v00, v01 = q.v01, v10 = q.v10, v11 = q.v11;
    const double lo = (1 - fx) * v00 + fx * v10, hi = (1 - fx) * v01 + fx * v11;
    const double dist = (1 - fy) * lo + fy * hi;
    if (!want_grad || dist > mindis) return dist;
    gy = (hi - lo) * inv;
    gx = ((1 - fy) * (v10 - v00) + fy * (v11 - v01)) * inv;
    return dist;
}
__device__ __forceinline__ double esdf(CMapView& m, double inv, double x, double y, bool want_grad, double mindis, double& gx, double& gy)
{
    return esdf_eval(esdf_fetch(m, inv, x, y), inv, want_grad, mindis, gx, gy);
}

// everything a cost evaluation needs besides x; lives in LDS (wave-uniform, written by lane 0 + barrier or by
// all lanes with the same value)
struct EvalCtx {
    int M, n, stage, evals, prob;
    double lam[2], rho[2], safe_dis, time_weight;
    const double* positions; // [M][2] way-points + final (stage 1)
    double head[2][3], tail[2][3], start_xy[2], final_xy[2];
    double xy_err[2]; // out
};

template <int P>
struct Lds {
    EvalCtx e;
    double T[P], iT[P]; // piece durations and their reciprocals
    double kp[2][P + 1], kv[2][P + 1], ka[2][P + 1]; // knot states per flat dimension
    double y[2][P][2];                                // knot system right-hand side / solution per dimension
    double coef[P * 12];                              // (6 i + q) * 2 + d
    double gdC[P * 12];
    double gdT[P];
    union { // the two-loop recursion of L-BFGS runs between two cost evaluations: its alpha[] takes the bytes of a node array of the evaluation
        double fx[P * NS]; // the Simpson integrands, then the chain coefficients (cos / sin of the heading per node: global workspace)
        double alpha[MEM_MAX];
    };
    double fy[P * NS];
    // node terms of the coefficient gradient, [order 0 1 2][dimension theta s] per node -- stored without the entries that are zero
    // by construction (no penalty acts on s itself: order 0 of s is never written; the order-2 terms come from the penalties of
    // pass A, which sit on the even nodes only): Ea = [node][order 0 of theta, order 1 of theta, order 1 of s] here, [even node][order 2 of theta,
    // order 2 of s] and the duration-gradient term per node in the problem's global workspace (Params::pcr: written once or twice,
    // read once per evaluation).  6.5 KB of LDS instead of 15: six workgroups per CU
    double Ea[P * NS * 3];
    union { // the pose at the Simpson panel ends is dead (after the pose-dependent terms) before the adjoint of the spline begins
        double posx[RES * P + 1];
        double gk[2][P + 1][3];                       // gradient w.r.t. the knot states
    };
    double posy[RES * P + 1];
    double x[3 * P], g[3 * P], d[3 * P], xp[3 * P], gp[3 * P];
    double pf[16];
};

extern __shared__ __align__(16) unsigned char lds_raw[];
// the kernel takes the offset of the dynamic LDS block (the low half of its generic address) and hands it down
// (through an empty asm: as a constant expression the address would be propagated into the callees, which would go back to
// the table lookup)
__device__ __forceinline__ unsigned lds_base_of_kernel()
{
    unsigned b = (unsigned)(unsigned long long)lds_raw;
    asm volatile("; dynamic LDS block at %0" : "+s"(b));
    return b;
}
template <int P>
__device__ __forceinline__ LDSQ Lds<P>& lds(unsigned base) { return *(LDSQ Lds<P>*)(unsigned long)base; }

// Knot system by parallel cyclic reduction: equation k (interior knot k = lane + 1) is
//   L_k y_{k-1} + D_k y_k + U_k y_{k+1} = r_k,   L_k = U_{k-1}',   2 x 2 blocks (csrc/minco_spline.h: knot_diag, knot_upper).
// One lane per knot, both flat dimensions as two right-hand sides.  A step with stride s eliminates the neighbours at
// distance s (alpha = -L D_{k-s}^-1, gamma = -U D_{k+s}^-1); after ceil(log2 nk) steps every equation stands alone.
// The sequential block Thomas sweeps this replaces kept one or two lanes busy for ~18 k cycles per solve; here the
// dependent chain is 4 - 5 steps.  The matrix is symmetric, so the adjoint system of the gradient is the same call.
struct B2 { double a, b, c, d; }; // [[a, b], [c, d]]
__device__ __forceinline__ B2 mul(const B2& x, const B2& y) { return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; }
__device__ __forceinline__ B2 inv2(const B2& m)
{
    const double id = 1.0 / (m.a * m.d - m.b * m.c);
    return {m.d * id, -m.b * id, -m.c * id, m.a * id};
}
// neighbour at distance S along the lanes: through the DPP row shifts when all knots sit in one row of 16 lanes (M <= 16:
// a float64 costs two 32-bit DPP moves at VALU speed), through the LDS permute otherwise (two ds_bpermute per float64,
// ~100 cycles of latency each time: 64 values per reduction step)
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int S, bool DPP> __device__ __forceinline__ double lane_up(double x) { if constexpr (DPP) return dpp_mov64<0x110 + S>(x); else return __shfl_up(x, S); }   // lane i <- lane i - S
template <int S, bool DPP> __device__ __forceinline__ double lane_down(double x) { if constexpr (DPP) return dpp_mov64<0x100 + S>(x); else return __shfl_down(x, S); } // lane i <- lane i + S
template <int S, bool DPP> __device__ __forceinline__ B2 lane_up(const B2& m) { return {lane_up<S, DPP>(m.a), lane_up<S, DPP>(m.b), lane_up<S, DPP>(m.c), lane_up<S, DPP>(m.d)}; }
template <int S, bool DPP> __device__ __forceinline__ B2 lane_down(const B2& m) { return {lane_down<S, DPP>(m.a), lane_down<S, DPP>(m.b), lane_down<S, DPP>(m.c), lane_down<S, DPP>(m.d)}; }

struct PcrState { B2 Lk, Dk, Uk; double r0, r1, r2, r3; };

template <int S, bool DPP>
__device__ __forceinline__ void pcr_step(PcrState& q, bool act, int e, int nk, double (&fac)[8])
{
    const B2 Lm = lane_up<S, DPP>(q.Lk), Dm = lane_up<S, DPP>(q.Dk), Um = lane_up<S, DPP>(q.Uk);
    const double m0 = lane_up<S, DPP>(q.r0), m1 = lane_up<S, DPP>(q.r1), m2 = lane_up<S, DPP>(q.r2), m3 = lane_up<S, DPP>(q.r3);
    const B2 Lp = lane_down<S, DPP>(q.Lk), Dp = lane_down<S, DPP>(q.Dk), Up = lane_down<S, DPP>(q.Uk);
    const double p0 = lane_down<S, DPP>(q.r0), p1 = lane_down<S, DPP>(q.r1), p2 = lane_down<S, DPP>(q.r2), p3 = lane_down<S, DPP>(q.r3);
    const bool hm = act && e - S >= 0, hp = act && e + S < nk;
    B2 al{0, 0, 0, 0}, ga{0, 0, 0, 0};
    if (hm) { const B2 t = mul(q.Lk, inv2(Dm)); al = {-t.a, -t.b, -t.c, -t.d}; }
    if (hp) { const B2 t = mul(q.Uk, inv2(Dp)); ga = {-t.a, -t.b, -t.c, -t.d}; }
    fac[0] = al.a; fac[1] = al.b; fac[2] = al.c; fac[3] = al.d; fac[4] = ga.a; fac[5] = ga.b; fac[6] = ga.c; fac[7] = ga.d;
    if (hm) {
        const B2 t = mul(al, Um);
        q.Dk = {q.Dk.a + t.a, q.Dk.b + t.b, q.Dk.c + t.c, q.Dk.d + t.d};
        q.r0 += al.a * m0 + al.b * m1; q.r1 += al.c * m0 + al.d * m1;
        q.r2 += al.a * m2 + al.b * m3; q.r3 += al.c * m2 + al.d * m3;
        q.Lk = mul(al, Lm);
    } else q.Lk = {0, 0, 0, 0};
    if (hp) {
        const B2 t = mul(ga, Lp);
        q.Dk = {q.Dk.a + t.a, q.Dk.b + t.b, q.Dk.c + t.c, q.Dk.d + t.d};
        q.r0 += ga.a * p0 + ga.b * p1; q.r1 += ga.c * p0 + ga.d * p1;
        q.r2 += ga.a * p2 + ga.b * p3; q.r3 += ga.c * p2 + ga.d * p3;
        q.Uk = mul(ga, Up);
    } else q.Uk = {0, 0, 0, 0};
}

// the same elimination on right-hand sides only, with the factors a factorising call left in the lane's registers (the knot matrix of an
// evaluation serves twice: spline, then its adjoint): a step is 16 shifted values + 16 multiply-adds instead of two 2 x 2
// inverses and four block products
template <int S, bool DPP>
__device__ __forceinline__ void pcr_replay(double (&r)[4], const double (&fac)[8])
{
    const double m0 = lane_up<S, DPP>(r[0]), m1 = lane_up<S, DPP>(r[1]), m2 = lane_up<S, DPP>(r[2]), m3 = lane_up<S, DPP>(r[3]);
    const double p0 = lane_down<S, DPP>(r[0]), p1 = lane_down<S, DPP>(r[1]), p2 = lane_down<S, DPP>(r[2]), p3 = lane_down<S, DPP>(r[3]);
    const B2 al{fac[0], fac[1], fac[2], fac[3]}, ga{fac[4], fac[5], fac[6], fac[7]}; // zeros where a neighbour is missing
    r[0] += al.a * m0 + al.b * m1; r[1] += al.c * m0 + al.d * m1;
    r[2] += al.a * m2 + al.b * m3; r[3] += al.c * m2 + al.d * m3;
    r[0] += ga.a * p0 + ga.b * p1; r[1] += ga.c * p0 + ga.d * p1;
    r[2] += ga.a * p2 + ga.b * p3; r[3] += ga.c * p2 + ga.d * p3;
}

// rhs / solution: y[d][lane][0..1] in LDS for d = 0, 1 (the caller's layout); all 64 lanes must call.  REPLAY = false
// factorises (and leaves the factors of the lane's knot in F / D: registers of the caller -- 4.6 KB of LDS until round 5, which
// cost a workgroup per CU), REPLAY = true solves with the factors of the last factorising call.
struct PcrFactors { double F[5][8], D[4]; };
template <int P, bool REPLAY>
__device__ __forceinline__ void knot_pcr(LDSQ Lds<P>& L, int M, const LDSQ double* T, LDSQ double (*y0)[2], LDSQ double (*y1)[2], PcrFactors& pf)
{
    constexpr bool DPP = P <= 16; // all knots (<= 15) in one DPP row
    constexpr int PL = P <= 16 ? 16 : 32;
    const int e = threadIdx.x, nk = M - 1;
    const bool act = e < nk;
    (void)PL; // idle lanes (e >= nk) compute zero factors of their own (no neighbour on either side) and never use D
    if constexpr (REPLAY) {
        double r[4] = {0.0, 0.0, 0.0, 0.0};
        if (act) { r[0] = y0[e][0]; r[1] = y0[e][1]; r[2] = y1[e][0]; r[3] = y1[e][1]; }
        if (1 < nk) pcr_replay<1, DPP>(r, pf.F[0]);
        if (2 < nk) pcr_replay<2, DPP>(r, pf.F[1]);
        if (4 < nk) pcr_replay<4, DPP>(r, pf.F[2]);
        if (8 < nk) pcr_replay<8, DPP>(r, pf.F[3]);
        if constexpr (P > 16) { if (16 < nk) pcr_replay<16, false>(r, pf.F[4]); }
        if (act) {
            const B2 di{pf.D[0], pf.D[1], pf.D[2], pf.D[3]};
            y0[e][0] = di.a * r[0] + di.b * r[1]; y0[e][1] = di.c * r[0] + di.d * r[1];
            y1[e][0] = di.a * r[2] + di.b * r[3]; y1[e][1] = di.c * r[2] + di.d * r[3];
        }
        return;
    }
    PcrState q{{0, 0, 0, 0}, {1, 0, 0, 1}, {0, 0, 0, 0}, 0.0, 0.0, 0.0, 0.0};
    if (act) {
        const int k = e + 1;
        const minco::InvT l(T[k - 1], L.iT[k - 1]), r(T[k], L.iT[k]);
        const minco::Sym2 dg = minco::knot_diag(l, r);
        q.Dk = {dg.a, dg.b, dg.b, dg.c};
        if (k > 1) { const minco::Mat2 u = minco::knot_upper(l); q.Lk = {u.a, u.c, u.b, u.d}; } // U_{k-1}'
        if (k < M - 1) { const minco::Mat2 u = minco::knot_upper(r); q.Uk = {u.a, u.b, u.c, u.d}; }
        q.r0 = y0[e][0]; q.r1 = y0[e][1]; q.r2 = y1[e][0]; q.r3 = y1[e][1];
    }
    if (1 < nk) pcr_step<1, DPP>(q, act, e, nk, pf.F[0]);
    if (2 < nk) pcr_step<2, DPP>(q, act, e, nk, pf.F[1]);
    if (4 < nk) pcr_step<4, DPP>(q, act, e, nk, pf.F[2]);
    if (8 < nk) pcr_step<8, DPP>(q, act, e, nk, pf.F[3]);
    if constexpr (P > 16) { if (16 < nk) pcr_step<16, false>(q, act, e, nk, pf.F[4]); }
    if (act) {
        const B2 di = inv2(q.Dk);
        pf.D[0] = di.a; pf.D[1] = di.b; pf.D[2] = di.c; pf.D[3] = di.d;
        y0[e][0] = di.a * q.r0 + di.b * q.r1; y0[e][1] = di.c * q.r0 + di.d * q.r1;
        y1[e][0] = di.a * q.r2 + di.b * q.r3; y1[e][1] = di.c * q.r2 + di.d * q.r3;
    }
}

// One cost callback.  x in L.x, gradient to L.g.  Returns the cost (wave-uniform).  *skipped is set when the
// reference's norm guard fires (cost 0, gradient untouched).
template <int P>
__device__ __attribute__((noinline)) double eval_cost(const Params* gp_in, unsigned lds_in)
{
    CParams* gp = (CParams*)uni_ptr(gp_in);
    CParams& prm = *gp;
    CConfig& c = prm.cfg;
    LDSQ Lds<P>& L = lds<P>(uni((int)lds_in));
    LDSQ EvalCtx& e = L.e;
    const int lane = threadIdx.x, M = uni(e.M), n = uni(e.n), NN = M * NS;
    const double xvI = c.standard_diff ? 0.0 : c.icr_xv;
    const SmoothL1 sl1(c.smooth_eps);
    const double map_inv = 1.0 / prm.map.res;
    __syncthreads();
    BE_STAMP(0)
    if (lane == 0) ++e.evals;
    // ---- norm guard (optimizer.cpp:635-636: `inf` is the macro 1 >> 30 = 0)
    double part = 0.0;
    for (int v = lane; v < n; v += 64) part += L.x[v] * L.x[v];
    if (sqrt(uni(wave_sum(part))) > 1e4) return 0.0; // cost 0, gradient untouched

    // ---- durations, knot positions
    double tpart = 0.0;
    for (int i = lane; i < M; i += 64) {
        const double T = t_of_tau(L.x[2 * (M - 1) + 1 + i]);
        L.T[i] = T;
        L.iT[i] = 1.0 / T;
        tpart += T;
    }
    const double sumT = uni(wave_sum(tpart));
    for (int k = lane; k <= M; k += 64)
        for (int d = 0; d < 2; ++d) {
            double p;
            if (k == 0) p = e.head[d][0];
            else if (k == M) p = (d == 1) ? L.x[2 * (M - 1)] : e.tail[d][0];
            else p = L.x[2 * (k - 1) + d];
            L.kp[d][k] = p;
            if (k == 0) { L.kv[d][0] = e.head[d][1]; L.ka[d][0] = e.head[d][2]; }
            if (k == M) { L.kv[d][M] = e.tail[d][1]; L.ka[d][M] = e.tail[d][2]; }
        }
    __syncthreads();
    BE_STAMP(1)
    // ---- knot system: right-hand sides (lane = (knot, dim)), factorisation + solves (lanes 0, 1)
    for (int t = lane; t < 2 * (M - 1); t += 64) {
        const int k = 1 + (t >> 1), d = t & 1;
        const minco::InvT l(L.T[k - 1], L.iT[k - 1]), r(L.T[k], L.iT[k]);
        double rr[2];
        minco::knot_rhs(l, r, L.kp[d][k - 1], L.kp[d][k], L.kp[d][k + 1], rr);
        if (k == 1) {
            const minco::Mat2 u = minco::knot_upper(l);
            rr[0] -= u.a * L.kv[d][0] + u.c * L.ka[d][0];
            rr[1] -= u.b * L.kv[d][0] + u.d * L.ka[d][0];
        }
        if (k == M - 1) {
            const minco::Mat2 u = minco::knot_upper(r);
            rr[0] -= u.a * L.kv[d][M] + u.b * L.ka[d][M];
            rr[1] -= u.c * L.kv[d][M] + u.d * L.ka[d][M];
        }
        L.y[d][k - 1][0] = rr[0];
        L.y[d][k - 1][1] = rr[1];
    }
    __syncthreads();
    BE_STAMP(2)
    // the lane's elimination factors leave through global memory (4.6 KB per problem, read back once for the adjoint solve below:
    // kept in LDS they cost a workgroup per CU, kept in registers across the passes in between a wavefront per SIMD, and
    // eliminating twice costs 9 % more instructions per plan)
    constexpr int PCR_PL = P <= 16 ? 16 : 32, PCR_ST = P <= 16 ? 4 : 5;
    GLBQ double* pcr_ws = (GLBQ double*)uni_ptr(prm.pcr) + (size_t)uni(e.prob) * WS_DOUBLES;
    GLBQ double* eb_ws = pcr_ws + WS_EB;       // [even node][2]
    GLBQ double* nt_ws = pcr_ws + WS_NODET;    // [node]
    GLBQ double* cs_ws = pcr_ws + WS_CS;       // [node][cos, sin]
    {
        PcrFactors pcr;
        knot_pcr<P, false>(L, M, L.T, L.y[0], L.y[1], pcr);
        if (lane < PCR_PL) {
#pragma unroll
            for (int st = 0; st < PCR_ST; ++st)
#pragma unroll
                for (int k = 0; k < 8; ++k) pcr_ws[(st * PCR_PL + lane) * 8 + k] = pcr.F[st][k];
#pragma unroll
            for (int k = 0; k < 4; ++k) pcr_ws[PCR_ST * PCR_PL * 8 + lane * 4 + k] = pcr.D[k];
        }
    }
    __syncthreads();
    for (int t = lane; t < 2 * (M - 1); t += 64) { // lane = (knot, dim): one round (was a walk over the knots on two lanes)
        const int k = 1 + (t >> 1), d = t & 1;
        L.kv[d][k] = L.y[d][k - 1][0];
        L.ka[d][k] = L.y[d][k - 1][1];
    }
    __syncthreads();
    // ---- coefficients, energy and its partial gradients: lane = (piece, dim)
    double epart = 0.0;
    for (int t = lane; t < 2 * M; t += 64) {
        const int i = t >> 1, d = t & 1;
        const double T = L.T[i];
        const minco::InvT q(T, L.iT[i]);
        double cc[6];
        minco::hermite(T, q, L.kp[d][i], L.kv[d][i], L.ka[d][i], L.kp[d][i + 1], L.kv[d][i + 1], L.ka[d][i + 1], cc, nullptr);
        for (int k = 0; k < 6; ++k) L.coef[(6 * i + k) * 2 + d] = cc[k];
        const double w = c.energy_w[d], t2 = T * T, t3 = t2 * T, t4 = t2 * t2, t5 = t4 * T;
        const double c3 = cc[3], c4 = cc[4], c5 = cc[5];
        epart += w * (36.0 * c3 * c3 * T + 144.0 * c4 * c3 * t2 + 192.0 * c4 * c4 * t3 + 240.0 * c5 * c3 * t3 + 720.0 * c5 * c4 * t4 +
                      720.0 * c5 * c5 * t5);
        L.gdC[(6 * i + 0) * 2 + d] = 0.0; L.gdC[(6 * i + 1) * 2 + d] = 0.0; L.gdC[(6 * i + 2) * 2 + d] = 0.0;
        L.gdC[(6 * i + 3) * 2 + d] = w * (72.0 * c3 * T + 144.0 * c4 * t2 + 240.0 * c5 * t3);
        L.gdC[(6 * i + 4) * 2 + d] = w * (144.0 * c3 * t2 + 384.0 * c4 * t3 + 720.0 * c5 * t4);
        L.gdC[(6 * i + 5) * 2 + d] = w * (240.0 * c3 * t3 + 720.0 * c4 * t4 + 1440.0 * c5 * t5);
        double gt = w * (36.0 * c3 * c3 + 288.0 * c4 * c3 * T + 576.0 * c4 * c4 * t2 + 720.0 * c5 * c3 * t2 + 2880.0 * c5 * c4 * t3 +
                         3600.0 * c5 * c5 * t4);
        gt += dpp_f64<0xB1, 0xF>(0.0, gt); // quad_perm [1, 0, 3, 2]: // the two dimensions of a piece sit on neighbouring lanes
        if (d == 0) L.gdT[i] = gt;
    }
    double cost_part = epart;
    __syncthreads();
    BE_STAMP(5)

    // ---- pass A over the nodes: flat state, Simpson integrands, pose-independent penalties
    const double w_mom = e.stage == 1 ? c.p_moment : c.w_moment, w_acc = e.stage == 1 ? c.p_acc : c.w_acc,
                 w_dom = e.stage == 1 ? c.p_domega : c.w_domega;
    for (int node = lane; node < NN; node += 64) {
        const int i = node / NS, j = node - i * NS;
        const double T = L.T[i], step = T / RES, t = j * (step / 2.0);
        const LDSQ double* ci = L.coef + 12 * i;
        double sg[2], d1[2], d2[2], d3[2];
        for (int d = 0; d < 2; ++d) {
            const double c0 = ci[d], c1 = ci[2 + d], c2 = ci[4 + d], c3 = ci[6 + d], c4 = ci[8 + d], c5 = ci[10 + d];
            sg[d] = ((((c5 * t + c4) * t + c3) * t + c2) * t + c1) * t + c0;
            d1[d] = (((5.0 * c5 * t + 4.0 * c4) * t + 3.0 * c3) * t + 2.0 * c2) * t + c1;
            d2[d] = ((20.0 * c5 * t + 12.0 * c4) * t + 6.0 * c3) * t + 2.0 * c2;
            d3[d] = (60.0 * c5 * t + 24.0 * c4) * t + 6.0 * c3;
        }
        double sy, cy;
        sincos(sg[0], &sy, &cy);
        cs_ws[2 * node] = cy;
        cs_ws[2 * node + 1] = sy;
        L.fx[node] = d1[1] * cy + d1[0] * xvI * sy;
        L.fy[node] = d1[1] * sy - d1[0] * xvI * cy;
        double gb[3][2] = {{0, 0}, {0, 0}, {0, 0}}, gT = 0.0;
        if ((j & 1) == 0) {
            const double alpha = (double)(j >> 1) / RES, omg = (j == 0 || j == NS - 1) ? 0.5 : 1.0, ws = omg * step;
            double f, df, v;
#define PENALISE(viol, weight, dviol_dt, apply)                           \
    if ((v = (viol)) > 0.0) {                                             \
        smoothed_l1(sl1, v, f, df);                                       \
        apply;                                                            \
        gT += omg * (weight) * (df * (dviol_dt) * step + f / RES);        \
        cost_part += ws * (weight) * f;                                   \
    }
            PENALISE(d2[1] * d2[1] - c.max_acc * c.max_acc, w_acc, 2.0 * alpha * d2[1] * d3[1], gb[2][1] += ws * w_acc * df * 2.0 * d2[1]);
            PENALISE(d2[0] * d2[0] - c.max_domega * c.max_domega, w_dom, 2.0 * alpha * d2[0] * d3[0],
                     gb[2][0] += ws * w_dom * df * 2.0 * d2[0]);
            if (e.stage == 2 && c.direct_v_omega) {
                PENALISE(d1[1] * d1[1] - c.max_vel * c.max_vel, w_mom, 2.0 * alpha * d1[1] * d2[1], gb[1][1] += ws * w_mom * df * 2.0 * d1[1]);
                PENALISE(d1[0] * d1[0] - c.max_omega * c.max_omega, w_mom, 2.0 * alpha * d1[0] * d2[0],
                         gb[1][0] += ws * w_mom * df * 2.0 * d1[0]);
            } else {
                for (int sym = -1; sym <= 1; sym += 2)
                    PENALISE(sym * c.max_vel * d1[0] + c.max_omega * d1[1] - c.max_vel * c.max_omega, w_mom,
                             alpha * (sym * c.max_vel * d2[0] + c.max_omega * d2[1]),
                             (gb[1][0] += ws * w_mom * df * sym * c.max_vel, gb[1][1] += ws * w_mom * df * c.max_omega));
                for (int sym = -1; sym <= 1; sym += 2)
                    PENALISE(sym * -c.min_vel * d1[0] - c.max_omega * d1[1] + c.min_vel * c.max_omega, w_mom,
                             alpha * (sym * -c.min_vel * d2[0] - c.max_omega * d2[1]),
                             (gb[1][0] += ws * w_mom * df * sym * -c.min_vel, gb[1][1] -= ws * w_mom * df * c.max_omega));
            }
            if (e.stage == 2)
                PENALISE(d1[0] * d1[0] * d1[1] * d1[1] - c.max_cen_acc * c.max_cen_acc, c.w_cen_acc,
                         2.0 * alpha * (d1[0] * d1[1] * d1[1] * d2[0] + d1[1] * d1[0] * d1[0] * d2[1]),
                         (gb[1][0] += ws * c.w_cen_acc * df * (2 * d1[0] * d1[1] * d1[1]),
                          gb[1][1] += ws * c.w_cen_acc * df * (2 * d1[0] * d1[0] * d1[1])));
        }
        LDSQ double* E = L.Ea + node * 3;
        E[0] = gb[0][0]; E[1] = gb[1][0]; E[2] = gb[1][1]; // gb[0][1] stays zero: nothing penalises s itself
        if ((j & 1) == 0) {
            GLBQ double* E2 = eb_ws + (i * (RES + 1) + (j >> 1)) * 2;
            E2[0] = gb[2][0]; E2[1] = gb[2][1];
        }
        nt_ws[node] = gT;
    }
    __syncthreads();
    BE_STAMP(6)

    // ---- Simpson panels and the running pose: prefix scan over the 8 M panels
    {
        double carry_x = e.start_xy[0], carry_y = e.start_xy[1];
        if (lane == 0) { L.posx[0] = carry_x; L.posy[0] = carry_y; }
        const int NP = RES * M;
        for (int base = 0; base < NP; base += 64) {
            const int p = base + lane;
            double ix = 0.0, iy = 0.0;
            if (p < NP) {
                const int i = p / RES, q = p - i * RES, n0 = i * NS + 2 * q;
                const double cint = div_res6(L.T[i]);
                // the reference accumulates  cint f0 + 4 cint f1 + cint f2  into a zeroed sum, in this order
                ix = cint * L.fx[n0]; ix += 4.0 * cint * L.fx[n0 + 1]; ix += cint * L.fx[n0 + 2];
                iy = cint * L.fy[n0]; iy += 4.0 * cint * L.fy[n0 + 1]; iy += cint * L.fy[n0 + 2];
            }
            ix = wave_prefix(ix); iy = wave_prefix(iy);
            if (p < NP) { L.posx[p + 1] = carry_x + ix; L.posy[p + 1] = carry_y + iy; }
            carry_x += lane63(ix);
            carry_y += lane63(iy);
        }
    }
    __syncthreads();
    BE_STAMP(7)
    if (lane == 0) {
        e.xy_err[0] = L.posx[RES * M] - e.final_xy[0];
        e.xy_err[1] = L.posy[RES * M] - e.final_xy[1];
    }
    __syncthreads(); // fx / fy are consumed; they now carry the position gradients of the chain rule
    for (int node = lane; node < NN; node += 64) { L.fx[node] = 0.0; L.fy[node] = 0.0; }
    __syncthreads();
    BE_STAMP(8)

    // ---- pose-dependent terms
    if (e.stage == 2) { // obstacle clearance at the even nodes
        for (int en = lane; en < (RES + 1) * M; en += 64) {
            const int i = en / (RES + 1), je = en - i * (RES + 1), j = 2 * je, node = i * NS + j;
            const double T = L.T[i], step = T / RES, alpha = (double)je / RES, omg = (j == 0 || j == NS - 1) ? 0.5 : 1.0, ws = omg * step;
            const double px = L.posx[RES * i + je], py = L.posy[RES * i + je], cy = cs_ws[2 * node], sy = cs_ws[2 * node + 1];
            const LDSQ double* ci = L.coef + 12 * i;
            const double t = j * (step / 2.0);
            const double d1th = (((5.0 * ci[10] * t + 4.0 * ci[8]) * t + 3.0 * ci[6]) * t + 2.0 * ci[4]) * t + ci[2];
            double gpx = 0.0, gpy = 0.0, gb0 = 0.0, gT = 0.0;
            // two check points at a time: both queries' loads are in flight before the first is interpolated; accumulated in order
            auto account = [&](const EsdfCell& cell, double bx, double by) {
                double gx, gy;
                const double sd = esdf_eval(cell, map_inv, true, e.safe_dis, gx, gy);
                const double viol = e.safe_dis - sd;
                if (viol > 0.0) {
                    double f, df;
                    smoothed_l1(sl1, viol, f, df);
                    const double rot = gx * (-sy * bx - cy * by) + gy * (cy * bx - sy * by);
                    gpx -= ws * c.w_collision * df * gx;
                    gpy -= ws * c.w_collision * df * gy;
                    gb0 -= ws * c.w_collision * df * rot;
                    gT += omg * c.w_collision * (df * (-alpha * d1th * rot) * step + f / RES);
                    cost_part += ws * c.w_collision * f;
                }
            };
            const int nq = c.n_check;
            for (int q = 0; q < nq; q += 2) {
                const bool two = q + 1 < nq;
                const double bx0 = c.check_pts[q][0], by0 = c.check_pts[q][1];
                const double bx1 = c.check_pts[two ? q + 1 : q][0], by1 = c.check_pts[two ? q + 1 : q][1];
                const EsdfCell c0 = esdf_fetch(prm.map, map_inv, px + cy * bx0 - sy * by0, py + sy * bx0 + cy * by0);
                const EsdfCell c1 = esdf_fetch(prm.map, map_inv, px + cy * bx1 - sy * by1, py + sy * bx1 + cy * by1);
                account(c0, bx0, by0);
                if (two) account(c1, bx1, by1);
            }
            L.fx[node] = gpx;
            L.fy[node] = gpy;
            L.Ea[node * 3 + 0] += gb0;
            nt_ws[node] += gT;
        }
    } else { // way-point attraction at the end of every piece
        const GLBQ double* way = (const GLBQ double*)e.positions;
        for (int i = lane; i < M; i += 64) {
            const double ex = L.posx[RES * (i + 1)] - way[2 * i], ey = L.posy[RES * (i + 1)] - way[2 * i + 1];
            cost_part += c.p_bigpath * (ex * ex + ey * ey);
            L.fx[i * NS + NS - 1] = c.p_bigpath * 2.0 * ex;
            L.fy[i * NS + NS - 1] = c.p_bigpath * 2.0 * ey;
        }
    }
    __syncthreads();
    BE_STAMP(9)
    // ---- chain coefficients: inclusive suffix sums over the nodes (+ the ALM terminal term)
    {
        double add_x = 0.0, add_y = 0.0;
        if (e.stage == 2) {
            const double ax = e.xy_err[0] + e.lam[0] / e.rho[0], ay = e.xy_err[1] + e.lam[1] / e.rho[1];
            if (lane == 0) cost_part += 0.5 * (e.rho[0] * ax * ax + e.rho[1] * ay * ay);
            add_x = e.rho[0] * ax;
            add_y = e.rho[1] * ay;
        }
        double carry_x = 0.0, carry_y = 0.0;
        const int rounds = (NN + 63) / 64;
        // the nodes of a round sit on the lanes in REVERSE order (lane l holds node 64 r + 63 - l): the suffix sum over the nodes is
        // the prefix scan over the lanes -- the same additions in the same order as reversing the lanes around the scan, without
        // the four LDS permutes per scan that the reversals were
        for (int r = rounds - 1; r >= 0; --r) {
            const int node = r * 64 + 63 - lane;
            double vx = node < NN ? L.fx[node] : 0.0, vy = node < NN ? L.fy[node] : 0.0;
            vx = wave_prefix(vx); vy = wave_prefix(vy);
            if (node < NN) { L.fx[node] = vx + carry_x + add_x; L.fy[node] = vy + carry_y + add_y; }
            carry_x += lane63(vx);
            carry_y += lane63(vy);
        }
    }
    __syncthreads();
    BE_STAMP(10)
    // ---- pass C: chain rule through the Simpson sums into the node terms
    for (int node = lane; node < NN; node += 64) {
        const int i = node / NS, j = node - i * NS;
        const double T = L.T[i], step = T / RES, t = j * (step / 2.0), cint = div_res6(T), ialpha = (double)j / (2 * RES);
        const double sw = (j == 0 || j == NS - 1) ? 1.0 : ((j & 1) ? 4.0 : 2.0);
        const double cx = L.fx[node] * sw, cyy = L.fy[node] * sw, cy = cs_ws[2 * node], sy = cs_ws[2 * node + 1];
        const LDSQ double* ci = L.coef + 12 * i;
        double d1[2], d2[2];
        for (int d = 0; d < 2; ++d) {
            const double c1 = ci[2 + d], c2 = ci[4 + d], c3 = ci[6 + d], c4 = ci[8 + d], c5 = ci[10 + d];
            d1[d] = (((5.0 * c5 * t + 4.0 * c4) * t + 3.0 * c3) * t + 2.0 * c2) * t + c1;
            d2[d] = ((20.0 * c5 * t + 12.0 * c4) * t + 6.0 * c3) * t + 2.0 * c2;
        }
        const double fxv = d1[1] * cy + d1[0] * xvI * sy, fyv = d1[1] * sy - d1[0] * xvI * cy;
        // d(dx)/d(theta coefficients) = b0 (-s' sin + th' xv cos) + b1 xv sin ; d(dy)/.. = b0 (s' cos - th' xv sin) - b1 xv cos
        // (the reference's sign of the th' xv sin term, optimizer.cpp:822; exact would be +)
        LDSQ double* E = L.Ea + node * 3;
        E[0] += cint * ((-d1[1] * sy + d1[0] * xvI * cy) * cx + (d1[1] * cy - d1[0] * xvI * sy) * cyy);
        E[1] += cint * xvI * (sy * cx - cy * cyy);
        E[2] += cint * (cy * cx + sy * cyy);
        const double XT = (d2[1] * cy - d1[1] * d1[0] * sy + d2[0] * xvI * sy + d1[0] * d1[0] * xvI * cy) * ialpha * cint + div_res6(fxv);
        const double YT = (d2[1] * sy + d1[1] * d1[0] * cy - d2[0] * xvI * cy + d1[0] * d1[0] * xvI * sy) * ialpha * cint + div_res6(fyv);
        nt_ws[node] += XT * cx + YT * cyy;
    }
    __syncthreads();
    BE_STAMP(11)
    // ---- node terms -> coefficient gradient: lane = (piece, power, dim); time gradient: lane = piece
    // One lane per (piece, dimension) carries the six accumulators of its powers: the running powers 1, t, t^2, ... of a node
    // are formed once (the same products in the same order as a power loop per exponent) and serve all six, the three node
    // terms are read once -- a fifth of the instructions of one lane per (piece, power, dimension), in one round.
    for (int t = lane; t < 2 * M; t += 64) {
        const int i = t >> 1, d = t & 1;
        const double half = L.T[i] / (2 * RES);
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = 0; j < NS; ++j) {
            const LDSQ double* E = L.Ea + (i * NS + j) * 3;
            const double tt = j * half, E0 = (d == 0) ? E[0] : 0.0, E1 = E[1 + d],
                         E2 = ((j & 1) == 0) ? eb_ws[(i * (RES + 1) + (j >> 1)) * 2 + d] : 0.0;
            double pw[6];
            pw[0] = 1.0;
#pragma unroll
            for (int k = 1; k <= 5; ++k) pw[k] = pw[k - 1] * tt;
#pragma unroll
            for (int q = 0; q <= 5; ++q) {
                const double p0 = pw[q], p1 = q >= 1 ? q * pw[q >= 1 ? q - 1 : 0] : 0.0, p2 = q >= 2 ? q * (q - 1) * pw[q >= 2 ? q - 2 : 0] : 0.0;
                acc[q] += p0 * E0 + p1 * E1 + p2 * E2;
            }
        }
#pragma unroll
        for (int q = 0; q <= 5; ++q) L.gdC[(6 * i + q) * 2 + d] += acc[q];
    }
    for (int i = lane; i < M; i += 64) {
        double acc = 0.0;
        for (int j = 0; j < NS; ++j) acc += nt_ws[i * NS + j];
        L.gdT[i] += acc;
    }
    __syncthreads();
    BE_STAMP(12)
    // ---- adjoint of the spline: coefficient gradient -> knot-state gradient -> knot system -> variables
    // phase 1: every (piece, dim) lane turns its coefficient gradient into the gradients of its two knot states;
    // the start knot is written directly, the end knot goes through a staging slot (Ea is free by now)
    for (int t = lane; t < 2 * M; t += 64) {
        const int i = t >> 1, d = t & 1;
        const minco::InvT q(L.T[i], L.iT[i]);
        double G[6], g0[3], g1[3];
        for (int k = 0; k < 6; ++k) G[k] = L.gdC[(6 * i + k) * 2 + d];
        minco::hermite_adjoint(q, G, g0, g1);
        for (int k = 0; k < 3; ++k) { L.gk[d][i][k] = g0[k]; L.Ea[t * 3 + k] = g1[k]; }
        if (i == M - 1) for (int k = 0; k < 3; ++k) L.gk[d][M][k] = 0.0;
    }
    __syncthreads();
    BE_STAMP(13)
    for (int t = lane; t < 2 * M; t += 64) {
        const int i = t >> 1, d = t & 1;
        for (int k = 0; k < 3; ++k) L.gk[d][i + 1][k] += L.Ea[t * 3 + k];
    }
    __syncthreads();
    BE_STAMP(14)
    for (int t = lane; t < 2 * (M - 1); t += 64) {
        const int k = 1 + (t >> 1), d = t & 1;
        L.y[d][k - 1][0] = L.gk[d][k][1];
        L.y[d][k - 1][1] = L.gk[d][k][2];
    }
    __syncthreads();
    BE_STAMP(15)
    { // K is symmetric: the same system, replayed with the factors of the spline solve (read back from the problem's workspace)
        PcrFactors pcr;
#pragma unroll
        for (int st = 0; st < 5; ++st)
#pragma unroll
            for (int k = 0; k < 8; ++k) pcr.F[st][k] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) pcr.D[k] = 0.0;
        if (lane < PCR_PL) {
#pragma unroll
            for (int st = 0; st < PCR_ST; ++st)
#pragma unroll
                for (int k = 0; k < 8; ++k) pcr.F[st][k] = pcr_ws[(st * PCR_PL + lane) * 8 + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) pcr.D[k] = pcr_ws[PCR_ST * PCR_PL * 8 + lane * 4 + k];
        }
        knot_pcr<P, true>(L, M, L.T, L.y[0], L.y[1], pcr);
    }
    __syncthreads();
    BE_STAMP(16)
    // way-point and tail gradients: lane = (knot 1..M, dim)
    for (int t = lane; t < 2 * M; t += 64) {
        const int k = 1 + (t >> 1), d = t & 1;
        double val = L.gk[d][k][0];
        // - sum_k' mu_k' . dF_k'/dp_k ; dF_k'/dp: from knot k'+... (see minco_spline.h knot_rhs)
        if (k + 1 <= M - 1) { // F_{k+1} depends on p_k as its left neighbour: dF/dp = (360 l4, -60 l3), l = T_k
            const minco::InvT l(L.T[k], L.iT[k]);
            val -= L.y[d][k][0] * (360.0 * l.i4) + L.y[d][k][1] * (-60.0 * l.i3);
        }
        if (k <= M - 1) { // F_k on its own position: -(360 l4 - 360 r4... ) see below
            const minco::InvT l(L.T[k - 1], L.iT[k - 1]), r(L.T[k], L.iT[k]);
            val -= L.y[d][k - 1][0] * (-360.0 * l.i4 + 360.0 * r.i4) + L.y[d][k - 1][1] * (60.0 * l.i3 + 60.0 * r.i3);
        }
        if (k - 1 >= 1) { // F_{k-1} depends on p_k as its right neighbour: dF/dp = (-360 r4, -60 r3), r = T_{k-1}
            const minco::InvT r(L.T[k - 1], L.iT[k - 1]);
            val -= L.y[d][k - 2][0] * (-360.0 * r.i4) + L.y[d][k - 2][1] * (-60.0 * r.i3);
        }
        if (k <= M - 1) L.g[2 * (k - 1) + d] = val;
        else if (d == 1) L.g[2 * (M - 1)] = val;
    }
    // duration gradients: lane = (piece, dim)
    for (int t = lane; t < 2 * M; t += 64) {
        const int i = t >> 1, d = t & 1;
        const double T = L.T[i];
        const minco::InvT q(T, L.iT[i]);
        double cc[6], dc[6], dE0[2], dE1[2];
        minco::hermite(T, q, L.kp[d][i], L.kv[d][i], L.ka[d][i], L.kp[d][i + 1], L.kv[d][i + 1], L.ka[d][i + 1], cc, dc);
        minco::piece_end_rows(T, cc, dc, dE0, dE1);
        double val = 0.0;
        for (int k = 3; k < 6; ++k) val += L.gdC[(6 * i + k) * 2 + d] * dc[k];
        if (i + 1 <= M - 1) val -= L.y[d][i][0] * dE1[0] + L.y[d][i][1] * dE1[1];
        if (i >= 1) val += L.y[d][i - 1][0] * dE0[0] + L.y[d][i - 1][1] * dE0[1];
        val += dpp_f64<0xB1, 0xF>(0.0, val); // quad_perm [1, 0, 3, 2]
        if (d == 0) {
            const double tau = L.x[2 * (M - 1) + 1 + i];
            L.g[2 * (M - 1) + 1 + i] = (L.gdT[i] + val + e.time_weight) * dt_dtau(tau);
        }
    }
    __syncthreads();
    BE_STAMP(17)
    double cost = uni(wave_sum(cost_part));
    cost += (e.stage == 1 ? c.p_time : e.time_weight) * sumT;
    return cost;
}

// ---- L-BFGS -----------------------------------------------------------------------------------------
enum { LB_CONVERGENCE = 0, LB_STOP = 1, LBE_INVALID_FUNCVAL = -1012, LBE_MINIMUMSTEP = -1011, LBE_MAXIMUMSTEP = -1010,
       LBE_MAXIMUMLINESEARCH = -1009, LBE_MAXIMUMITERATION = -1008, LBE_WIDTHTOOSMALL = -1007, LBE_INVALIDPARAMETERS = -1006,
       LBE_INCREASEGRADIENT = -1005 };

template <int P>
__device__ double vdot(const LDSQ double* a, const LDSQ double* b, int n)
{
    double s = 0.0;
    for (int v = threadIdx.x; v < n; v += 64) s += a[v] * b[v];
    return uni(wave_sum(s));
}
template <int P>
__device__ double vmaxabs(const LDSQ double* a, int n)
{
    double s = 0.0;
    for (int v = threadIdx.x; v < n; v += 64) s = fmax(s, fabs(a[v]));
    return uni(wave_max(s));
}

// a / b with r = 1 / b (correctly rounded) known in advance: two Newton corrections of a * r through exact residuals
// (Markstein's division: the result is the correctly rounded quotient) -- five dependent multiply-adds on the chain of the
// two-loop recursion instead of the ~12 dependent instructions of the division macro (v_div_scale, v_rcp_f64, ...).
__device__ __forceinline__ double div_by_known(double a, double b, double r)
{
    const double q0 = a * r;
    const double q1 = fma(fma(-b, q0, a), r, q0);
    return fma(fma(-b, q1, a), r, q1);
}

// CH (s, y) pairs of the L-BFGS history in registers, as loaded: lane v holds entries v, v + 64, ... (clamped to the slab's
// last slot, nstride - 1, where y's and its reciprocal ride: lane 63 of the last round has them).  Pairs first, first + dir,
// ... (mod m).  EVERY load is issued whatever the number of pairs wanted and n are, and nothing here looks at a loaded value:
// with loads under conditions the compiler cannot count the outstanding ones and waits for all of them -- the next chunk's
// included -- before the first use; the consumer masks (pair_values).
template <int NVL, int CH>
struct PairChunk {
    double s[CH][NVL], y[CH][NVL];
};
template <int NVL, int CH>
__device__ __forceinline__ void load_pairs(PairChunk<NVL, CH>& c, const GLBQ double* hist, int nstride, int m, int first, int dir, int lane)
{
#pragma unroll
    for (int cidx = 0; cidx < CH; ++cidx) {
        int jj = first + dir * cidx;
        if (m >= CH) { jj += jj < 0 ? m : 0; jj -= jj >= m ? m : 0; }
        else jj = ((jj % m) + m) % m;
        const GLBQ double* sj = hist + (size_t)jj * 2 * nstride;
#pragma unroll
        for (int r = 0; r < NVL; ++r) {
            const int at = min(lane + 64 * r, nstride - 1);
            c.s[cidx][r] = sj[at];
            c.y[cidx][r] = sj[nstride + at];
        }
    }
}
// pair cidx of a chunk: the vectors with the lanes beyond n zeroed, y's and 1 / y's
template <int NVL, int CH>
__device__ __forceinline__ void pair_values(const PairChunk<NVL, CH>& c, int cidx, int n, int lane, double (&sv)[NVL], double (&yv)[NVL], double& ys,
                                            double& rys)
{
#pragma unroll
    for (int r = 0; r < NVL; ++r) {
        const bool in = lane + 64 * r < n;
        sv[r] = in ? c.s[cidx][r] : 0.0;
        yv[r] = in ? c.y[cidx][r] : 0.0;
    }
    ys = lane63(c.s[cidx][NVL - 1]);
    rys = lane63(c.y[cidx][NVL - 1]);
}


// ---- two-loop recursion on aligned chunks of 8 pairs with their cross products at hand ------------------------------------
// The pair-at-a-time recursion is one chain: every pair pays a wavefront sum, a division and an update before the next may
// start (~260 cycles for a lone wavefront, tools/micro/wave_sum_f64.hip), and a plan with 50 pairs in its history walks it a
// hundred times per iteration.  Inside an aligned chunk of 8 slots the chain shortens to scalar work when the cross products
// G[k][l] = s_k . y_l (k < l) are known -- and they are properties of the history, computed once when pair l arrives:
//   first loop  (l descending):  a_l = (s_l . q_top - sum_{j > l} a_j G[l][j]) / y's_l,        q -= a_l y_l
//   second loop (l ascending):   b_l = (y_l . r_base + sum_{j < l} (a_j - b_j) G[j][l]) / y's_l,  r += (a_l - b_l) s_l
// (q_top / r_base: the vector as the chunk finds it).  The 8 products of a chunk against that vector are independent and are
// summed TOGETHER (sum8: a transposing butterfly, 64 instructions for 8 wavefront sums instead of 8 x 19); the recurrence runs
// lane-parallel (lane k carries row k of G and its running numerator: a step is the division in every lane, one v_readlane
// for the pair whose turn it is, one multiply-add on the numerators and one on the vector).  Same algebra as lbfgs.hpp:704-735,
// sums associate differently (float64 rounding).  Used while the history has not wrapped (slots 0 .. bound - 1, which is every
// iteration of a run of up to mem_size iterations); after that the pair-at-a-time form below takes over.
__device__ __forceinline__ double xor4_f64(double x)
{
    const int xl = __double2loint(x), xh = __double2hiint(x);
    int lo = __builtin_amdgcn_update_dpp(0, xl, 0x104, 0xF, 0x5, false); // row_shl:4 into banks 0, 2: lane i <- i + 4
    int hi = __builtin_amdgcn_update_dpp(0, xh, 0x104, 0xF, 0x5, false);
    lo = __builtin_amdgcn_update_dpp(lo, xl, 0x114, 0xF, 0xA, false);   // row_shr:4 into banks 1, 3: lane i <- i - 4
    hi = __builtin_amdgcn_update_dpp(hi, xh, 0x114, 0xF, 0xA, false);
    return __hiloint2double(hi, lo);
}
// sums over the wavefront of 8 vectors at once: lane i returns the total of v[i & 7]
__device__ __forceinline__ double sum8(const double (&v)[8], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    double w[4], x[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { // lanes i, i ^ 1: the even lane keeps vector 2 i, the odd one vector 2 i + 1
        const double keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        w[i] = keep + dpp_f64_nofill<0xB1>(send); // quad_perm [1, 0, 3, 2]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b1 ? w[2 * i + 1] : w[2 * i], send = b1 ? w[2 * i] : w[2 * i + 1];
        x[i] = keep + dpp_f64_nofill<0x4E>(send); // quad_perm [2, 3, 0, 1]
    }
    const double keep = b2 ? x[1] : x[0], send = b2 ? x[0] : x[1];
    double y = keep + xor4_f64(send);   // lane i: vector i & 7, summed over its group of 8 lanes
    y += dpp_f64_nofill<0x128>(y);      // row_ror:8: over the row of 16
    { // rows 0 + 2 and 1 + 3, then even + odd (v_permlane32_swap / v_permlane16_swap, gfx950)
        const auto l32 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(y), (unsigned)__double2loint(y), false, false);
        const auto h32 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(y), (unsigned)__double2hiint(y), false, false);
        y = __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
        const auto l16 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(y), (unsigned)__double2loint(y), false, false);
        const auto h16 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(y), (unsigned)__double2hiint(y), false, false);
        y = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
    }
    return y;
}
__device__ __forceinline__ double lane_of(double v, int l) // wave-uniform l
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// one aligned chunk in registers: lane v holds entry v of the 8 s and y vectors (slots 8 c .. 8 c + 7; entries beyond n are
// zeros / y's: finite, and they meet zeros of the other factor), lane k (mod 8) row k of G (first loop) or of its transpose
// (second loop) and y's, 1 / y's of slot 8 c + k
struct GramChunk {
    double s[8], y[8], g[8], ys, rys;
};
__device__ __forceinline__ void load_gram_chunk(GramChunk& c, const GLBQ double* hist, const GLBQ double* gram, int nstride, int chunk, int goff, int lane)
{
    const int at = min(lane, nstride - 1), k = lane & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const GLBQ double* sj = hist + (size_t)(8 * chunk + i) * 2 * nstride;
        c.s[i] = sj[at];
        c.y[i] = sj[nstride + at];
    }
    const GLBQ double* row = gram + goff + chunk * 64 + k * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) c.g[i] = row[i];
    c.ys = gram[GRAM_YS + 8 * chunk + k];
    c.rys = gram[GRAM_RYS + 8 * chunk + k];
}

// line_search_lewisoverton: x, g in L.x / L.g; s = L.d; xp, gp in L.xp / L.gp
template <int P>
__device__ __forceinline__ int line_search(const Params* gp, unsigned lbase, const LbfgsParam& pr, double& f, double& stp, double stpmin,
                                           double stpmax)
{
    LDSQ Lds<P>& L = lds<P>(lbase);
    const int n = uni(L.e.n), lane = threadIdx.x;
    int count = 0;
    bool brackt = false, touched = false;
    double mu = 0.0, nu = stpmax;
    if (!(stp > 0.0)) return LBE_INVALIDPARAMETERS;
    const double dginit = vdot<P>(L.gp, L.d, n);
    if (0.0 < dginit) return LBE_INCREASEGRADIENT;
    const double finit = f, dgtest = pr.f_dec_coeff * dginit, dstest = pr.s_curv_coeff * dginit;
    for (;;) {
        for (int v = lane; v < n; v += 64) L.x[v] = L.xp[v] + stp * L.d[v];
        __syncthreads();
        f = uni(eval_cost<P>(gp, lbase)); // a function's return value arrives in vector registers
        ++count;
        if (isinf(f) || isnan(f)) return LBE_INVALID_FUNCVAL;
        if (pr.past > 0 && fabs(finit - f) / (fabs(finit) + 1.0) < pr.delta / pr.past) return count;
        if (f > finit + stp * dgtest) {
            nu = stp;
            brackt = true;
        } else if (vdot<P>(L.g, L.d, n) < dstest) {
            mu = stp;
        } else {
            return count;
        }
        if (pr.max_linesearch <= count) return LBE_MAXIMUMLINESEARCH;
        if (brackt && (nu - mu) < pr.machine_prec * nu) return LBE_WIDTHTOOSMALL;
        stp = brackt ? 0.5 * (mu + nu) : stp * 2.0;
        if (stp < stpmin) return LBE_MINIMUMSTEP;
        if (stp > stpmax) {
            if (touched) return LBE_MAXIMUMSTEP;
            touched = true;
            stp = stpmax;
        }
    }
}

// lbfgs_optimize.  hist: [mem][2][nstride] doubles of this problem.  iter_cap > 0 limits the iterations.
template <int P>
__device__ __attribute__((noinline)) int lbfgs(const Params* gp_in, unsigned lds_in, const LbfgsParam pr_in, double* hist_in, int nstride_in,
                                               int iter_cap_in, double& f_out, int& k_out)
{
    const Params* gp = uni_ptr(gp_in);
    const unsigned lbase = uni((int)lds_in);
    const LbfgsParam pr = uni(pr_in);
    GLBQ double* hist = (GLBQ double*)uni_ptr(hist_in);
    const int nstride = uni(nstride_in), iter_cap = uni(iter_cap_in);
    LDSQ Lds<P>& L = lds<P>(lbase);
    const int n = uni(L.e.n), lane = threadIdx.x, m = pr.mem_size;
    int ret, k = 1, end = 0, bound = 0;
    double fx = uni(eval_cost<P>(gp, lbase));
    if (lane == 0) L.pf[0] = fx;
    for (int v = lane; v < n; v += 64) L.d[v] = -L.g[v];
    __syncthreads();
    double gn = vmaxabs<P>(L.g, n), xn = vmaxabs<P>(L.x, n);
    if (gn / fmax(1.0, xn) < pr.g_epsilon) {
        ret = LB_CONVERGENCE;
    } else {
        double step = 1.0 / sqrt(vdot<P>(L.d, L.d, n));
        for (;;) {
            for (int v = lane; v < n; v += 64) { L.xp[v] = L.x[v]; L.gp[v] = L.g[v]; }
            __syncthreads();
            const int ls = line_search<P>(gp, lbase, pr, fx, step, pr.min_step, pr.max_step);
            if (ls < 0) {
                for (int v = lane; v < n; v += 64) { L.x[v] = L.xp[v]; L.g[v] = L.gp[v]; }
                __syncthreads();
                ret = ls;
                break;
            }
            gn = vmaxabs<P>(L.g, n);
            xn = vmaxabs<P>(L.x, n);
            if (gn / fmax(1.0, xn) < pr.g_epsilon) { ret = LB_CONVERGENCE; break; }
            if (pr.past > 0) {
                if (pr.past <= k) {
                    const double rate = fabs(uni(L.pf[k % pr.past]) - fx) / fmax(1.0, fabs(fx));
                    if (rate < pr.delta) { ret = LB_STOP; break; }
                }
                __syncthreads();
                if (lane == 0) L.pf[k % pr.past] = fx;
                __syncthreads();
            }
            if ((pr.max_iterations != 0 && pr.max_iterations <= k) || (iter_cap > 0 && iter_cap <= k)) { ret = LBE_MAXIMUMITERATION; break; }
            ++k;
            GLBQ double* sk = hist + (size_t)end * 2 * nstride;
            GLBQ double* yk = sk + nstride;
            double pys = 0.0, pyy = 0.0, pss = 0.0, pgg = 0.0;
            for (int v = lane; v < n; v += 64) {
                const double s = L.x[v] - L.xp[v], y = L.g[v] - L.gp[v];
                sk[v] = s;
                yk[v] = y;
                pys += y * s; pyy += y * y; pss += s * s; pgg += L.gp[v] * L.gp[v];
                L.d[v] = -L.g[v];
            }
            const double ys = uni(wave_sum(pys)), yy = uni(wave_sum(pyy));
            const double cau = uni(wave_sum(pss)) * sqrt(uni(wave_sum(pgg))) * pr.cautious_factor;
            // y's and its reciprocal ride in the last slot of the pair's s / y slabs (n <= nstride - 1: the slot is free) and
            // come back with the chunk loads of the recursion
            // (the slots between n and the last one are zeroed: the chunked recursion multiplies whole slabs)
            for (int v = n + lane; v < nstride - 1; v += 64) { sk[v] = 0.0; yk[v] = 0.0; }
            if (lane == 0) {
                const double rys = 1.0 / ys;
                sk[nstride - 1] = ys; yk[nstride - 1] = rys;
                if (gp->gram) {
                    GLBQ double* gram = (GLBQ double*)gp->gram + (size_t)L.e.prob * GRAM_DOUBLES;
                    gram[GRAM_YS + end] = ys; gram[GRAM_RYS + end] = rys;
                }
            }
            __syncthreads();
            BE_STAMP(20)
            if (ys > cau) {
                const int slot_new = end;
                ++bound;
                if (bound > m) bound = m;
                end = (end + 1) % m;
                bool chunked = false;
                if constexpr (3 * P <= 64) chunked = (end != 0 && bound == end); // the history sits in slots 0 .. bound - 1
                if (chunked) {
                    GLBQ double* gram = (GLBQ double*)gp->gram + (size_t)uni(L.e.prob) * GRAM_DOUBLES;
                    const int top = (bound - 1) >> 3, hi_top = (bound - 1) & 7, pnew = slot_new & 7;
                    double dq = (lane < n) ? L.d[lane] : 0.0;
                    const double ynew = (lane < n) ? L.g[lane] - L.gp[lane] : 0.0;
                    GramChunk A, B;
                    // first loop, chunks top .. 0; the newest pair's column of G is formed on the way (its chunk is the first)
                    auto first = [&](GramChunk& c, int ch, auto full) {
                        constexpr bool FULL = decltype(full)::value;
                        double pr8[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) pr8[i] = c.s[i] * dq;
                        double D = sum8(pr8, lane);
                        if (!FULL) { // the top chunk: column pnew of G = s_k . y_new, k < pnew (lane k), stored and patched into the rows at hand
#pragma unroll
                            for (int i = 0; i < 8; ++i) pr8[i] = c.s[i] * ynew;
                            const double col = sum8(pr8, lane);
                            const int k = lane & 7;
                            if (lane < pnew) {
                                gram[GRAM_G + ch * 64 + k * 8 + pnew] = col;
                                gram[GRAM_GT + ch * 64 + pnew * 8 + k] = col;
                            }
#pragma unroll
                            for (int i = 0; i < 8; ++i) c.g[i] = (i == pnew) ? (k < pnew ? col : 0.0) : c.g[i];
                        }
#pragma unroll
                        for (int l = 7; l >= 0; --l) {
                            if (FULL || l <= hi_top) {
                                const double a = lane_of(div_by_known(D, c.ys, c.rys), l);
                                D = fma(-a, c.g[l], D);
                                dq = fma(-a, c.y[l], dq);
                            }
                        }
                        const double afin = div_by_known(D, c.ys, c.rys); // lane k: a of slot 8 ch + k (rows of G are zero up to the diagonal)
                        if (lane < 8) L.alpha[8 * ch + lane] = afin;
                        dq = (lane < n) ? dq : 0.0;
                    };
                    load_gram_chunk(A, hist, gram, nstride, top, GRAM_G, lane);
                    for (int ch = top;; ch -= 2) {
                        // (past chunk 0 the next thing needed is chunk 0 again, with the rows of the transpose: the second loop's first chunk)
                        load_gram_chunk(B, hist, gram, nstride, max(ch - 1, 0), ch >= 1 ? GRAM_G : GRAM_GT, lane);
                        if (ch == top) first(A, ch, std::false_type{}); else first(A, ch, std::true_type{});
                        if (ch == 0) break;
                        load_gram_chunk(A, hist, gram, nstride, max(ch - 2, 0), ch >= 2 ? GRAM_G : GRAM_GT, lane);
                        first(B, ch - 1, std::true_type{});
                        if (ch == 1) break;
                    }
                    dq *= ys / yy;
                    __syncthreads(); // alpha[] and the new column of the transpose are read back below
                    BE_STAMP(21)
                    if (gp->stamps && blockIdx.x == 0 && threadIdx.x == 0) { GLBQ long long* st_ = (GLBQ long long*)gp->stamps; st_[40] += 1; st_[41] += bound; }
                    auto second = [&](GramChunk& c, int ch, auto full) {
                        constexpr bool FULL = decltype(full)::value;
                        double pr8[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) pr8[i] = c.y[i] * dq;
                        double E = sum8(pr8, lane);
                        const double al = L.alpha[8 * ch + (lane & 7)];
#pragma unroll
                        for (int l = 0; l < 8; ++l) {
                            if (FULL || l <= hi_top) {
                                const double cf = lane_of(al - div_by_known(E, c.ys, c.rys), l);
                                E = fma(cf, c.g[l], E);
                                dq = fma(cf, c.s[l], dq);
                            }
                        }
                        dq = (lane < n) ? dq : 0.0;
                    };
                    // chunk 0 with the transposed rows is in flight since the end of the first loop -- in B when top is even, in A when
                    // it is odd -- unless chunk 0 is the top chunk itself, whose row of the transpose has just been written
                    if (top == 0) load_gram_chunk(A, hist, gram, nstride, 0, GRAM_GT, lane);
                    else if ((top & 1) == 0) A = B;
                    for (int ch = 0;; ch += 2) {
                        load_gram_chunk(B, hist, gram, nstride, min(ch + 1, top), GRAM_GT, lane);
                        if (ch == top) second(A, ch, std::false_type{}); else second(A, ch, std::true_type{});
                        if (ch == top) break;
                        load_gram_chunk(A, hist, gram, nstride, min(ch + 2, top), GRAM_GT, lane);
                        if (ch + 1 == top) second(B, ch + 1, std::false_type{}); else second(B, ch + 1, std::true_type{});
                        if (ch + 1 == top) break;
                    }
                    if (lane < n) L.d[lane] = dq;
                    BE_STAMP(22)
                } else {
                // Two-loop recursion (lbfgs.hpp:704-735).  d stays in registers (NVL values per lane); the (s, y)
                // pairs come from HBM in chunks of CH pairs whose loads are all in flight together, and the loads of
                // the NEXT chunk are issued before the current one is worked on (two register buffers): the memory
                // latency of a chunk hides under the dependent chain of the one before it.  Same operations in the
                // same order as the pair-at-a-time form.
                constexpr int NVL = (3 * P + 63) / 64, CH = 8;
                typedef PairChunk<NVL, CH> Chunk;
                double dreg[NVL];
#pragma unroll
                for (int r = 0; r < NVL; ++r) dreg[r] = (lane + 64 * r < n) ? L.d[lane + 64 * r] : 0.0;
                auto wrap = [m](int v) { while (v < 0) v += m; while (v >= m) v -= m; return v; };
                int j = end, left = bound;
                // first loop: pairs j - 1, j - 2, ... (newest first)
                // (a whole chunk runs as straight-line code: the preparation of a pair overlaps the chain of the one before)
                auto first_loop = [&](const Chunk& c, int cnt, auto full) {
#pragma unroll
                    for (int cidx = 0; cidx < CH; ++cidx) {
                        if (decltype(full)::value || cidx < cnt) {
                            j = (j == 0 ? m : j) - 1;
                            double sv[NVL], yv[NVL], ysj, rysj;
                            pair_values(c, cidx, n, lane, sv, yv, ysj, rysj);
                            double p = 0.0;
#pragma unroll
                            for (int r = 0; r < NVL; ++r) p += sv[r] * dreg[r];
                            const double a = div_by_known(uni(wave_sum(p)), ysj, rysj);
                            L.alpha[j] = a; // every lane, the same value
#pragma unroll
                            for (int r = 0; r < NVL; ++r) dreg[r] += (-a) * yv[r];
                        }
                    }
                };
                {
                    Chunk A, B;
                    int cntA = min(CH, left), cntB;
                    load_pairs(A, hist, nstride, m, wrap(j - 1), -1, lane);
                    for (;;) {
                        left -= cntA;
                        cntB = min(CH, left);
                        load_pairs(B, hist, nstride, m, wrap(j - 1 - cntA), -1, lane);
                        if (cntA == CH) first_loop(A, cntA, std::true_type{}); else first_loop(A, cntA, std::false_type{});
                        if (cntB == 0) break;
                        left -= cntB;
                        cntA = min(CH, left);
                        load_pairs(A, hist, nstride, m, wrap(j - 1 - cntB), -1, lane);
                        if (cntB == CH) first_loop(B, cntB, std::true_type{}); else first_loop(B, cntB, std::false_type{});
                        if (cntA == 0) break;
                    }
                }
                const double sc = ys / yy;
#pragma unroll
                for (int r = 0; r < NVL; ++r) dreg[r] *= sc;
                __syncthreads(); // alpha[] written by lane 0 is read back below
                BE_STAMP(21)
                if (gp->stamps && blockIdx.x == 0 && threadIdx.x == 0) { GLBQ long long* st_ = (GLBQ long long*)gp->stamps; st_[40] += 1; st_[41] += bound; }
                // second loop: pairs j, j + 1, ... (oldest first)
                auto second_loop = [&](const Chunk& c, int cnt, auto full) {
#pragma unroll
                    for (int cidx = 0; cidx < CH; ++cidx) {
                        if (decltype(full)::value || cidx < cnt) {
                            double sv[NVL], yv[NVL], ysj, rysj;
                            pair_values(c, cidx, n, lane, sv, yv, ysj, rysj);
                            double p = 0.0;
#pragma unroll
                            for (int r = 0; r < NVL; ++r) p += yv[r] * dreg[r];
                            const double beta = div_by_known(uni(wave_sum(p)), ysj, rysj);
                            const double a = L.alpha[j];
#pragma unroll
                            for (int r = 0; r < NVL; ++r) dreg[r] += (a - beta) * sv[r];
                            j = (j + 1 == m) ? 0 : j + 1;
                        }
                    }
                };
                {
                    Chunk A, B;
                    left = bound;
                    int cntA = min(CH, left), cntB;
                    load_pairs(A, hist, nstride, m, j, 1, lane);
                    for (;;) {
                        left -= cntA;
                        cntB = min(CH, left);
                        load_pairs(B, hist, nstride, m, wrap(j + cntA), 1, lane);
                        if (cntA == CH) second_loop(A, cntA, std::true_type{}); else second_loop(A, cntA, std::false_type{});
                        if (cntB == 0) break;
                        left -= cntB;
                        cntA = min(CH, left);
                        load_pairs(A, hist, nstride, m, wrap(j + cntB), 1, lane);
                        if (cntB == CH) second_loop(B, cntB, std::true_type{}); else second_loop(B, cntB, std::false_type{});
                        if (cntA == 0) break;
                    }
                }
#pragma unroll
                for (int r = 0; r < NVL; ++r)
                    if (lane + 64 * r < n) L.d[lane + 64 * r] = dreg[r];
                BE_STAMP(22)
                }
            }
            __syncthreads();
            step = 1.0;
        }
    }
    f_out = fx;
    k_out = k;
    return ret;
}

// final collision check (optimizer.cpp:474-571): 16 Simpson panels per piece, ESDF at the panel ends
template <int P>
__device__ __attribute__((noinline)) bool final_collision(const Params* gp_in, unsigned lds_in, double& min_dist)
{
    CParams* gp = (CParams*)uni_ptr(gp_in);
    CParams& prm = *gp;
    CConfig& c = prm.cfg;
    LDSQ Lds<P>& L = lds<P>(uni((int)lds_in));
    const LDSQ EvalCtx& e = L.e;
    const int lane = threadIdx.x, M = uni(e.M), R = c.final_check_num, NP = R * M;
    const double xvI = c.standard_diff ? 0.0 : c.icr_xv, map_inv = 1.0 / prm.map.res;
    double carry_x = e.start_xy[0], carry_y = e.start_xy[1], mind = 1.79769313486231570815e+308;
    int first_hit = 0x7fffffff;
    for (int base = 0; base < NP; base += 64) {
        const int p = base + lane;
        double ix = 0.0, iy = 0.0;
        if (p < NP) {
            const int i = p / R, q = p - i * R;
            const double T = L.T[i], half = T / R / 2.0, cint = T / R / 6.0;
            const LDSQ double* ci = L.coef + 12 * i;
            for (int s = 0; s < 3; ++s) {
                const double t = (2 * q + s) * half;
                double sg, d1[2];
                sg = ((((ci[10] * t + ci[8]) * t + ci[6]) * t + ci[4]) * t + ci[2]) * t + ci[0];
                for (int d = 0; d < 2; ++d)
                    d1[d] = (((5.0 * ci[10 + d] * t + 4.0 * ci[8 + d]) * t + 3.0 * ci[6 + d]) * t + 2.0 * ci[4 + d]) * t + ci[2 + d];
                double sy, cy;
                sincos(sg, &sy, &cy);
                const double w = (s == 1) ? 4.0 * cint : cint;
                ix += w * (d1[1] * cy + d1[0] * xvI * sy);
                iy += w * (d1[1] * sy - d1[0] * xvI * cy);
            }
        }
        ix = wave_prefix(ix); iy = wave_prefix(iy);
        if (p < NP) {
            double gx, gy;
            const double sd = esdf(prm.map, map_inv, carry_x + ix, carry_y + iy, false, 0.0, gx, gy);
            if (sd < c.final_min_safe_dis && p < first_hit) first_hit = p;
            L.fx[lane] = sd; // staging for the ordered minimum below
        }
        __syncthreads();
        // the reference stops at the first hit: its reported minimum covers the panels up to and including it
        int hit = first_hit;
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) hit = min(hit, __shfl_xor(hit, mm));
        double local = (p < NP && p <= hit) ? L.fx[lane] : 1.79769313486231570815e+308;
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) local = fmin(local, __shfl_xor(local, mm));
        mind = fmin(mind, local);
        carry_x += __shfl(ix, 63);
        carry_y += __shfl(iy, 63);
        first_hit = hit;
        __syncthreads();
        if (hit != 0x7fffffff) break;
    }
    min_dist = uni(mind);
    return uni(first_hit) != 0x7fffffff;
}

template <int P>
__device__ void load_problem(const Params& prm, unsigned lbase, int b)
{
    LDSQ Lds<P>& L = lds<P>(lbase);
    LDSQ EvalCtx& e = L.e;
    const ProblemStore& s = prm.prob;
    if (threadIdx.x == 0) {
        e.M = s.M[b];
        e.prob = b;
        e.n = 3 * e.M - 1;
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < 3; ++k) { e.head[d][k] = s.head[(size_t)b * 6 + d * 3 + k]; e.tail[d][k] = s.tail[(size_t)b * 6 + d * 3 + k]; }
        e.start_xy[0] = s.start_xy[(size_t)b * 2]; e.start_xy[1] = s.start_xy[(size_t)b * 2 + 1];
        e.final_xy[0] = s.final_xy[(size_t)b * 2]; e.final_xy[1] = s.final_xy[(size_t)b * 2 + 1];
        e.positions = s.positions + (size_t)b * s.P * 2;
        e.evals = 0;
        e.xy_err[0] = e.xy_err[1] = 0.0;
    }
    __syncthreads();
}

} // namespace

// Round 5: LDS admits EIGHT workgroups per CU (20.0 KB each; round 4: 40.4 KB, four) and the kernel keeps to 256 registers
// (waves_per_eu(2, 2): 256 and 340 B of scratch against 254 and 300 B with a SIMD to itself): two wavefronts on every SIMD.
// 8192 plans 49.0 -> 34.7 ms, the same evaluations plan by plan (a launch is total work / slots: a block padded to three
// workgroups per CU takes 63.5 ms; four 49.0, five 44.1, six 38.8, seven 37.2, eight 34.7 -- one residency of 2048 plans pays 5 %
// for the longer evaluation: 18.5 -> 19.5 ms).  What left the LDS: the elimination factors of the knot system, the order-2 node
// terms, the duration-gradient terms and cos / sin of the heading per node go through a per-problem global workspace (written once or
// twice, read once or twice per evaluation, L2; in registers the factors cost 292 registers, eliminating twice 9 % more
// instructions); the node terms are stored without their structural zeros; the knot-state gradient shares the bytes of the Simpson
// poses (dead by then), the two-loop recursion's alpha[] those of a node array (it runs between two evaluations).  What fits a CU
// was measured (tools/micro/lds_fit.hip): 8 workgroups up to 20480 B each, 7 up to 23040, 6 up to 26880, 5 up to 31744 -- not
// 160 KB / n: the runtime's occupancy query is one too optimistic just below those sizes.
template <int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void backend_kernel(const Params* __restrict__ gp)
{
    if (gp->stamps && blockIdx.x == 0 && threadIdx.x == 0) gp->stamps[63] = (long long)__builtin_readcyclecounter();
    const Params& prm = *gp;
    const unsigned lbase = lds_base_of_kernel();
    LDSQ Lds<P>& L = lds<P>(lbase);
    LDSQ EvalCtx& e = L.e;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= prm.count) return;
    const int b = prm.order ? prm.order[blockIdx.x] : (int)blockIdx.x;
    const Config& c = prm.cfg;
    load_problem<P>(prm, lbase, b);
    const int M = uni(e.M), n = uni(e.n), nstride = 3 * prm.prob.P;
    double* hist = prm.hist + (size_t)b * MEM_MAX * 2 * nstride;
    const bool cut = prm.prob.if_cut[b] != 0;

    if (prm.mode == MODE_EVAL || prm.mode == MODE_LBFGS) {
        for (int v = lane; v < n; v += 64) L.x[v] = prm.x_io[(size_t)b * nstride + v];
        for (int v = lane; v < n; v += 64) L.g[v] = 0.0;
        if (lane == 0) {
            for (int q = 0; q < 2; ++q) {
                e.lam[q] = prm.lam_in ? prm.lam_in[(size_t)b * 2 + q] : (cut ? c.cut_lam0[q] : c.lam0[q]);
                e.rho[q] = prm.rho_in ? prm.rho_in[(size_t)b * 2 + q] : (cut ? c.cut_rho0[q] : c.rho0[q]);
            }
            e.stage = prm.stage;
            e.safe_dis = prm.safe_dis;
            e.time_weight = prm.time_weight;
        }
        __syncthreads();
        double cost;
        int ret = 0, iters = 0;
        if (prm.mode == MODE_EVAL) {
            cost = uni(eval_cost<P>(gp, lbase));
        } else {
            LbfgsParam pr = prm.stage == 1 ? c.path_lbfgs : c.lbfgs;
            if (prm.stage == 1 && fabs(e.tail[1][0]) < c.shot_path_horizon) pr.past = c.shot_path_past;
            ret = uni(lbfgs<P>(gp, lbase, pr, hist, nstride, prm.max_iter, cost, iters));
        }
        __syncthreads();
        for (int v = lane; v < n; v += 64) {
            prm.g_out[(size_t)b * nstride + v] = L.g[v];
            prm.x_io[(size_t)b * nstride + v] = L.x[v];
        }
        if (lane == 0) {
            prm.cost_out[b] = cost;
            prm.err_out[(size_t)b * 2] = e.xy_err[0];
            prm.err_out[(size_t)b * 2 + 1] = e.xy_err[1];
            prm.ret_out[(size_t)b * 3] = ret;
            prm.ret_out[(size_t)b * 3 + 1] = iters;
            prm.ret_out[(size_t)b * 3 + 2] = e.evals;
        }
        return;
    }

    // ---- MODE_PLAN: MSPlanner::minco_plan
    double safe;
    { // getDistanceReal(start) * 0.85 (nearest cell, sdf_map.cpp:865-871)
        const MapView& m = prm.map;
        double dr = 10000.0;
        const double x = e.start_xy[0], y = e.start_xy[1];
        if (!(x < m.x_lo || y < m.y_lo || x > m.x_hi || y > m.y_hi)) {
            const double inv = 1.0 / m.res;
            const int ix = min(max((int)((x - m.x_lo) * inv), 0), m.nx - 1), iy = min(max((int)((y - m.y_lo) * inv), 0), m.ny - 1);
            dr = m.dist[(size_t)ix * m.ny + iy];
        }
        safe = fmin(dr * 0.85, c.safe_dis);
    }
    double tw = c.w_time, cost = 0.0, min_dist = 0.0;
    int attempts = 0, alm_rounds = 0, lb_ret = 0, path_ret = 0, total_evals = 0;
    bool collision = true;
    const double tol = cut ? c.cut_tol : c.tol;
    const double tail_s0 = prm.prob.tail[(size_t)b * 6 + 3];
    while (attempts < c.safe_replan_max) {
        // get_state + the initial decision vector (optimizer.cpp:222-249, 277-286)
        __syncthreads();
        for (int v = lane; v < 2 * (M - 1); v += 64) L.x[v] = prm.prob.inner[(size_t)b * (prm.prob.P - 1) * 2 + v];
        const double tau0 = tau_of_t(prm.prob.init_T[b]);
        for (int i = lane; i < M; i += 64) L.x[2 * (M - 1) + 1 + i] = tau0;
        if (lane == 0) {
            L.x[2 * (M - 1)] = tail_s0;
            e.tail[1][0] = tail_s0;
            for (int q = 0; q < 2; ++q) { e.lam[q] = cut ? c.cut_lam0[q] : c.lam0[q]; e.rho[q] = cut ? c.cut_rho0[q] : c.rho0[q]; }
            e.safe_dis = safe;
            e.time_weight = tw;
            e.evals = 0;
            e.stage = 1;
        }
        __syncthreads();
        // stage 1
        LbfgsParam pp = c.path_lbfgs;
        if (fabs(tail_s0) < c.shot_path_horizon) pp.past = c.shot_path_past;
        int iters;
        path_ret = uni(lbfgs<P>(gp, lbase, pp, hist, nstride, 0, cost, iters));
        // stage 2: augmented-Lagrangian loop
        __syncthreads();
        if (lane == 0) e.stage = 2;
        __syncthreads();
        alm_rounds = 0;
        for (;;) {
            lb_ret = uni(lbfgs<P>(gp, lbase, c.lbfgs, hist, nstride, 0, cost, iters));
            ++alm_rounds;
            __syncthreads();
            const double ex = uni(e.xy_err[0]), ey = uni(e.xy_err[1]);
            if (sqrt(ex * ex + ey * ey) < tol) break;
            if (alm_rounds >= c.max_alm_rounds) break;
            __syncthreads();
            if (lane == 0)
                for (int q = 0; q < 2; ++q) {
                    e.lam[q] += e.rho[q] * e.xy_err[q];
                    e.rho[q] = fmin((1 + (cut ? c.cut_gamma[q] : c.gamma[q])) * e.rho[q], cut ? c.cut_rho_max[q] : c.rho_max[q]);
                }
            __syncthreads();
        }
        __syncthreads();
        total_evals += uni(e.evals);
        ++attempts;
        // final trajectory: setParameters(finalInnerpoints, finalpieceTime) -- one more pass leaves T and the
        // coefficients of the final x in LDS (its cost and gradient are not used)
        (void)eval_cost<P>(gp, lbase);
        __syncthreads();
        collision = uni((int)final_collision<P>(gp, lbase, min_dist)) != 0;
        if (!collision) break;
        tw *= 0.75;
    }
    // ---- results
    __syncthreads();
    const ResultStore& r = prm.res;
    for (int v = lane; v < 2 * (M - 1); v += 64) r.inner[(size_t)b * (prm.prob.P - 1) * 2 + v] = L.x[v];
    for (int i = lane; i < M; i += 64) r.T[(size_t)b * prm.prob.P + i] = L.T[i];
    for (int v = lane; v < 12 * M; v += 64) r.coef[(size_t)b * prm.prob.P * 12 + v] = L.coef[v];
    if (lane == 0) {
        Status& st = r.status[b];
        st.ok = collision ? 0 : 1;
        st.attempts = attempts;
        st.alm_rounds = alm_rounds;
        st.evals = total_evals;
        st.lbfgs_ret = lb_ret;
        st.path_ret = path_ret;
        st.collision = collision ? 1 : 0;
        st.n_pieces = M;
        st.cost = cost;
        st.xy_err[0] = e.xy_err[0];
        st.xy_err[1] = e.xy_err[1];
        st.min_dist = min_dist;
        st.tail_s = L.x[2 * (M - 1)];
        r.ok[b] = collision ? 0 : 1;
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < 3; ++k) r.tail[(size_t)b * 6 + d * 3 + k] = (d == 1 && k == 0) ? L.x[2 * (M - 1)] : e.tail[d][k];
    }
}

// eight workgroups of the 16-piece build share a CU's 160 KB: 20000 B today (round 4: 40400 B, four workgroups)
static_assert(16 * NS >= MEM_MAX, "alpha[] shares the bytes of fx[]");
static_assert(sizeof(Lds<16>) <= 20480, "Lds<16> must leave room for eight workgroups per CU (tools/micro/lds_fit.hip: 8 x 20480 B fit, 7 up to 23040, 6 up to 26880, 5 up to 31744)");

size_t lds_bytes(int P)
{
    return P <= 16 ? sizeof(Lds<16>) : sizeof(Lds<32>);
}

hipError_t launch(const Params& p, const Params* d_params, int P, hipStream_t s)
{
    const void* fn = P <= 16 ? (const void*)backend_kernel<16> : (const void*)backend_kernel<32>;
    const size_t lds = lds_bytes(P);
    static size_t configured[16][2] = {{0}};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 15;
    const int v = P <= 16 ? 0 : 1;
    if (lds > configured[dev][v]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev][v] = lds;
    }
    e = hipMemcpyAsync(const_cast<Params*>(d_params), &p, sizeof(Params), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    void* args[] = {&d_params};
    e = hipLaunchKernel(fn, dim3(p.count), dim3(64), args, lds, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

} // namespace backend
