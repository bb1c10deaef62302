// nmpc_core.h -- per-stage numerics of the batched NMPC real-time iteration.
//
// Device functions used by the HIP kernels in nmpc_kernels.hip.  They are plain
// scalar float32 code (one value per lane), so the same header also compiles
// with a host compiler: tests/harness/cpu_core_harness.cpp strings the functions
// together for ONE problem on the CPU to check the algebra against the oracle
// when no GPU is around.  That harness is test-only; the product library only
// contains the GPU path.
//
// What is computed (reference = /root/reference/planning_ddr_opt/nmpc_controller,
// CG = UAV_CAR_model/build/quadrotor_mpc_codegen):
//  * ddr_linearize   -- one shooting interval of the ICR skid-steer model:
//    the state after one 2-stage Gauss-Legendre step plus dphi/dx, dphi/du
//    (reference: CG/acado_integrator.c:62-123 model, :261-449 IRK + forward
//    sensitivities by Newton iterations on a 6x6 system).  The model's heading
//    rate does not depend on the state and (x, y) do not feed back, so the
//    implicit stage equations are explicit here: psi_s = psi + c_s*h*w and the
//    step is a 2-point Gauss quadrature of f(psi_s).  Same map, same
//    derivatives, no Newton loop, no LU.
//  * stage cost      -- Gauss-Newton stage data from the weighting matrix
//    (reference: CG/acado_solver.c:103-211 slicing of W, :365-440 gradient).
//  * riccati_step / forward_step -- the reference condenses the QP to a dense
//    2N x 2N problem (CG/acado_solver.c:327-363) and runs qpOASES' active-set
//    homotopy on it (externals/qpoases/SRC/QProblemB.cpp).  The QP is strictly
//    convex, so its minimiser is unique; here it is computed stage-wise: for a
//    given working set (each control free / at lower / at upper bound) a
//    Riccati sweep solves the equality-constrained QP exactly in O(N), the
//    forward sweep returns the step and the bound multipliers, and the working
//    set is updated primal-dual style until it reproduces itself.
#ifndef ALORE_NMPC_CORE_H
#define ALORE_NMPC_CORE_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NMPC_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define NMPC_HD inline
#endif

namespace nmpc {

// working-set status of one control (same meaning as qpOASES SubjectToStatus)
enum : int { ST_FREE = 0, ST_LOWER = 1, ST_UPPER = 2 };

// per-problem status codes: qpOASES returnValue numbers where one applies
// (externals/qpoases/INCLUDE/MessageHandling.hpp:64-170)
enum : int {
    RET_OK = 0,
    RET_INIT_FAILED_CHOLESKY = 31,      // a stage Hessian pivot was not positive
    RET_INIT_FAILED_INFEASIBILITY = 33, // lb > ub for some control
    RET_MAX_NWSR_REACHED = 58           // working set still changing at the iteration cap
};

// tolerances of the working-set update
constexpr float TOL_PRIMAL = 1e-6f; // a free control must leave the box by this much to be fixed
constexpr float TOL_DUAL = 1e-7f;   // a multiplier must have the wrong sign by this much to be freed
constexpr float BOUNDTOL = 1e-10f;  // lb, ub closer than this: equality (Constants.hpp:88)

// integrator constants for step h: 2-stage Gauss-Legendre nodes
struct IrkConst {
    float h;   // step
    float hh;  // h/2 (both weights)
    float c1h; // c1*h, c1 = 1/2 + sqrt(3)/6
    float c2h; // c2*h, c2 = 1/2 - sqrt(3)/6
};

NMPC_HD IrkConst make_irk(float h)
{
    IrkConst c;
    c.h = h;
    c.hh = 0.5f * h;
    c.c1h = (float)(0.78867513459481288 * (double)h);
    c.c2h = (float)(0.21132486540518712 * (double)h);
    return c;
}

// 1/x to float32 accuracy: hardware reciprocal + one Newton step on the device
NMPC_HD float rcp_f(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
#else
    return 1.0f / x;
#endif
}

// reciprocal of a Riccati pivot: the hardware reciprocal is accurate to 1 ulp, like the division it stands for
NMPC_HD float pivot_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// sin and cos enter the step only multiplied by h/2 (5e-3 for the reference's dt): the hardware
// sine/cosine (argument in revolutions, absolute error ~1e-6) perturb the shooting defect by
// < 1e-8, far inside the float32 rounding of the state itself
NMPC_HD void sincos_f(float x, float* s, float* c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float t = x * 0.15915494309189535f;
    *s = __builtin_amdgcn_sinf(t);
    *c = __builtin_amdgcn_cosf(t);
#else
    *s = (float)std::sin((double)x);
    *c = (float)std::cos((double)x);
#endif
}

// linearisation of one shooting interval.  Gx = I + [0 0 a; 0 0 b; 0 0 0],
// Gu = [B00 B01; B10 B11; B20 -B20] (columns: vr, vl).
struct StageLin {
    float a, b;
    float B00, B01, B10, B11, B20;
    float phi0, phi1, phi2; // state at the end of the interval
};

// x = (px, py, psi), u = (vr, vl), od = (xv, yr, yl)
NMPC_HD void ddr_linearize(const IrkConst& K, float px, float py, float psi, float vr, float vl, float xv,
                           float yr, float yl, StageLin& o)
{
    const float inv = rcp_f(yl - yr);
    const float dv = vr - vl;
    const float w = dv * inv;                     // heading rate
    const float lon = (vr * yl - vl * yr) * inv;  // body-x speed
    const float lat = w * xv;                     // lateral slip speed
    float s1, c1, s2, c2;
    sincos_f(psi + K.c1h * w, &s1, &c1);
    sincos_f(psi + K.c2h * w, &s2, &c2);
    const float C = c1 + c2, S = s1 + s2;
    const float Cw = K.c1h * c1 + K.c2h * c2; // sum_s c_s h cos(psi_s)
    const float Sw = K.c1h * s1 + K.c2h * s2;
    const float fx = lon * C + lat * S;  // sum_s f_x(psi_s)
    const float fy = lon * S - lat * C;  // sum_s f_y(psi_s)
    o.phi0 = px + K.hh * fx;
    o.phi1 = py + K.hh * fy;
    o.phi2 = psi + K.h * w;
    o.a = -K.hh * fy; // d phi_x / d psi
    o.b = K.hh * fx;  // d phi_y / d psi
    // d/d(vr), d/d(vl) of (lon, lat, w)
    const float lon_r = yl * inv, lon_l = -yr * inv;
    const float lat_r = xv * inv, lat_l = -lat_r;
    const float gxw = lat * Cw - lon * Sw; // sum_s c_s h * d f_x/d psi (psi_s)
    const float gyw = lon * Cw + lat * Sw;
    o.B00 = K.hh * (lon_r * C + lat_r * S + inv * gxw);
    o.B01 = K.hh * (lon_l * C + lat_l * S - inv * gxw);
    o.B10 = K.hh * (lon_r * S - lat_r * C + inv * gyw);
    o.B11 = K.hh * (lon_l * S - lat_l * C - inv * gyw);
    o.B20 = K.h * inv; // d phi_psi / d vr ; d/d vl = -B20
}

// symmetric 3x3 as 6 floats
struct Sym3 {
    float m00, m01, m02, m11, m12, m22;
};

// quadratic cost-to-go  V(dx) = 1/2 dx' P dx + p' dx
struct Value {
    Sym3 P;
    float p0, p1, p2;
};

// everything the backward step needs about stage k
struct StageQP {
    float a, b, B00, B01, B10, B11, B20; // dynamics dx+ = A dx + B du + d
    float d0, d1, d2;
    Sym3 Q;            // state Hessian of stage k (unused for k = 0)
    float q0, q1, q2;  // state gradient
    float R00, R01, R11;
    float r0, r1;      // control gradient
    int st0, st1;      // working-set status of the two controls
    float v0, v1;      // the bound a fixed control sits on
};

// affine records produced by the backward step, consumed by the forward step:
//   control 0:  val0 = c0 . dx + f0
//   control 1:  val1 = c1 . dx + e1 * du0 + f1
// val is the control move if the control is free, its bound multiplier if fixed.
struct Policy {
    float c00, c01, c02, f0;
    float c10, c11, c12, e1, f1;
};

// One backward Riccati step with the controls eliminated one at a time
// (control 1 first, then control 0); a fixed control is substituted by its
// bound value instead of being minimised over.  Straight-line code: the two
// cases of each control differ only in selected operands.  Returns false if a
// pivot needed for a free control is not positive.
NMPC_HD bool riccati_step(const StageQP& s, Value& V, Policy& pol, bool need_value)
{
    const Sym3 P = V.P;
    const float B21 = -s.B20;
    // s = P d + p
    const float s0 = P.m00 * s.d0 + P.m01 * s.d1 + P.m02 * s.d2 + V.p0;
    const float s1 = P.m01 * s.d0 + P.m11 * s.d1 + P.m12 * s.d2 + V.p1;
    const float s2 = P.m02 * s.d0 + P.m12 * s.d1 + P.m22 * s.d2 + V.p2;
    // PB = P B
    const float PB00 = P.m00 * s.B00 + P.m01 * s.B10 + P.m02 * s.B20;
    const float PB10 = P.m01 * s.B00 + P.m11 * s.B10 + P.m12 * s.B20;
    const float PB20 = P.m02 * s.B00 + P.m12 * s.B10 + P.m22 * s.B20;
    const float PB01 = P.m00 * s.B01 + P.m01 * s.B11 + P.m02 * B21;
    const float PB11 = P.m01 * s.B01 + P.m11 * s.B11 + P.m12 * B21;
    const float PB21 = P.m02 * s.B01 + P.m12 * s.B11 + P.m22 * B21;
    // Huu = R + B' P B
    const float H00 = s.R00 + s.B00 * PB00 + s.B10 * PB10 + s.B20 * PB20;
    const float H01 = s.R01 + s.B00 * PB01 + s.B10 * PB11 + s.B20 * PB21;
    const float H11 = s.R11 + s.B01 * PB01 + s.B11 * PB11 + B21 * PB21;
    // Hux = B' P A, A = I + e(a,b): row j = (PB0j, PB1j, a PB0j + b PB1j + PB2j)
    float G00 = PB00, G01 = PB10, G02 = s.a * PB00 + s.b * PB10 + PB20;
    const float G10 = PB01, G11 = PB11, G12 = s.a * PB01 + s.b * PB11 + PB21;
    // hu = r + B' s
    float hu0 = s.r0 + s.B00 * s0 + s.B10 * s1 + s.B20 * s2;
    const float hu1 = s.r1 + s.B01 * s0 + s.B11 * s1 + B21 * s2;

    // ---- eliminate control 1
    const bool free1 = (s.st1 == ST_FREE);
    const bool bad1 = free1 && !(H11 > 0.0f);
    const float inv11 = pivot_rcp(H11);
    const float w1 = free1 ? inv11 : 0.0f;        // 1/H11 if minimised over
    const float z1 = free1 ? -hu1 * inv11 : s.v1; // value of du1 at dx = 0, du0 = 0
    const float t1 = w1 * H01;
    const float g1s = free1 ? -w1 : 1.0f; // record = g1s * (G1, H01) for both cases
    pol.c10 = g1s * G10;
    pol.c11 = g1s * G11;
    pol.c12 = g1s * G12;
    pol.e1 = g1s * H01;
    pol.f1 = free1 ? z1 : hu1 + H11 * s.v1;
    const float H00r = H00 - t1 * H01;
    G00 -= t1 * G10; G01 -= t1 * G11; G02 -= t1 * G12;
    hu0 += H01 * z1;
    // ---- eliminate control 0
    const bool free0 = (s.st0 == ST_FREE);
    const bool bad0 = free0 && !(H00r > 0.0f);
    const float inv00 = pivot_rcp(H00r);
    const float w0 = free0 ? inv00 : 0.0f;
    const float z0 = free0 ? -hu0 * inv00 : s.v0;
    const float g0s = free0 ? -w0 : 1.0f;
    pol.c00 = g0s * G00;
    pol.c01 = g0s * G01;
    pol.c02 = g0s * G02;
    pol.f0 = free0 ? z0 : hu0 + H00r * s.v0;
    const bool ok = !(bad0 || bad1);
    if (!need_value) return ok;

    // Hxx = Q + A' P A  and hx = q + A' s
    const float m02 = s.a * P.m00 + s.b * P.m01 + P.m02;
    const float m12 = s.a * P.m01 + s.b * P.m11 + P.m12;
    const float m22 = s.a * P.m02 + s.b * P.m12 + P.m22;
    Sym3 X;
    X.m00 = s.Q.m00 + P.m00;
    X.m01 = s.Q.m01 + P.m01;
    X.m11 = s.Q.m11 + P.m11;
    X.m02 = s.Q.m02 + m02;
    X.m12 = s.Q.m12 + m12;
    X.m22 = s.Q.m22 + (s.a * m02 + s.b * m12 + m22);
    float hx0 = s.q0 + s0;
    float hx1 = s.q1 + s1;
    float hx2 = s.q2 + (s.a * s0 + s.b * s1 + s2);
    // control 1 out:  Hxx -= w1 G1'G1 ; hx += G1' z1   (G1 = original row 1)
    const float wg10 = w1 * G10, wg11 = w1 * G11, wg12 = w1 * G12;
    X.m00 -= wg10 * G10; X.m01 -= wg10 * G11; X.m02 -= wg10 * G12;
    X.m11 -= wg11 * G11; X.m12 -= wg11 * G12; X.m22 -= wg12 * G12;
    hx0 += G10 * z1; hx1 += G11 * z1; hx2 += G12 * z1;
    // control 0 out (G0 is the reduced row)
    const float wg00 = w0 * G00, wg01 = w0 * G01, wg02 = w0 * G02;
    X.m00 -= wg00 * G00; X.m01 -= wg00 * G01; X.m02 -= wg00 * G02;
    X.m11 -= wg01 * G01; X.m12 -= wg01 * G02; X.m22 -= wg02 * G02;
    hx0 += G00 * z0; hx1 += G01 * z0; hx2 += G02 * z0;
    V.P = X;
    V.p0 = hx0; V.p1 = hx1; V.p2 = hx2;
    return ok;
}

// result of the forward step at one stage
struct StageStep {
    float du0, du1; // control move (fixed controls sit on their bound)
    float mu0, mu1; // bound multipliers (0 for free controls), qpOASES sign: >0 at lower, <0 at upper
    int nst0, nst1; // updated working-set status
};

// primal-dual working-set update of one control, branch-free
NMPC_HD int next_status(int st, float val, float lb, float ub)
{
    const int from_free = (val < lb - TOL_PRIMAL) ? ST_LOWER : ((val > ub + TOL_PRIMAL) ? ST_UPPER : ST_FREE);
    const int from_lower = (val < -TOL_DUAL) ? ST_FREE : ST_LOWER;
    const int from_upper = (val > TOL_DUAL) ? ST_FREE : ST_UPPER;
    const int ns = (st == ST_FREE) ? from_free : ((st == ST_LOWER) ? from_lower : from_upper);
    return (ub - lb > BOUNDTOL) ? ns : ST_LOWER; // equality-bounded controls stay fixed
}

NMPC_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// dx = state deviation at stage k; lb/ub = QP bounds on du (lbValues - u, ubValues - u)
NMPC_HD void forward_step(const Policy& pol, int st0, int st1, float dx0, float dx1, float dx2, float lb0,
                          float ub0, float lb1, float ub1, StageStep& o)
{
    const float val0 = pol.c00 * dx0 + pol.c01 * dx1 + pol.c02 * dx2 + pol.f0;
    const float b0 = (st0 == ST_UPPER) ? ub0 : lb0;
    const float raw0 = (st0 == ST_FREE) ? val0 : b0; // exact solution of the working-set QP
    const float val1 = pol.c10 * dx0 + pol.c11 * dx1 + pol.c12 * dx2 + pol.e1 * raw0 + pol.f1;
    const float b1 = (st1 == ST_UPPER) ? ub1 : lb1;
    const float raw1 = (st1 == ST_FREE) ? val1 : b1;
    o.nst0 = next_status(st0, val0, lb0, ub0);
    o.nst1 = next_status(st1, val1, lb1, ub1);
    // raw values propagate (they solve the working-set QP exactly); once the
    // working set has settled they are inside the box up to TOL_PRIMAL
    o.du0 = raw0;
    o.du1 = raw1;
    o.mu0 = (st0 == ST_FREE) ? 0.0f : val0;
    o.mu1 = (st1 == ST_FREE) ? 0.0f : val1;
}

// ---- safeguard: primal active-set iteration -----------------------------------------------------
// The primal-dual update above changes every violated control at once; it is exact when it stops but
// it can cycle (seen on far-off warm starts with stale duals).  After AS_SWITCH sweeps the solver
// continues as a textbook primal active-set method on the same stage-wise factorisation: a feasible
// point `cur` moves towards the minimiser `tgt` of the current working-set QP up to the first blocking
// bound (that control is added), a full step releases the fixed control whose multiplier has the most
// wrong sign, and a full step without such a control is the solution.  One Riccati sweep per change,
// finite for a strictly convex QP.
constexpr int AS_SWITCH = 16;
constexpr float AS_NONE = 3.0e38f;

// step length at which free control (cur -> tgt) meets a bound it would cross; AS_NONE if it does not
NMPC_HD float asm_ratio(int st, float cur, float tgt, float lb, float ub, int& hit)
{
    hit = ST_FREE;
    if (st != ST_FREE) return AS_NONE;
    if (tgt > ub + TOL_PRIMAL) { hit = ST_UPPER; return (ub - cur) / (tgt - cur); }
    if (tgt < lb - TOL_PRIMAL) { hit = ST_LOWER; return (lb - cur) / (tgt - cur); }
    return AS_NONE;
}
// how wrong the sign of the multiplier of a fixed control is (0: fine)
NMPC_HD float asm_violation(int st, float mult, float lb, float ub)
{
    if (!(ub - lb > BOUNDTOL)) return 0.0f; // equality-bounded controls stay fixed
    if (st == ST_LOWER) return (mult < -TOL_DUAL) ? -mult : 0.0f;
    if (st == ST_UPPER) return (mult > TOL_DUAL) ? mult : 0.0f;
    return 0.0f;
}
// working set of a feasible point: a control sitting on a bound is fixed there
NMPC_HD int asm_status_of(float cur, float lb, float ub)
{
    return (cur <= lb) ? ST_LOWER : ((cur >= ub) ? ST_UPPER : ST_FREE);
}

// initial working set from the previous dual, as qpOASES guesses it
// (QProblemB.cpp:1010-1036 with ZERO == 0 in float)
NMPC_HD int status_from_dual(float y, float lb, float ub)
{
    const int s = (y > 0.0f) ? ST_LOWER : ((y < 0.0f) ? ST_UPPER : ST_FREE);
    return (ub - lb > BOUNDTOL) ? s : ST_LOWER;
}

} // namespace nmpc
#endif
