// wb_dynamics.h -- rigid-body dynamics of the B2 + Z1 tree for one GPU lane (float64, registers only).
//
// The reference has no whole-body dynamics (SURVEY.md 8(a) row A-RB); the model numbers come from its URDF through
// tools/gen_b2z1_model.py (b2z1_model.h).  Formulation: recursive Newton-Euler in 3-D vectors, body coordinates,
// classical accelerations -- the tree is five chains on a floating base (4 legs x 3 joints, arm x 6 joints), every
// joint frame a pure translation and every axis a coordinate axis, so a joint transform is one plane rotation.  A
// chain is walked down (velocities, accelerations, body wrenches; 6 doubles kept per link) and back up (joint
// torques, wrench handed to the parent); chains only meet at the base.  Everything is unrolled over the constexpr
// model table: no indexed register arrays survive.
//
//   q = [p | rpy (ZYX) | 18 joint angles]     v = [omega_base | v_base (base frame) | 18 joint rates]
//   tau = RNEA(q, v, a, f):  rows 0-2 moment about the base origin, 3-5 force (base frame), 6-23 joint torques;
//   foot forces f are WORLD-frame forces at the four foot points.  Same conventions as oracle/wb_oracle.py, which
//   uses 6-D spatial algebra instead.
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WB_FN __device__ __forceinline__
// the five chains are independent: left alone, the scheduler interleaves them and keeps 400+ registers live
#define WB_FENCE __builtin_amdgcn_sched_barrier(0);
#else // host build of the same source: tests/harness/wb_dynamics_harness.cpp (CPU test of the formulas, no GPU)
#define WB_FN inline
#define WB_FENCE
// glibc declares sincos(double, double*, double*) in <cmath> under _GNU_SOURCE (g++ default)
#endif

#include "b2z1_model.h"

namespace wb {

struct V3 {
    double x, y, z;
};
WB_FN V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
WB_FN V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
WB_FN V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
WB_FN V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// R(axis, q) v  and  R(axis, q)^T v  with c = cos q, s = sin q; R maps child coordinates to parent coordinates
template <int AX>
WB_FN V3 rot(double c, double s, V3 v)
{
    if (AX == 0) return {v.x, c * v.y - s * v.z, s * v.y + c * v.z};
    if (AX == 1) return {c * v.x + s * v.z, v.y, -s * v.x + c * v.z};
    return {c * v.x - s * v.y, s * v.x + c * v.y, v.z};
}
template <int AX>
WB_FN V3 rotT(double c, double s, V3 v)
{
    if (AX == 0) return {v.x, c * v.y + s * v.z, -s * v.y + c * v.z};
    if (AX == 1) return {c * v.x - s * v.z, v.y, s * v.x + c * v.z};
    return {c * v.x + s * v.y, -s * v.x + c * v.y, v.z};
}
template <int AX>
WB_FN V3 unit() { return {AX == 0 ? 1.0 : 0.0, AX == 1 ? 1.0 : 0.0, AX == 2 ? 1.0 : 0.0}; }
template <int AX>
WB_FN double comp(V3 v) { return AX == 0 ? v.x : (AX == 1 ? v.y : v.z); }

template <int I>
WB_FN V3 origin() { return {b2z1::ORIGIN[3 * I], b2z1::ORIGIN[3 * I + 1], b2z1::ORIGIN[3 * I + 2]}; }
template <int I>
WB_FN V3 com() { return {b2z1::COM[3 * I], b2z1::COM[3 * I + 1], b2z1::COM[3 * I + 2]}; }
template <int I>
WB_FN V3 inertia_times(V3 w)
{
    constexpr double xx = b2z1::INERTIA[6 * I], xy = b2z1::INERTIA[6 * I + 1], xz = b2z1::INERTIA[6 * I + 2];
    constexpr double yy = b2z1::INERTIA[6 * I + 3], yz = b2z1::INERTIA[6 * I + 4], zz = b2z1::INERTIA[6 * I + 5];
    return {xx * w.x + xy * w.y + xz * w.z, xy * w.x + yy * w.y + yz * w.z, xz * w.x + yz * w.y + zz * w.z};
}
// wrench of body I about its own origin from its angular velocity / acceleration and the acceleration of its origin
template <int I>
WB_FN void body_wrench(V3 w, V3 wd, V3 ac, V3& F, V3& N)
{
    const V3 c = com<I>();
    const V3 acom = ac + cross(wd, c) + cross(w, cross(w, c));
    F = b2z1::MASS[I] * acom;
    N = inertia_times<I>(wd) + cross(w, inertia_times<I>(w)) + cross(c, F);
}

// One evaluation point, described relative to base vectors kept in LDS (every lane of a wavefront evaluates the same
// (q, v, a, f) up to a perturbation, a unit vector or a mask): value = scale * base[i] + (i == unit ? amount : 0).
struct Eval {
    const double* q;   // [24]  p | rpy | joints   (p is never read)
    const double* v;   // [24]
    const double* a;   // [24]
    const double* f;   // [12] world-frame foot forces
    double sv, sa, sf; // scales of v, a, f (0 or 1)
    int uq, uv, ua, uf; // index that receives an increment (-1: none)
    double dq, dv, da, df;
    double g;          // gravity (0 to switch it off)
    // sin / cos of the 21 angles q[3..23] at the base point (sin at [2 t], cos at [2 t + 1], t = index - 3), or NULL;
    // with a table, the angle that carries the increment dq is rotated by (sin dq, cos dq) = (ps, pc) instead of
    // being recomputed: one sincos per lane instead of 21 per evaluation
    const double* trig = nullptr;
    double ps = 0.0, pc = 1.0;
    WB_FN void SC(int i, double* s, double* c) const
    {
        if (trig) {
            const double s0 = trig[2 * (i - 3)], c0 = trig[2 * (i - 3) + 1];
            const bool me = i == uq;
            *s = me ? s0 * pc + c0 * ps : s0;
            *c = me ? c0 * pc - s0 * ps : c0;
        } else {
            sincos(Q(i), s, c);
        }
    }
    WB_FN double Q(int i) const { return q[i] + (i == uq ? dq : 0.0); }
    WB_FN double V(int i) const { return sv * v[i] + (i == uv ? dv : 0.0); }
    WB_FN double A(int i) const { return sa * a[i] + (i == ua ? da : 0.0); }
    WB_FN double F(int i) const { return sf * f[i] + (i == uf ? df : 0.0); }
};

// Where the 24 outputs of one evaluation go (kept out of registers: tau_i is stored the moment it is known).
//   mode 0: out[i * stride] = scale * tau_i        mode 1: out[i * stride] = scale * (out[i * stride] - tau_i)
struct Sink {
    double* out;
    int stride;
    double scale;
    int mode;
    WB_FN void put(int i, double val) const
    {
        double* p = out + i * stride;
        *p = mode == 0 ? scale * val : scale * (*p - val);
    }
};

// rotation base -> world from rpy (ZYX); R0^T v
struct BaseRot {
    double cr, sr, cp, sp, cy, sy;
    WB_FN void set(double r, double p, double y) { sincos(r, &sr, &cr); sincos(p, &sp, &cp); sincos(y, &sy, &cy); }
    WB_FN void set(const struct Eval& e);
    WB_FN V3 toBase(V3 v) const { return rotT<0>(cr, sr, rotT<1>(cp, sp, rotT<2>(cy, sy, v))); }
    WB_FN V3 toWorld(V3 v) const { return rot<2>(cy, sy, rot<1>(cp, sp, rot<0>(cr, sr, v))); }
    // d(rpy)/dt = E omega_body
    WB_FN V3 rates(V3 w) const
    {
        const double tp = sp / cp;
        return {w.x + sr * tp * w.y + cr * tp * w.z, cr * w.y - sr * w.z, (sr * w.y + cr * w.z) / cp};
    }
};

WB_FN void BaseRot::set(const Eval& e) { e.SC(3, &sr, &cr); e.SC(4, &sp, &cp); e.SC(5, &sy, &cy); }

template <int FIRST, int D, int LEN, int FOOT>
struct Chain {
    // walks link FIRST + D given the parent's (w, wd, ac) and the foot force in the parent's frame; returns the wrench
    // (about the PARENT's origin, parent coordinates) that this link and everything below it needs
    static WB_FN void walk(const Eval& e, V3 w, V3 wd, V3 ac, V3 fe, const Sink& tau, V3& f_up, V3& n_up)
    {
        constexpr int I = FIRST + D;
        constexpr int AX = b2z1::AXIS[I];
        const V3 p = origin<I>();
        double s, c;
        e.SC(5 + I, &s, &c);
        const double qd = e.V(5 + I), qdd = e.A(5 + I);
        const V3 ax = unit<AX>();
        const V3 wl = rotT<AX>(c, s, w);
        const V3 aci = rotT<AX>(c, s, ac + cross(wd, p) + cross(w, cross(w, p)));
        const V3 wdi = rotT<AX>(c, s, wd) + qdd * ax + qd * cross(wl, ax);
        const V3 wi = wl + qd * ax;
        const V3 fei = rotT<AX>(c, s, fe);
        V3 F, N;
        body_wrench<I>(wi, wdi, aci, F, N);
        if (D + 1 < LEN) {
            V3 fc, nc;
            Chain<FIRST, (D + 1 < LEN ? D + 1 : D), LEN, FOOT>::walk(e, wi, wdi, aci, fei, tau, fc, nc);
            F = F + fc;
            N = N + nc;
        } else if (FOOT >= 0) {
            const V3 r = {b2z1::FOOT_POINT[3 * (FOOT >= 0 ? FOOT : 0)], b2z1::FOOT_POINT[3 * (FOOT >= 0 ? FOOT : 0) + 1],
                          b2z1::FOOT_POINT[3 * (FOOT >= 0 ? FOOT : 0) + 2]};
            F = F - fei;
            N = N - cross(r, fei);
        }
        tau.put(5 + I, comp<AX>(N));
        const V3 fp = rot<AX>(c, s, F);
        f_up = fp;
        n_up = rot<AX>(c, s, N) + cross(p, fp);
    }
};

// The four legs are one structure with four sets of constants (joint axes x, y, y at depths 0, 1, 2): ONE copy of the walk, run
// four times over the model table (wave-uniform index: scalar loads), instead of four unrolled copies -- the stage kernel's time
// is instruction fetch to a sixth (its RNEA alone was 55 KB against a 64 KB instruction cache shared by two CUs).
static_assert(b2z1::AXIS[1] == 0 && b2z1::AXIS[2] == 1 && b2z1::AXIS[3] == 1 && b2z1::AXIS[4] == 0 && b2z1::AXIS[5] == 1 && b2z1::AXIS[6] == 1 &&
                  b2z1::AXIS[7] == 0 && b2z1::AXIS[8] == 1 && b2z1::AXIS[9] == 1 && b2z1::AXIS[10] == 0 && b2z1::AXIS[11] == 1 && b2z1::AXIS[12] == 1,
              "the legs share their joint axes");
WB_FN V3 origin_rt(int I) { return {b2z1::ORIGIN[3 * I], b2z1::ORIGIN[3 * I + 1], b2z1::ORIGIN[3 * I + 2]}; }
WB_FN void body_wrench_rt(int I, V3 w, V3 wd, V3 ac, V3& F, V3& N)
{
    const V3 c = {b2z1::COM[3 * I], b2z1::COM[3 * I + 1], b2z1::COM[3 * I + 2]};
    const double xx = b2z1::INERTIA[6 * I], xy = b2z1::INERTIA[6 * I + 1], xz = b2z1::INERTIA[6 * I + 2];
    const double yy = b2z1::INERTIA[6 * I + 3], yz = b2z1::INERTIA[6 * I + 4], zz = b2z1::INERTIA[6 * I + 5];
    auto inertia = [&](V3 u) -> V3 { return {xx * u.x + xy * u.y + xz * u.z, xy * u.x + yy * u.y + yz * u.z, xz * u.x + yz * u.y + zz * u.z}; };
    const V3 acom = ac + cross(wd, c) + cross(w, cross(w, c));
    F = b2z1::MASS[I] * acom;
    N = inertia(wd) + cross(w, inertia(w)) + cross(c, F);
}
template <int D>
struct LegChain {
    static WB_FN void walk(const Eval& e, int first, int foot, V3 w, V3 wd, V3 ac, V3 fe, const Sink& tau, V3& f_up, V3& n_up)
    {
        const int I = first + D;
        constexpr int AX = D == 0 ? 0 : 1;
        const V3 p = origin_rt(I);
        double s, c;
        e.SC(5 + I, &s, &c);
        const double qd = e.V(5 + I), qdd = e.A(5 + I);
        const V3 ax = unit<AX>();
        const V3 wl = rotT<AX>(c, s, w);
        const V3 aci = rotT<AX>(c, s, ac + cross(wd, p) + cross(w, cross(w, p)));
        const V3 wdi = rotT<AX>(c, s, wd) + qdd * ax + qd * cross(wl, ax);
        const V3 wi = wl + qd * ax;
        const V3 fei = rotT<AX>(c, s, fe);
        V3 F, N;
        body_wrench_rt(I, wi, wdi, aci, F, N);
        if (D + 1 < 3) {
            V3 fc, nc;
            LegChain<(D + 1 < 3 ? D + 1 : D)>::walk(e, first, foot, wi, wdi, aci, fei, tau, fc, nc);
            F = F + fc;
            N = N + nc;
        } else {
            const V3 r = {b2z1::FOOT_POINT[3 * foot], b2z1::FOOT_POINT[3 * foot + 1], b2z1::FOOT_POINT[3 * foot + 2]};
            F = F - fei;
            N = N - cross(r, fei);
        }
        tau.put(5 + I, comp<AX>(N));
        const V3 fp = rot<AX>(c, s, F);
        f_up = fp;
        n_up = rot<AX>(c, s, N) + cross(p, fp);
    }
};

// the 24 rows of RNEA at the evaluation point go to the sink
WB_FN void rnea(const Eval& e, const Sink& tau)
{
    BaseRot R;
    R.set(e);
    const V3 w0 = {e.V(0), e.V(1), e.V(2)}, vl = {e.V(3), e.V(4), e.V(5)};
    const V3 wd0 = {e.A(0), e.A(1), e.A(2)};
    const V3 ac0 = V3{e.A(3), e.A(4), e.A(5)} + cross(w0, vl) + R.toBase({0.0, 0.0, e.g});
    V3 F, N;
    body_wrench<0>(w0, wd0, ac0, F, N);
    V3 fc, nc;
    WB_FENCE
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int leg = 0; leg < 4; ++leg) {
        LegChain<0>::walk(e, 1 + 3 * leg, leg, w0, wd0, ac0, R.toBase({e.F(3 * leg), e.F(3 * leg + 1), e.F(3 * leg + 2)}), tau, fc, nc);
        F = F + fc;
        N = N + nc;
        WB_FENCE
    }
    Chain<13, 0, 6, -1>::walk(e, w0, wd0, ac0, {0.0, 0.0, 0.0}, tau, fc, nc);
    F = F + fc;
    N = N + nc;
    tau.put(0, N.x); tau.put(1, N.y); tau.put(2, N.z);
    tau.put(3, F.x); tau.put(4, F.y); tau.put(5, F.z);
}

} // namespace wb
