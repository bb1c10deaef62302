// minco_spline.h -- the minimum-jerk (s = 3) spline of the reference's flat space (theta, s), solved the
// way a GPU wants it.
//
// The reference (P/back_end/include/gcopter/minco.hpp:817-898, MINCO_S3NU::setParameters) stacks the 6
// coefficients of all M quintic pieces into one 6M x 6M banded system (continuity rows up to the 4th
// derivative, way-point rows, boundary rows) and runs a pivot-free band LU over 13 diagonals (:99-157),
// and its adjoint (:169-199, :1139-1209) for gradients.  Here the same spline is parametrised by its KNOT
// STATES: position (given: head, way-points, tail), velocity and acceleration.  A quintic is fixed by
// (p, v, a) at both ends (Hermite form, closed form below), which satisfies the position / velocity /
// acceleration continuity and way-point rows identically; what remains are the jerk and snap continuity
// rows at the M - 1 interior knots, i.e. the stationarity of the jerk energy w.r.t. (v_k, a_k).  Written as
//      F_k = E1(T_{k-1}) (z_{k-1}, z_k) - E0(T_k) (z_k, z_{k+1}) = 0,   E = (-snap, jerk) at a piece end,
// that is a SYMMETRIC POSITIVE DEFINITE block-tridiagonal system with 2 x 2 blocks in the unknowns
// y_k = (v_k, a_k) (it is the Hessian of the energy), solved by a block Thomas sweep without pivoting in
// O(M) with ~60 flops per knot, both flat dimensions sharing the factorisation; the coefficients then
// follow per (piece, dimension) independently -- lane-parallel.  Because the matrix is symmetric the
// adjoint solve of the gradient propagation is the same sweep with another right-hand side.
// Same spline as the reference's to rounding (tests: oracle/backend_oracle.c be_spline, dense NumPy solve).
//
// Plain doubles and pointers; compiled by hipcc for the kernels and by g++ for the CPU unit test.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MINCO_HD __host__ __device__ inline
#else
#define MINCO_HD inline
#endif

namespace minco {

// powers of 1 / T of one piece
struct InvT {
    double i1, i2, i3, i4, i5;
    MINCO_HD explicit InvT(double T)
    {
        i1 = 1.0 / T; i2 = i1 * i1; i3 = i2 * i1; i4 = i2 * i2; i5 = i4 * i1;
    }
    // with 1 / T at hand (formed once per piece by the caller: the division heads a dependent chain wherever this is built)
    MINCO_HD InvT(double, double inv)
    {
        i1 = inv; i2 = i1 * i1; i3 = i2 * i1; i4 = i2 * i2; i5 = i4 * i1;
    }
};

// Quintic through (p0, v0, a0) at 0 and (p1, v1, a1) at T, ascending coefficients c[0..5];
// dc (optional) = d c / d T with the knot states held fixed.
MINCO_HD void hermite(double T, const InvT& q, double p0, double v0, double a0, double p1, double v1, double a1,
                      double c[6], double* dc)
{
    const double D = p1 - p0 - v0 * T - 0.5 * a0 * T * T, Dv = v1 - v0 - a0 * T, Da = a1 - a0;
    c[0] = p0; c[1] = v0; c[2] = 0.5 * a0;
    c[3] = (10.0 * D - 4.0 * Dv * T + 0.5 * Da * T * T) * q.i3;
    c[4] = (-15.0 * D + 7.0 * Dv * T - Da * T * T) * q.i4;
    c[5] = (6.0 * D - 3.0 * Dv * T + 0.5 * Da * T * T) * q.i5;
    if (dc) {
        const double Dp = -v0 - a0 * T, Dvp = -a0;
        dc[0] = dc[1] = dc[2] = 0.0;
        dc[3] = (10.0 * Dp - 4.0 * Dv - 4.0 * T * Dvp + Da * T) * q.i3 - 3.0 * c[3] * q.i1;
        dc[4] = (-15.0 * Dp + 7.0 * Dv + 7.0 * T * Dvp - 2.0 * Da * T) * q.i4 - 4.0 * c[4] * q.i1;
        dc[5] = (6.0 * Dp - 3.0 * Dv - 3.0 * T * Dvp + Da * T) * q.i5 - 5.0 * c[5] * q.i1;
    }
}

// transpose of d c / d (p0, v0, a0, p1, v1, a1) applied to a coefficient gradient G[6]
MINCO_HD void hermite_adjoint(const InvT& q, const double G[6], double g0[3], double g1[3])
{
    g0[0] = G[0] - 10.0 * q.i3 * G[3] + 15.0 * q.i4 * G[4] - 6.0 * q.i5 * G[5];
    g0[1] = G[1] - 6.0 * q.i2 * G[3] + 8.0 * q.i3 * G[4] - 3.0 * q.i4 * G[5];
    g0[2] = 0.5 * G[2] - 1.5 * q.i1 * G[3] + 1.5 * q.i2 * G[4] - 0.5 * q.i3 * G[5];
    g1[0] = 10.0 * q.i3 * G[3] - 15.0 * q.i4 * G[4] + 6.0 * q.i5 * G[5];
    g1[1] = -4.0 * q.i2 * G[3] + 7.0 * q.i3 * G[4] - 3.0 * q.i4 * G[5];
    g1[2] = 0.5 * q.i1 * G[3] - q.i2 * G[4] + 0.5 * q.i3 * G[5];
}

// 2 x 2 blocks of the knot system.  Row order (-snap, jerk), unknown order (v, a).
//   lower(T_left)  = coefficient of y_{k-1};  upper(T_right) = coefficient of y_{k+1} = lower(T_right)^T
//   diag = dl(T_left) + dr(T_right)
struct Sym2 { double a, b, c; }; // [[a, b], [b, c]]
struct Mat2 { double a, b, c, d; }; // [[a, b], [c, d]]
MINCO_HD Mat2 knot_upper(const InvT& r) { return Mat2{168.0 * r.i3, -24.0 * r.i2, 24.0 * r.i2, -3.0 * r.i1}; }
MINCO_HD Sym2 knot_diag(const InvT& l, const InvT& r)
{
    return Sym2{192.0 * (l.i3 + r.i3), 36.0 * (r.i2 - l.i2), 9.0 * (l.i1 + r.i1)};
}
// right-hand side of knot k from the three positions around it
MINCO_HD void knot_rhs(const InvT& l, const InvT& r, double pm, double p, double pp, double out[2])
{
    out[0] = 360.0 * l.i4 * (p - pm) + 360.0 * r.i4 * (pp - p);
    out[1] = -60.0 * l.i3 * (p - pm) + 60.0 * r.i3 * (pp - p);
}
// d F_k / d p of the neighbouring knots (for the gradient w.r.t. way-points): F_k = K y - rhs
//   dF_k/dp_{k-1} = (360 l4, -60 l3);  dF_k/dp_{k+1} = (-360 r4, -60 r3);  dF_k/dp_k = -(sum of the two)
MINCO_HD Sym2 inv(const Sym2& s)
{
    const double det = s.a * s.c - s.b * s.b, id = 1.0 / det;
    return Sym2{s.c * id, -s.b * id, s.a * id};
}

// Block Thomas for the SPD knot system with nk = M - 1 interior knots.
//   T[M] durations; sinv[nk] receives the inverted Schur blocks (kept for further solves);
//   rhs[nk][2] is overwritten by the solution (v_k, a_k).
MINCO_HD void knot_factor(int M, const double* T, Sym2* sinv)
{
    Sym2 prev{0, 0, 0};
    for (int k = 1; k < M; ++k) {
        const InvT l(T[k - 1]), r(T[k]);
        Sym2 s = knot_diag(l, r);
        if (k > 1) { // s -= L S^-1 L^T with L = upper(T_{k-1})^T
            const Mat2 u = knot_upper(l); // U_{k-1}; L_k = U^T = [[u.a, u.c], [u.b, u.d]]
            // W = S^-1 U  (2x2), then L W = U^T W
            const double w00 = prev.a * u.a + prev.b * u.c, w01 = prev.a * u.b + prev.b * u.d;
            const double w10 = prev.b * u.a + prev.c * u.c, w11 = prev.b * u.b + prev.c * u.d;
            s.a -= u.a * w00 + u.c * w10;
            s.b -= u.a * w01 + u.c * w11;
            s.c -= u.b * w01 + u.d * w11;
        }
        prev = inv(s);
        sinv[k - 1] = prev;
    }
}
MINCO_HD void knot_solve(int M, const double* T, const Sym2* sinv, double* rhs /* [nk][2], stride 2 */, int stride = 2)
{
    for (int k = 2; k < M; ++k) { // forward: rhs_k -= L_k S_{k-1}^-1 rhs_{k-1}
        const InvT l(T[k - 1]);
        const Mat2 u = knot_upper(l);
        const Sym2& s = sinv[k - 2];
        const double* rp = rhs + (size_t)(k - 2) * stride;
        double* rk = rhs + (size_t)(k - 1) * stride;
        const double w0 = s.a * rp[0] + s.b * rp[1], w1 = s.b * rp[0] + s.c * rp[1];
        rk[0] -= u.a * w0 + u.c * w1;
        rk[1] -= u.b * w0 + u.d * w1;
    }
    for (int k = M - 1; k >= 1; --k) { // backward: y_k = S_k^-1 (rhs_k - U_k y_{k+1})
        double* rk = rhs + (size_t)(k - 1) * stride;
        double r0 = rk[0], r1 = rk[1];
        if (k < M - 1) {
            const InvT r(T[k]);
            const Mat2 u = knot_upper(r);
            const double* yn = rhs + (size_t)k * stride;
            r0 -= u.a * yn[0] + u.b * yn[1];
            r1 -= u.c * yn[0] + u.d * yn[1];
        }
        const Sym2& s = sinv[k - 1];
        rk[0] = s.a * r0 + s.b * r1;
        rk[1] = s.b * r0 + s.c * r1;
    }
}

// (-snap, jerk) at the start (E0) and at the end (E1) of a piece, and their T-derivatives with the knot
// states fixed, from its coefficients c and dc = dc/dT
MINCO_HD void piece_end_rows(double T, const double c[6], const double dc[6], double dE0[2], double dE1[2])
{
    dE0[0] = -24.0 * dc[4];
    dE0[1] = 6.0 * dc[3];
    dE1[0] = -(24.0 * dc[4] + 120.0 * dc[5] * T + 120.0 * c[5]);
    dE1[1] = 6.0 * dc[3] + 24.0 * dc[4] * T + 60.0 * dc[5] * T * T + 24.0 * c[4] + 120.0 * c[5] * T;
}

// ---- whole-spline helpers for one flat dimension (used one thread per (message, dimension) when a
//      planner message is turned into coefficients; the optimiser kernel spreads the same steps over lanes)
// p[M+1] knot positions, (v0, a0), (vM, aM) boundary states; work: sinv[M-1], y[(M-1)*2]; coef out: c[i*6+q]
MINCO_HD void spline_1d(int M, const double* T, const double* p, double v0, double a0, double vM, double aM, Sym2* sinv,
                        double* y, double* coef)
{
    for (int k = 1; k < M; ++k) {
        const InvT l(T[k - 1]), r(T[k]);
        knot_rhs(l, r, p[k - 1], p[k], p[k + 1], y + 2 * (k - 1));
        if (k == 1) { // known (v0, a0): rhs -= L_1 y_0, L_1 = upper(T_0)^T
            const Mat2 u = knot_upper(l);
            y[0] -= u.a * v0 + u.c * a0;
            y[1] -= u.b * v0 + u.d * a0;
        }
        if (k == M - 1) { // known (vM, aM): rhs -= U_{M-1} y_M
            const Mat2 u = knot_upper(r);
            y[2 * (k - 1)] -= u.a * vM + u.b * aM;
            y[2 * (k - 1) + 1] -= u.c * vM + u.d * aM;
        }
    }
    knot_factor(M, T, sinv);
    knot_solve(M, T, sinv, y);
    for (int i = 0; i < M; ++i) {
        const InvT q(T[i]);
        const double vs = i == 0 ? v0 : y[2 * (i - 1)], as = i == 0 ? a0 : y[2 * (i - 1) + 1];
        const double ve = i == M - 1 ? vM : y[2 * i], ae = i == M - 1 ? aM : y[2 * i + 1];
        hermite(T[i], q, p[i], vs, as, p[i + 1], ve, ae, coef + 6 * i, nullptr);
    }
}

// ---- evaluation on a stored trajectory: dur[n], coef[n][2][6] (ascending powers) -----------------
MINCO_HD int locate(const double* dur, int n, double& t)
{ // piece lookup of the reference's Trajectory class (P/back_end/include/gcopter/trajectory.hpp:472-490):
  // t becomes the local time; past the end the last piece is extrapolated
    int idx;
    double d = 0.0;
    for (idx = 0; idx < n && t > (d = dur[idx]); ++idx) t -= d;
    if (idx == n) {
        --idx;
        t += dur[idx];
    }
    return idx;
}
// The same lookup on durations that are already in registers: dreg[i] = dur[i] for i < min(n, PRE) (loaded by the caller with
// independent loads).  locate() above issues one DEPENDENT load per piece it walks past -- three evaluations of a trajectory
// of ten pieces were thirty round trips to L2 in the reference sampler.  Same subtraction chain, same bits.
template <int PRE>
MINCO_HD int locate_pre(const double (&dreg)[PRE], const double* dur, int n, double& t)
{
    int idx = 0;
    bool go = true;
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const bool step = go && i < n && t > dreg[i];
        t = step ? t - dreg[i] : t;
        idx += step ? 1 : 0;
        go = step;
    }
    if (go && idx < n) { // longer than the preloaded part: the plain walk for the rest
        double d = 0.0;
        for (; idx < n && t > (d = dur[idx]); ++idx) t -= d;
    }
    if (idx == n) {
        --idx;
        t += dur[idx];
    }
    return idx;
}
template <int PRE>
MINCO_HD void eval_pv_pre(const double (&dreg)[PRE], const double* dur, const double* coef, int n, double t, double p[2], double v[2])
{
    double tl = t;
    const int i = locate_pre<PRE>(dreg, dur, n, tl);
    const double* c = coef + i * 12;
    for (int d = 0; d < 2; ++d) {
        const double* cd = c + d * 6;
        p[d] = ((((cd[5] * tl + cd[4]) * tl + cd[3]) * tl + cd[2]) * tl + cd[1]) * tl + cd[0];
        v[d] = (((5.0 * cd[5] * tl + 4.0 * cd[4]) * tl + 3.0 * cd[3]) * tl + 2.0 * cd[2]) * tl + cd[1];
    }
}
MINCO_HD void eval_pv(const double* dur, const double* coef, int n, double t, double p[2], double v[2])
{
    double tl = t;
    const int i = locate(dur, n, tl);
    const double* c = coef + i * 12;
    for (int d = 0; d < 2; ++d) {
        const double* cd = c + d * 6;
        p[d] = ((((cd[5] * tl + cd[4]) * tl + cd[3]) * tl + cd[2]) * tl + cd[1]) * tl + cd[0];
        v[d] = (((5.0 * cd[5] * tl + 4.0 * cd[4]) * tl + 3.0 * cd[3]) * tl + 2.0 * cd[2]) * tl + cd[1];
    }
}
MINCO_HD void eval_a(const double* dur, const double* coef, int n, double t, double a[2])
{
    double tl = t;
    const int i = locate(dur, n, tl);
    const double* c = coef + i * 12;
    for (int d = 0; d < 2; ++d) {
        const double* cd = c + d * 6;
        a[d] = ((20.0 * cd[5] * tl + 12.0 * cd[4]) * tl + 6.0 * cd[3]) * tl + 2.0 * cd[2];
    }
}
// world-frame velocity of the tracked point for flat state p = (theta, s), v = (theta', s'), ICR offset xv
MINCO_HD double xdot(const double p[2], const double v[2], double xv) { return v[1] * cos(p[0]) + v[0] * xv * sin(p[0]); }
MINCO_HD double ydot(const double p[2], const double v[2], double xv) { return v[1] * sin(p[0]) - v[0] * xv * cos(p[0]); }
// both at once with one sincos (xdot / ydot evaluate cos and sin separately: four float64 trigonometric calls per node)
MINCO_HD void xydot(const double p[2], const double v[2], double xv, double& xd, double& yd)
{
    double sn, cs;
    sincos(p[0], &sn, &cs);
    xd = v[1] * cs + v[0] * xv * sn;
    yd = v[1] * sn - v[0] * xv * cs;
}

// Simpson increment of (x, y) over [t0, t0 + len]
MINCO_HD void simpson_panel(const double* dur, const double* coef, int n, double xv, double t0, double len, double& dx,
                            double& dy)
{
    double p1[2], p2[2], p3[2], v1[2], v2[2], v3[2];
    eval_pv(dur, coef, n, t0, p1, v1);
    eval_pv(dur, coef, n, t0 + len / 2.0, p2, v2);
    eval_pv(dur, coef, n, t0 + len, p3, v3);
    double x1, y1, x2, y2, x3, y3;
    xydot(p1, v1, xv, x1, y1); xydot(p2, v2, xv, x2, y2); xydot(p3, v3, xv, x3, y3);
    dx = len / 6.0 * (x1 + 4.0 * x2 + x3);
    dy = len / 6.0 * (y1 + 4.0 * y2 + y3);
}

} // namespace minco
