// oracle/minco_band.h -- TEST INFRASTRUCTURE ONLY (used by oracle/traj_oracle.hpp).
//
// The reference's formulation of the minimum-jerk spline, restated: one 6M x 6M banded system over all
// quintic coefficients, pivot-free band LU -- P/back_end/include/gcopter/minco.hpp:43-199 (BandedSystem,
// storage (i, j) at [(i - j + q) * n + j]) and :817-898 (MINCO_S3NU::setParameters), with
// P = /root/reference/planning_ddr_opt.  The product solves the same spline through its knot states
// (alore_legged_manipulator_amd/csrc/minco_spline.h); this file is what that is checked against, together
// with the dense NumPy solve of tests/test_backend_oracle.py.  Not linked into any product library.
#pragma once

#include <cmath>

#define MINCO_HD inline

namespace minco_band {

constexpr int BAND_LOWER = 6, BAND_UPPER = 6, BAND_ROWS = BAND_LOWER + BAND_UPPER + 1;

// doubles needed for the band matrix / the right-hand side of an M-piece spline
MINCO_HD int band_doubles(int M) { return BAND_ROWS * 6 * M; }
MINCO_HD int rhs_doubles(int M) { return 6 * M * 2; }

struct Band {
    double* a;
    int n;
    MINCO_HD double& at(int i, int j) const { return a[(i - j + BAND_UPPER) * n + j]; }
};

MINCO_HD void band_lu(const Band& A)
{
    const int n = A.n;
    for (int k = 0; k <= n - 2; ++k) {
        const int iM = (k + BAND_LOWER < n - 1) ? k + BAND_LOWER : n - 1;
        double cVl = A.at(k, k);
        for (int i = k + 1; i <= iM; ++i)
            if (A.at(i, k) != 0.0) A.at(i, k) /= cVl;
        const int jM = (k + BAND_UPPER < n - 1) ? k + BAND_UPPER : n - 1;
        for (int j = k + 1; j <= jM; ++j) {
            cVl = A.at(k, j);
            if (cVl != 0.0)
                for (int i = k + 1; i <= iM; ++i)
                    if (A.at(i, k) != 0.0) A.at(i, j) -= A.at(i, k) * cVl;
        }
    }
}

// b: n rows of 2 columns, row-major, solved in place
MINCO_HD void band_solve2(const Band& A, double* b)
{
    const int n = A.n;
    for (int j = 0; j <= n - 1; ++j) {
        const int iM = (j + BAND_LOWER < n - 1) ? j + BAND_LOWER : n - 1;
        for (int i = j + 1; i <= iM; ++i)
            if (A.at(i, j) != 0.0) {
                b[i * 2] -= A.at(i, j) * b[j * 2];
                b[i * 2 + 1] -= A.at(i, j) * b[j * 2 + 1];
            }
    }
    for (int j = n - 1; j >= 0; --j) {
        b[j * 2] /= A.at(j, j);
        b[j * 2 + 1] /= A.at(j, j);
        const int iM = (j - BAND_UPPER > 0) ? j - BAND_UPPER : 0;
        for (int i = iM; i <= j - 1; ++i)
            if (A.at(i, j) != 0.0) {
                b[i * 2] -= A.at(i, j) * b[j * 2];
                b[i * 2 + 1] -= A.at(i, j) * b[j * 2 + 1];
            }
    }
}

// Minimum-jerk spline through `inner` ((M-1) x 2) with boundary p/v/a `head`, `tail` ([dim][p, v, a]) and
// piece durations T[M].  band (band_doubles(M)) and rhs (rhs_doubles(M)) are workspace; on return
// rhs[(6 i + k) * 2 + d] is the coefficient of t^k of dimension d on piece i.
MINCO_HD void spline_solve(int M, const double* T, const double* inner, const double head[2][3],
                           const double tail[2][3], double* band, double* rhs)
{
    const int n = 6 * M;
    for (int i = 0; i < band_doubles(M); ++i) band[i] = 0.0;
    for (int i = 0; i < rhs_doubles(M); ++i) rhs[i] = 0.0;
    const Band A{band, n};
    A.at(0, 0) = 1.0; A.at(1, 1) = 1.0; A.at(2, 2) = 2.0;
    for (int d = 0; d < 2; ++d) { rhs[0 * 2 + d] = head[d][0]; rhs[1 * 2 + d] = head[d][1]; rhs[2 * 2 + d] = head[d][2]; }
    for (int i = 0; i < M - 1; ++i) {
        const int r = 6 * i;
        const double t1 = T[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        A.at(r + 3, r + 3) = 6.0; A.at(r + 3, r + 4) = 24.0 * t1; A.at(r + 3, r + 5) = 60.0 * t2; A.at(r + 3, r + 9) = -6.0;
        A.at(r + 4, r + 4) = 24.0; A.at(r + 4, r + 5) = 120.0 * t1; A.at(r + 4, r + 10) = -24.0;
        A.at(r + 5, r) = 1.0; A.at(r + 5, r + 1) = t1; A.at(r + 5, r + 2) = t2; A.at(r + 5, r + 3) = t3;
        A.at(r + 5, r + 4) = t4; A.at(r + 5, r + 5) = t5;
        A.at(r + 6, r) = 1.0; A.at(r + 6, r + 1) = t1; A.at(r + 6, r + 2) = t2; A.at(r + 6, r + 3) = t3;
        A.at(r + 6, r + 4) = t4; A.at(r + 6, r + 5) = t5; A.at(r + 6, r + 6) = -1.0;
        A.at(r + 7, r + 1) = 1.0; A.at(r + 7, r + 2) = 2 * t1; A.at(r + 7, r + 3) = 3 * t2; A.at(r + 7, r + 4) = 4 * t3;
        A.at(r + 7, r + 5) = 5 * t4; A.at(r + 7, r + 7) = -1.0;
        A.at(r + 8, r + 2) = 2.0; A.at(r + 8, r + 3) = 6 * t1; A.at(r + 8, r + 4) = 12 * t2; A.at(r + 8, r + 5) = 20 * t3;
        A.at(r + 8, r + 8) = -2.0;
        rhs[(r + 5) * 2] = inner[i * 2];
        rhs[(r + 5) * 2 + 1] = inner[i * 2 + 1];
    }
    const int e = 6 * M;
    const double t1 = T[M - 1], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
    A.at(e - 3, e - 6) = 1.0; A.at(e - 3, e - 5) = t1; A.at(e - 3, e - 4) = t2; A.at(e - 3, e - 3) = t3;
    A.at(e - 3, e - 2) = t4; A.at(e - 3, e - 1) = t5;
    A.at(e - 2, e - 5) = 1.0; A.at(e - 2, e - 4) = 2 * t1; A.at(e - 2, e - 3) = 3 * t2; A.at(e - 2, e - 2) = 4 * t3;
    A.at(e - 2, e - 1) = 5 * t4;
    A.at(e - 1, e - 4) = 2; A.at(e - 1, e - 3) = 6 * t1; A.at(e - 1, e - 2) = 12 * t2; A.at(e - 1, e - 1) = 20 * t3;
    for (int d = 0; d < 2; ++d) {
        rhs[(e - 3) * 2 + d] = tail[d][0]; rhs[(e - 2) * 2 + d] = tail[d][1]; rhs[(e - 1) * 2 + d] = tail[d][2];
    }
    band_lu(A);
    band_solve2(A, rhs);
}

// ---- evaluation on a stored trajectory: dur[n], coef[n][2][6] (ascending powers) -----------------
MINCO_HD int locate(const double* dur, int n, double& t)
{ // Trajectory::locatePieceIdx: t becomes the local time; past the end the last piece is extrapolated
    int idx;
    double d = 0.0;
    for (idx = 0; idx < n && t > (d = dur[idx]); ++idx) t -= d;
    if (idx == n) {
        --idx;
        t += dur[idx];
    }
    return idx;
}
MINCO_HD void eval_pv(const double* dur, const double* coef, int n, double t, double p[2], double v[2])
{
    double tl = t;
    const int i = locate(dur, n, tl);
    const double* c = coef + i * 12;
    for (int d = 0; d < 2; ++d) {
        double tn = 1.0, pp = 0.0, vv = 0.0;
        const double* cd = c + d * 6;
        for (int k = 0; k <= 5; ++k) { pp += tn * cd[k]; tn *= tl; }
        tn = 1.0;
        for (int k = 1; k <= 5; ++k) { vv += k * tn * cd[k]; tn *= tl; }
        p[d] = pp; v[d] = vv;
    }
}
// world-frame velocity of the tracked point for flat state p = (theta, s), v = (theta', s'), ICR offset xv
MINCO_HD double xdot(const double p[2], const double v[2], double xv) { return v[1] * cos(p[0]) + v[0] * xv * sin(p[0]); }
MINCO_HD double ydot(const double p[2], const double v[2], double xv) { return v[1] * sin(p[0]) - v[0] * xv * cos(p[0]); }

// Simpson increment of (x, y) over [t0, t0 + len]
MINCO_HD void simpson_panel(const double* dur, const double* coef, int n, double xv, double t0, double len, double& dx,
                            double& dy)
{
    double p1[2], p2[2], p3[2], v1[2], v2[2], v3[2];
    eval_pv(dur, coef, n, t0, p1, v1);
    eval_pv(dur, coef, n, t0 + len / 2.0, p2, v2);
    eval_pv(dur, coef, n, t0 + len, p3, v3);
    dx = len / 6.0 * (xdot(p1, v1, xv) + 4.0 * xdot(p2, v2, xv) + xdot(p3, v3, xv));
    dy = len / 6.0 * (ydot(p1, v1, xv) + 4.0 * ydot(p2, v2, xv) + ydot(p3, v3, xv));
}

} // namespace minco_band
