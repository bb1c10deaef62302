/* oracle/backend_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * float64 CPU restatement of the reference's back_end trajectory optimiser.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file; the product
 * (alore_legged_manipulator_amd/) never does.
 *
 * PARITY UNPINNED against the reference binary (Eigen, ROS and PCL are absent from this image, the
 * reference has no tests for this path).  Pinned instead by tests/test_backend_oracle.py:
 *   - the spline against a dense numpy.linalg solve of the 6M x 6M system and C0..C4 continuity,
 *   - every gradient against central finite differences (exact_chain = 1 for the obstacle term, see
 *     be_config), the adjoint against the dense transpose solve,
 *   - the Simpson positions against scipy.integrate.quad, the ESDF query against an analytic field.
 *
 * What follows what (P = /root/reference/planning_ddr_opt):
 *   band_*            P/back_end/include/gcopter/minco.hpp:43-199      BandedSystem (no pivoting)
 *   be_spline         minco.hpp:817-898                                 MINCO_S3NU::setParameters
 *   be_energy         minco.hpp:915-992                                 getEnergy + partial gradients
 *   be_spline_adjoint minco.hpp:1139-1209                               propogateArcYawLenghGrad
 *   smoothed_l1       P/back_end/src/optimizer.cpp:1069-1086            positiveSmoothedL1
 *   t_of_tau ...      optimizer.cpp:573-591, 1088-1106                  time diffeomorphism
 *   be_esdf           P/utils/plan_env/src/sdf_map.cpp:760-863          getDistWithGradBilinear (3 overloads)
 *   penalty()         optimizer.cpp:694-1067 (stage 2), :1319-1591 (stage 1)
 *   be_eval           optimizer.cpp:631-692, 1272-1317                  the two L-BFGS callbacks
 *   lbfgs(), search() P/back_end/include/gcopter/lbfgs.hpp:276-396, 440-756
 *   be_optimize       optimizer.cpp:251-472                             stage 1, then the ALM loop
 *   be_final_collision optimizer.cpp:474-571
 *   be_minco_plan     optimizer.cpp:169-220                             retry with 0.75 x time weight
 *
 * Reference quirks kept on purpose (each would change iterates if "fixed"):
 *   - `inf` is the macro `1 >> 30` = 0 (front_end/traj_representation.h:21): a callback called with
 *     |x| > 1e4 returns cost 0 and leaves the gradient untouched (optimizer.cpp:635-636, 1275-1277);
 *   - stage 1 charges PathpenaltyWeights.time_weight in the cost and penaltyWeights.time_weight in the
 *     gradient (optimizer.cpp:1308 vs :1312);
 *   - unOccupied_traj_num_ is -1 (optimizer.cpp:225), so the piece-time balance term never fires;
 *   - the chain rule for obstacle terms counts the node itself with its full Simpson weight
 *     (optimizer.cpp:944-947), and with if_standard_diff = false the derivative of the y increment w.r.t. the
 *     yaw coefficients has one sign wrong (optimizer.cpp:822, 990; the launch default is the standard model,
 *     where the term vanishes); exact_chain = 1 gives the true derivatives for the finite-difference pin;
 *   - the ALM update reads the terminal error of the LAST cost evaluation, which after a failed line
 *     search is a rejected trial point (optimizer.cpp:398, lbfgs.hpp:604-611), and lbfgs returns that
 *     trial's cost.
 */
#include "backend_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- banded system, half-bandwidth 6, row-major band: A(i, j) at a[i * 13 + (j - i + 6)] --------- */
#define BW 6
#define BCOLS (2 * BW + 1)
static inline double *band_at(double *a, int i, int j) { return a + (size_t)i * BCOLS + (j - i + BW); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static void band_lu(double *a, int n)
{
    for (int k = 0; k + 1 < n; ++k) {
        const int last = imin(k + BW, n - 1);
        const double piv = *band_at(a, k, k);
        for (int i = k + 1; i <= last; ++i) {
            double *l = band_at(a, i, k);
            if (*l == 0.0) continue;
            *l /= piv;
            for (int j = k + 1; j <= last; ++j) *band_at(a, i, j) -= *l * *band_at(a, k, j);
        }
    }
}
static void band_solve(const double *a_, int n, double *b /* n x 2 */)
{
    double *a = (double *)a_;
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i <= imin(j + BW, n - 1); ++i) {
            const double l = *band_at(a, i, j);
            b[2 * i] -= l * b[2 * j];
            b[2 * i + 1] -= l * b[2 * j + 1];
        }
    for (int j = n - 1; j >= 0; --j) {
        const double d = *band_at(a, j, j);
        b[2 * j] /= d;
        b[2 * j + 1] /= d;
        for (int i = imax(0, j - BW); i < j; ++i) {
            const double u = *band_at(a, i, j);
            b[2 * i] -= u * b[2 * j];
            b[2 * i + 1] -= u * b[2 * j + 1];
        }
    }
}
static void band_solve_adj(const double *a_, int n, double *b)
{
    double *a = (double *)a_;
    for (int j = 0; j < n; ++j) {
        const double d = *band_at(a, j, j);
        b[2 * j] /= d;
        b[2 * j + 1] /= d;
        for (int i = j + 1; i <= imin(j + BW, n - 1); ++i) {
            const double u = *band_at(a, j, i);
            b[2 * i] -= u * b[2 * j];
            b[2 * i + 1] -= u * b[2 * j + 1];
        }
    }
    for (int j = n - 1; j >= 0; --j)
        for (int i = imax(0, j - BW); i < j; ++i) {
            const double l = *band_at(a, j, i);
            b[2 * i] -= l * b[2 * j];
            b[2 * i + 1] -= l * b[2 * j + 1];
        }
}

/* rows of the minimum-jerk system for piece durations T; a must hold 6M * 13 zeros on entry */
static void spline_matrix(int M, const double *T, double *a)
{
    *band_at(a, 0, 0) = 1.0;
    *band_at(a, 1, 1) = 1.0;
    *band_at(a, 2, 2) = 2.0;
    for (int i = 0; i < M; ++i) {
        const int c = 6 * i; /* first column of piece i */
        const double t = T[i], t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        const double pos[6] = {1.0, t, t2, t3, t4, t5};
        const double vel[6] = {0.0, 1.0, 2 * t, 3 * t2, 4 * t3, 5 * t4};
        const double acc[6] = {0.0, 0.0, 2.0, 6 * t, 12 * t2, 20 * t3};
        if (i + 1 < M) {
            /* junction i: jerk, snap continuity; end position = inner point; position, velocity,
             * acceleration continuity -- rows c+3 .. c+8 */
            *band_at(a, c + 3, c + 3) = 6.0; *band_at(a, c + 3, c + 4) = 24.0 * t; *band_at(a, c + 3, c + 5) = 60.0 * t2;
            *band_at(a, c + 3, c + 9) = -6.0;
            *band_at(a, c + 4, c + 4) = 24.0; *band_at(a, c + 4, c + 5) = 120.0 * t;
            *band_at(a, c + 4, c + 10) = -24.0;
            for (int k = 0; k < 6; ++k) {
                *band_at(a, c + 5, c + k) = pos[k];
                *band_at(a, c + 6, c + k) = pos[k];
                if (k >= 1) *band_at(a, c + 7, c + k) = vel[k];
                if (k >= 2) *band_at(a, c + 8, c + k) = acc[k];
            }
            *band_at(a, c + 6, c + 6) = -1.0;
            *band_at(a, c + 7, c + 7) = -1.0;
            *band_at(a, c + 8, c + 8) = -2.0;
        } else {
            for (int k = 0; k < 6; ++k) {
                *band_at(a, c + 3, c + k) = pos[k];
                if (k >= 1) *band_at(a, c + 4, c + k) = vel[k];
                if (k >= 2) *band_at(a, c + 5, c + k) = acc[k];
            }
        }
    }
}

void be_spline(int M, const double *T, const double *inner, const double head[2][3], const double tail[2][3],
               double *coef)
{
    const int n = 6 * M;
    double *a = (double *)calloc((size_t)n * BCOLS, sizeof(double));
    memset(coef, 0, sizeof(double) * n * 2);
    spline_matrix(M, T, a);
    for (int d = 0; d < 2; ++d) {
        for (int k = 0; k < 3; ++k) {
            coef[2 * k + d] = head[d][k];
            coef[2 * (n - 3 + k) + d] = tail[d][k];
        }
        for (int i = 0; i + 1 < M; ++i) coef[2 * (6 * i + 5) + d] = inner[2 * i + d];
    }
    band_lu(a, n);
    band_solve(a, n, coef);
    free(a);
}

/* getEnergy, getEnergyPartialGradByCoeffs, getEnergyPartialGradByTimes */
double be_energy(int M, const double *T, const double *c, const double w[2], double *gdC, double *gdT)
{
    double e = 0.0;
    for (int i = 0; i < M; ++i) {
        const double t1 = T[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        const double *c3 = c + 2 * (6 * i + 3), *c4 = c3 + 2, *c5 = c4 + 2;
        double d33 = 0, d43 = 0, d44 = 0, d53 = 0, d54 = 0, d55 = 0;
        for (int d = 0; d < 2; ++d) {
            d33 += c3[d] * w[d] * c3[d]; d43 += c4[d] * w[d] * c3[d]; d44 += c4[d] * w[d] * c4[d];
            d53 += c5[d] * w[d] * c3[d]; d54 += c5[d] * w[d] * c4[d]; d55 += c5[d] * w[d] * c5[d];
        }
        e += 36.0 * d33 * t1 + 144.0 * d43 * t2 + 192.0 * d44 * t3 + 240.0 * d53 * t3 + 720.0 * d54 * t4 + 720.0 * d55 * t5;
        if (gdC) {
            double *g = gdC + 2 * 6 * i;
            for (int d = 0; d < 2; ++d) {
                g[0 + d] = g[2 + d] = g[4 + d] = 0.0;
                g[6 + d] = 72.0 * c3[d] * w[d] * t1 + 144.0 * c4[d] * w[d] * t2 + 240.0 * c5[d] * w[d] * t3;
                g[8 + d] = 144.0 * c3[d] * w[d] * t2 + 384.0 * c4[d] * w[d] * t3 + 720.0 * c5[d] * w[d] * t4;
                g[10 + d] = 240.0 * c3[d] * w[d] * t3 + 720.0 * c4[d] * w[d] * t4 + 1440.0 * c5[d] * w[d] * t5;
            }
        }
        if (gdT)
            gdT[i] = 36.0 * d33 + 288.0 * d43 * t1 + 576.0 * d44 * t2 + 720.0 * d53 * t2 + 2880.0 * d54 * t3 + 3600.0 * d55 * t4;
    }
    return e;
}

/* propogateArcYawLenghGrad: lambda = A^-T gdC; gradients w.r.t. inner points, durations, tail position */
void be_spline_adjoint(int M, const double *T, const double *c, const double *gdC, const double *gdT, double *grad_pts,
                       double *grad_T, double grad_tail[2])
{
    const int n = 6 * M;
    double *a = (double *)calloc((size_t)n * BCOLS, sizeof(double));
    double *lam = (double *)malloc(sizeof(double) * n * 2);
    spline_matrix(M, T, a);
    band_lu(a, n);
    memcpy(lam, gdC, sizeof(double) * n * 2);
    band_solve_adj(a, n, lam);
    for (int i = 0; i + 1 < M; ++i) {
        grad_pts[2 * i] = lam[2 * (6 * i + 5)];
        grad_pts[2 * i + 1] = lam[2 * (6 * i + 5) + 1];
    }
    for (int i = 0; i < M; ++i) {
        const double t1 = T[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2;
        const double *ci = c + 2 * 6 * i;
        double s = 0.0;
        for (int d = 0; d < 2; ++d) {
            /* minus the time derivative of each row of A(T) c that contains T_i, i.e. minus the next
             * higher derivative of the piece at its end */
            const double c1 = ci[2 + d], c2 = ci[4 + d], c3 = ci[6 + d], c4 = ci[8 + d], c5 = ci[10 + d];
            const double nvel = -(c1 + 2.0 * t1 * c2 + 3.0 * t2 * c3 + 4.0 * t3 * c4 + 5.0 * t4 * c5);
            const double nacc = -(2.0 * c2 + 6.0 * t1 * c3 + 12.0 * t2 * c4 + 20.0 * t3 * c5);
            const double njer = -(6.0 * c3 + 24.0 * t1 * c4 + 60.0 * t2 * c5);
            const double nsnp = -(24.0 * c4 + 120.0 * t1 * c5);
            const double ncrk = -120.0 * c5;
            const double *l = lam + 2 * (6 * i + 3) + d; /* rows 6i+3 .. (stride 2) */
            if (i + 1 < M)
                s += nsnp * l[0] + ncrk * l[2] + nvel * l[4] + nvel * l[6] + nacc * l[8] + njer * l[10];
            else
                s += nvel * l[0] + nacc * l[2] + njer * l[4];
        }
        grad_T[i] = s + gdT[i];
    }
    grad_tail[0] = lam[2 * (n - 3)];
    grad_tail[1] = lam[2 * (n - 3) + 1];
    free(lam);
    free(a);
}

static void smoothed_l1(double eps, double x, double *f, double *df)
{
    if (x < eps) {
        const double f3 = 1.0 / (eps * eps), f4 = -0.5 * f3 / eps;
        *f = (f4 * x + f3) * x * x * x;
        *df = (4.0 * f4 * x + 3.0 * f3) * x * x;
    } else {
        *f = x - 0.5 * eps;
        *df = 1.0;
    }
}
static double t_of_tau(double v) { return v > 0.0 ? ((0.5 * v + 1.0) * v + 1.0) : 1.0 / ((0.5 * v - 1.0) * v + 1.0); }
static double tau_of_t(double t) { return t > 1.0 ? (sqrt(2.0 * t - 1.0) - 1.0) : (1.0 - sqrt(2.0 / t - 1.0)); }
static double dt_dtau(double v)
{
    if (v > 0) return v + 1.0;
    const double den = (0.5 * v - 1.0) * v + 1.0;
    return (1.0 - v) / (den * den);
}

/* mode 0: (pos, grad) -> 100 outside; 1: (pos, grad, mindis) -> 1e10 outside, gradient only when
 * dist <= mindis; 2: (pos) -> 1e10 outside.  grad may be NULL for mode 2. */
double be_esdf(const be_map *m, double x, double y, double grad[2], int mode, double mindis)
{
    const double out = mode == 0 ? 100.0 : 1e10, inv = 1.0 / m->res;
    if (x < m->x_lo || y < m->y_lo || x > m->x_hi || y > m->y_hi) {
        if (grad && mode != 2) grad[0] = grad[1] = 0.0;
        return out;
    }
    int ix = (int)((x - m->x_lo) * inv - 0.5), iy = (int)((y - m->y_lo) * inv - 0.5);
    ix = imin(imax(ix, 0), m->nx - 1);
    iy = imin(imax(iy, 0), m->ny - 1);
    if (ix >= m->nx - 1 || iy >= m->ny - 1) {
        if (grad && mode != 2) grad[0] = grad[1] = 0.0;
        return out;
    }
    const double fx = (x - ((ix + 0.5) * m->res + m->x_lo)) * inv, fy = (y - ((iy + 0.5) * m->res + m->y_lo)) * inv;
    const double v00 = m->dist[(size_t)ix * m->ny + iy], v01 = m->dist[(size_t)ix * m->ny + iy + 1];
    const double v10 = m->dist[(size_t)(ix + 1) * m->ny + iy], v11 = m->dist[(size_t)(ix + 1) * m->ny + iy + 1];
    const double lo = (1 - fx) * v00 + fx * v10, hi = (1 - fx) * v01 + fx * v11;
    const double dist = (1 - fy) * lo + fy * hi;
    if (mode == 2 || (mode == 1 && dist > mindis)) return dist;
    grad[1] = (hi - lo) * inv;
    grad[0] = ((1 - fy) * (v10 - v00) + fy * (v11 - v01)) * inv;
    return dist;
}
static double esdf_nearest(const be_map *m, double x, double y) /* getDistanceReal, sdf_map.cpp:865-871 */
{
    if (x < m->x_lo || y < m->y_lo || x > m->x_hi || y > m->y_hi) return 10000.0;
    const double inv = 1.0 / m->res;
    const int ix = imin(imax((int)((x - m->x_lo) * inv), 0), m->nx - 1), iy = imin(imax((int)((y - m->y_lo) * inv), 0), m->ny - 1);
    return m->dist[(size_t)ix * m->ny + iy];
}

void be_default_config(be_config *c)
{
    memset(c, 0, sizeof(*c));
    /* plan_manager/config/car3ms.yaml */
    c->max_vel = 3.0; c->min_vel = 0.0; c->max_acc = 2.0; c->max_omega = 3.0; c->max_domega = 4.0;
    c->max_cen_acc = 50.0; c->direct_v_omega = 0;
    c->n_check = 2;
    c->check_pts[0][0] = 0.3; c->check_pts[0][1] = 0.0; c->check_pts[1][0] = -0.3; c->check_pts[1][1] = 0.0;
    /* back_end/config/global_planning3ms.yaml */
    c->smooth_eps = 0.01;
    c->path_lbfgs = (be_lbfgs_param){256, 2, 8000, 64, 0.0, 5.0e-2, 0.0, 1.0e20, 1.0e-4, 0.9, 1.0e-6, 1.0e-16};
    c->shot_path_past = 8; c->shot_path_horizon = 0.5;
    c->p_time = 20; c->p_bigpath = 200000; c->p_mean_time = 100; c->p_moment = 1000; c->p_acc = 100; c->p_domega = 100;
    c->energy_w[0] = 0.33; c->energy_w[1] = 1.0;
    c->lbfgs = (be_lbfgs_param){256, 3, 8000, 64, 0.0, 5.0e-4, 1.0e-32, 1.0e20, 1.0e-4, 0.9, 1.0e-6, 1.0e-16};
    c->mean_lo = 0.5; c->mean_hi = 2.0;
    c->w_time = 50; c->w_acc = 300; c->w_domega = 300; c->w_collision = 500000; c->w_moment = 300; c->w_mean_time = 300;
    c->w_cen_acc = 300;
    for (int k = 0; k < 2; ++k) {
        c->lam0[k] = 0; c->rho0[k] = 1.0e4; c->rho_max[k] = 1.0e10; c->gamma[k] = 9.0;
        c->cut_lam0[k] = 0; c->cut_rho0[k] = 1.0e3; c->cut_rho_max[k] = 1.0e10; c->cut_gamma[k] = 5.0;
    }
    c->tol = 0.01; c->cut_tol = 0.5;
    c->sparse_res = 8;
    c->safe_dis = 0.6; c->final_min_safe_dis = 0.10; c->final_check_num = 16; c->safe_replan_max = 3;
    /* plan_manager/launch/planner_sim.launch:41-46 */
    c->icr_xv = 0.2; c->standard_diff = 1;
    c->max_alm_rounds = 64;
    c->exact_chain = 0;
}

/* ---- the penalty functional of both stages ---------------------------------------------------------- */
typedef struct {
    const be_config *c;
    const be_map *map;
    const be_problem *p;
    int stage;
    const double *lam, *rho;
    double safe_dis;
} pen_ctx;

static double penalty(const pen_ctx *k, const double *T, const double *coef, double *gdC, double *gdT, double xy_err[2],
                      double *xy_nodes)
{
    const be_config *c = k->c;
    const int M = k->p->M, R = c->sparse_res, S = 2 * R, NS = S + 1, NN = M * NS;
    const double xv = c->standard_diff ? 0.0 : c->icr_xv;
    const double w_mom = k->stage == 1 ? c->p_moment : c->w_moment, w_acc = k->stage == 1 ? c->p_acc : c->w_acc,
                 w_dom = k->stage == 1 ? c->p_domega : c->w_domega;
    /* d(dy)/d(theta) of the ICR model: the reference writes - theta' xv sin(theta) (optimizer.cpp:822, 990) where
     * the derivative of -theta' xv cos(theta) is + theta' xv sin(theta); kept unless exact_chain is set */
    const double ysign = c->exact_chain ? -1.0 : 1.0;
    double cost = 0.0;
    /* per node: d(dx)/d(c_s), d(dx)/d(c_theta) (6 each), d(dx)/dT, same for y; chain coefficients */
    double *XS = (double *)calloc((size_t)NN * 28 + (size_t)NN * 2, sizeof(double));
    double *XTh = XS + (size_t)NN * 6, *YS = XTh + (size_t)NN * 6, *YTh = YS + (size_t)NN * 6;
    double *XT = YTh + (size_t)NN * 6, *YT = XT + NN, *chX = YT + NN, *chY = chX + NN;
    double *pw = chY + NN; /* NN: Simpson weight pattern handled below */
    (void)pw;
    double cur[2] = {k->p->start_xy[0], k->p->start_xy[1]}; /* CurrentPointXY */
    double end_xy[2] = {k->p->start_xy[0], k->p->start_xy[1]};
    double *panel = (double *)malloc(sizeof(double) * 2 * R);

    for (int i = 0; i < M; ++i) {
        const double *ci = coef + 2 * 6 * i;
        const double step = T[i] / R, half = step / 2.0, cint = T[i] / (R * 6);
        for (int q = 0; q < 2 * R; ++q) panel[q] = 0.0;
        double s1 = 0.0;
        for (int j = 0; j <= S; ++j) {
            const int node = i * NS + j;
            const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s3 * s2;
            const double b0[6] = {1.0, s1, s2, s3, s4, s5};
            const double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
            const double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
            const double b3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
            s1 += half;
            double sg[2] = {0, 0}, d1[2] = {0, 0}, d2[2] = {0, 0}, d3[2] = {0, 0};
            for (int q = 0; q < 6; ++q)
                for (int d = 0; d < 2; ++d) {
                    sg[d] += ci[2 * q + d] * b0[q]; d1[d] += ci[2 * q + d] * b1[q];
                    d2[d] += ci[2 * q + d] * b2[q]; d3[d] += ci[2 * q + d] * b3[q];
                }
            const double cy = cos(sg[0]), sy = sin(sg[0]);
            const double ialpha = 1.0 / S * j;
            const double fx = d1[1] * cy + d1[0] * xv * sy, fy = d1[1] * sy - d1[0] * xv * cy;
            if (j % 2 == 0) {
                if (j != 0) { panel[2 * (j / 2 - 1)] += cint * fx; panel[2 * (j / 2 - 1) + 1] += cint * fy; }
                if (j != S) { panel[2 * (j / 2)] += cint * fx; panel[2 * (j / 2) + 1] += cint * fy; }
            } else {
                panel[2 * (j / 2)] += 4.0 * cint * fx;
                panel[2 * (j / 2) + 1] += 4.0 * cint * fy;
            }
            for (int q = 0; q < 6; ++q) {
                XS[node * 6 + q] = b1[q] * cy;
                XTh[node * 6 + q] = b0[q] * (-d1[1] * sy + d1[0] * xv * cy) + b1[q] * sy * xv;
                YS[node * 6 + q] = b1[q] * sy;
                YTh[node * 6 + q] = b0[q] * (d1[1] * cy - ysign * d1[0] * xv * sy) - b1[q] * cy * xv;
            }
            XT[node] = (d2[1] * cy - d1[1] * d1[0] * sy + d2[0] * xv * sy + d1[0] * d1[0] * xv * cy) * ialpha * cint + fx / (R * 6);
            YT[node] = (d2[1] * sy + d1[1] * d1[0] * cy - d2[0] * xv * cy + d1[0] * d1[0] * xv * sy) * ialpha * cint + fy / (R * 6);
            if (j % 2) continue;

            /* ---- even node: inequality penalties (weights: trapezoid in time) ---- */
            const double alpha = 1.0 / R * ((double)j / 2), omg = (j == 0 || j == S) ? 0.5 : 1.0, ws = omg * step;
            double gb[3][2] = {{0, 0}, {0, 0}, {0, 0}}; /* d cost / d (sigma, sigma', sigma'') */
            double f, df, v;
#define PENALISE(viol, weight, dviol_dt, apply)                                             \
    if ((v = (viol)) > 0.0) {                                                               \
        smoothed_l1(c->smooth_eps, v, &f, &df);                                             \
        apply;                                                                              \
        gdT[i] += omg * (weight) * (df * (dviol_dt) * step + f / R);                        \
        cost += ws * (weight) * f;                                                          \
    }
            if (k->stage == 2) {
                PENALISE(d2[1] * d2[1] - c->max_acc * c->max_acc, w_acc, 2.0 * alpha * d2[1] * d3[1],
                         gb[2][1] += ws * w_acc * df * 2.0 * d2[1]);
                PENALISE(d2[0] * d2[0] - c->max_domega * c->max_domega, w_dom, 2.0 * alpha * d2[0] * d3[0],
                         gb[2][0] += ws * w_dom * df * 2.0 * d2[0]);
            }
            if (k->stage == 2 && c->direct_v_omega) {
                PENALISE(d1[1] * d1[1] - c->max_vel * c->max_vel, w_mom, 2.0 * alpha * d1[1] * d2[1],
                         gb[1][1] += ws * w_mom * df * 2.0 * d1[1]);
                PENALISE(d1[0] * d1[0] - c->max_omega * c->max_omega, w_mom, 2.0 * alpha * d1[0] * d2[0],
                         gb[1][0] += ws * w_mom * df * 2.0 * d1[0]);
            } else {
                for (int sym = -1; sym <= 1; sym += 2)
                    PENALISE(sym * c->max_vel * d1[0] + c->max_omega * d1[1] - c->max_vel * c->max_omega, w_mom,
                             alpha * (sym * c->max_vel * d2[0] + c->max_omega * d2[1]),
                             (gb[1][0] += ws * w_mom * df * sym * c->max_vel, gb[1][1] += ws * w_mom * df * c->max_omega));
                for (int sym = -1; sym <= 1; sym += 2)
                    PENALISE(sym * -c->min_vel * d1[0] - c->max_omega * d1[1] + c->min_vel * c->max_omega, w_mom,
                             alpha * (sym * -c->min_vel * d2[0] - c->max_omega * d2[1]),
                             (gb[1][0] += ws * w_mom * df * sym * -c->min_vel, gb[1][1] -= ws * w_mom * df * c->max_omega));
            }
            if (k->stage == 1) { /* stage 1 takes acceleration limits after the moment rows */
                PENALISE(d2[1] * d2[1] - c->max_acc * c->max_acc, w_acc, 2.0 * alpha * d2[1] * d3[1],
                         gb[2][1] += ws * w_acc * df * 2.0 * d2[1]);
                PENALISE(d2[0] * d2[0] - c->max_domega * c->max_domega, w_dom, 2.0 * alpha * d2[0] * d3[0],
                         gb[2][0] += ws * w_dom * df * 2.0 * d2[0]);
            }
            if (k->stage == 2) {
                PENALISE(d1[0] * d1[0] * d1[1] * d1[1] - c->max_cen_acc * c->max_cen_acc, c->w_cen_acc,
                         2.0 * alpha * (d1[0] * d1[1] * d1[1] * d2[0] + d1[1] * d1[0] * d1[0] * d2[1]),
                         (gb[1][0] += ws * c->w_cen_acc * df * (2 * d1[0] * d1[1] * d1[1]),
                          gb[1][1] += ws * c->w_cen_acc * df * (2 * d1[0] * d1[0] * d1[1])));
                /* obstacle clearance of the body points at the integrated position */
                if (j != 0) { cur[0] += panel[2 * (j / 2 - 1)]; cur[1] += panel[2 * (j / 2 - 1) + 1]; }
                if (xy_nodes) { xy_nodes[2 * node] = cur[0]; xy_nodes[2 * node + 1] = cur[1]; }
                double gpos[2] = {0, 0};
                int hit = 0;
                for (int q = 0; q < c->n_check; ++q) {
                    const double px = c->check_pts[q][0], py = c->check_pts[q][1];
                    double ge[2] = {0, 0};
                    const double sd = be_esdf(k->map, cur[0] + cy * px - sy * py, cur[1] + sy * px + cy * py, ge, 1, k->safe_dis);
                    /* d(R(theta) cp)/d theta = (-sy px - cy py, cy px - sy py) */
                    const double rot = ge[0] * (-sy * px - cy * py) + ge[1] * (cy * px - sy * py);
                    PENALISE(k->safe_dis - sd, c->w_collision, -alpha * d1[0] * rot,
                             (hit = 1, gpos[0] -= ws * c->w_collision * df * ge[0], gpos[1] -= ws * c->w_collision * df * ge[1],
                              gb[0][0] -= ws * c->w_collision * df * rot));
                }
                if (hit) {
                    /* every node whose Simpson sample enters this position; the reference adds to all
                     * nodes up to and including this one */
                    const int upto = c->exact_chain ? node - 1 : node;
                    for (int q = 0; q <= upto; ++q) { chX[q] += gpos[0]; chY[q] += gpos[1]; }
                    if (c->exact_chain && j != 0) { /* the node closes one panel only: half of an interior weight */
                        const double wfix = (j == S) ? 1.0 : 0.5;
                        chX[node] += wfix * gpos[0];
                        chY[node] += wfix * gpos[1];
                    }
                }
            }
#undef PENALISE
            for (int q = 0; q < 6; ++q)
                for (int d = 0; d < 2; ++d) gdC[2 * (6 * i + q) + d] += b0[q] * gb[0][d] + b1[q] * gb[1][d] + b2[q] * gb[2][d];
        }
        double sx = 0.0, sy_ = 0.0;
        for (int q = 0; q < R; ++q) { sx += panel[2 * q]; sy_ += panel[2 * q + 1]; }
        end_xy[0] += sx;
        end_xy[1] += sy_;
        if (k->stage == 1) { /* way-point attraction at the end of every piece */
            const double ex = end_xy[0] - k->p->positions[2 * i], ey = end_xy[1] - k->p->positions[2 * i + 1];
            for (int q = 0; q < (i + 1) * NS; ++q) { chX[q] += c->p_bigpath * 2.0 * ex; chY[q] += c->p_bigpath * 2.0 * ey; }
            cost += c->p_bigpath * (ex * ex + ey * ey);
        }
        for (int j = 0; j <= S; ++j) { /* the stored derivative blocks carry the Simpson step */
            const int node = i * NS + j;
            for (int q = 0; q < 6; ++q) { XS[node * 6 + q] *= cint; XTh[node * 6 + q] *= cint; YS[node * 6 + q] *= cint; YTh[node * 6 + q] *= cint; }
        }
    }
    xy_err[0] = end_xy[0] - k->p->final_xy[0];
    xy_err[1] = end_xy[1] - k->p->final_xy[1];
    if (k->stage == 2) { /* augmented-Lagrangian terminal position */
        const double ax = xy_err[0] + k->lam[0] / k->rho[0], ay = xy_err[1] + k->lam[1] / k->rho[1];
        cost += 0.5 * (k->rho[0] * ax * ax + k->rho[1] * ay * ay);
        for (int q = 0; q < NN; ++q) { chX[q] += k->rho[0] * ax; chY[q] += k->rho[1] * ay; }
    }
    /* chain rule through the Simpson sums (weights 1 4 2 ... 2 4 1 inside each piece) */
    for (int i = 0; i < M; ++i)
        for (int j = 0; j <= S; ++j) {
            const int node = i * NS + j;
            const double sw = (j == 0 || j == S) ? 1.0 : ((j % 2) ? 4.0 : 2.0);
            const double cx = chX[node] * sw, cyy = chY[node] * sw;
            for (int q = 0; q < 6; ++q) {
                gdC[2 * (6 * i + q) + 1] += XS[node * 6 + q] * cx + YS[node * 6 + q] * cyy;
                gdC[2 * (6 * i + q) + 0] += XTh[node * 6 + q] * cx + YTh[node * 6 + q] * cyy;
            }
            gdT[i] += XT[node] * cx + YT[node] * cyy;
        }
    free(panel);
    free(XS);
    return cost;
}

double be_eval(const be_config *c, const be_map *map, const be_problem *p, int stage, const double *x, double *g,
               const double lam[2], const double rho[2], double safe_dis, double time_weight, double xy_err[2], double *xy_nodes)
{
    const int M = p->M, n = 3 * M - 1;
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) nrm += x[i] * x[i];
    if (sqrt(nrm) > 1e4) return 0.0; /* `inf` macro = 0; gradient left as it is */
    double *T = (double *)malloc(sizeof(double) * (M + 6 * M * 2 * 2 + M + M + 2 * M));
    double *coef = T + M, *gdC = coef + 12 * M, *gdT = gdC + 12 * M, *gT = gdT + M, *gP = gT + M;
    double tail[2][3];
    memcpy(tail, p->tail, sizeof(tail));
    tail[1][0] = x[2 * (M - 1)];
    const double *tau = x + 2 * (M - 1) + 1;
    double sumT = 0.0;
    for (int i = 0; i < M; ++i) { T[i] = t_of_tau(tau[i]); sumT += T[i]; }
    be_spline(M, T, x, p->head, tail, coef);
    double cost = be_energy(M, T, coef, c->energy_w, gdC, gdT);
    pen_ctx k = {c, map, p, stage, lam, rho, safe_dis};
    cost += penalty(&k, T, coef, gdC, gdT, xy_err, xy_nodes);
    double gtail[2];
    be_spline_adjoint(M, T, coef, gdC, gdT, gP, gT, gtail);
    cost += (stage == 1 ? c->p_time : time_weight) * sumT;
    for (int i = 0; i < M; ++i) g[2 * (M - 1) + 1 + i] = (gT[i] + time_weight) * dt_dtau(tau[i]);
    for (int i = 0; i < 2 * (M - 1); ++i) g[i] = gP[i];
    g[2 * (M - 1)] = gtail[1];
    free(T);
    return cost;
}

/* ---- L-BFGS with the Lewis-Overton line search --------------------------------------------------- */
enum { LB_CONVERGENCE = 0, LB_STOP = 1, LB_CANCELED = 2, LBE_INVALID_FUNCVAL = -1012, LBE_MINIMUMSTEP = -1011,
       LBE_MAXIMUMSTEP = -1010, LBE_MAXIMUMLINESEARCH = -1009, LBE_MAXIMUMITERATION = -1008, LBE_WIDTHTOOSMALL = -1007,
       LBE_INVALIDPARAMETERS = -1006, LBE_INCREASEGRADIENT = -1005 };

typedef struct {
    const be_config *c;
    const be_map *map;
    const be_problem *p;
    int stage;
    const double *lam, *rho;
    double safe_dis, time_weight;
    double xy_err[2];
    int evals;
} opt_ctx;

static double call_cost(opt_ctx *o, const double *x, double *g)
{
    ++o->evals;
    double e[2];
    const double f = be_eval(o->c, o->map, o->p, o->stage, x, g, o->lam, o->rho, o->safe_dis, o->time_weight, e, NULL);
    double nrm = 0.0;
    for (int i = 0; i < 3 * o->p->M - 1; ++i) nrm += x[i] * x[i];
    if (!(sqrt(nrm) > 1e4)) { o->xy_err[0] = e[0]; o->xy_err[1] = e[1]; }
    return f;
}
static double dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
static double maxabs(const double *a, int n) { double m = 0; for (int i = 0; i < n; ++i) if (fabs(a[i]) > m) m = fabs(a[i]); return m; }

static int search(opt_ctx *o, int n, double *x, double *f, double *g, double *stp, const double *s, const double *xp,
                  const double *gp, double stpmin, double stpmax, const be_lbfgs_param *pr)
{
    int count = 0, brackt = 0, touched = 0;
    double mu = 0.0, nu = stpmax;
    if (!(*stp > 0.0)) return LBE_INVALIDPARAMETERS;
    const double dginit = dot(gp, s, n);
    if (0.0 < dginit) return LBE_INCREASEGRADIENT;
    const double finit = *f, dgtest = pr->f_dec_coeff * dginit, dstest = pr->s_curv_coeff * dginit;
    for (;;) {
        for (int i = 0; i < n; ++i) x[i] = xp[i] + *stp * s[i];
        *f = call_cost(o, x, g);
        ++count;
        if (isinf(*f) || isnan(*f)) return LBE_INVALID_FUNCVAL;
        if (pr->past > 0 && fabs(finit - *f) / (fabs(finit) + 1.0) < pr->delta / pr->past) return count;
        if (*f > finit + *stp * dgtest) {
            nu = *stp;
            brackt = 1;
        } else if (dot(g, s, n) < dstest) {
            mu = *stp;
        } else {
            return count;
        }
        if (pr->max_linesearch <= count) return LBE_MAXIMUMLINESEARCH;
        if (brackt && (nu - mu) < pr->machine_prec * nu) return LBE_WIDTHTOOSMALL;
        *stp = brackt ? 0.5 * (mu + nu) : *stp * 2.0;
        if (*stp < stpmin) return LBE_MINIMUMSTEP;
        if (*stp > stpmax) {
            if (touched) return LBE_MAXIMUMSTEP;
            touched = 1;
            *stp = stpmax;
        }
    }
}

static int lbfgs(opt_ctx *o, int n, double *x, double *f_out, const be_lbfgs_param *pr, int iter_cap, int *n_iter)
{
    const int m = pr->mem_size, npast = pr->past > 1 ? pr->past : 1;
    double *xp = (double *)calloc((size_t)4 * n + npast + 2 * m + 2 * (size_t)n * m, sizeof(double));
    double *g = xp + n, *gp = g + n, *d = gp + n, *pf = d + n, *alpha = pf + npast, *ys_ = alpha + m;
    double *S = ys_ + m, *Y = S + (size_t)n * m;
    int ret, k = 1, end = 0, bound = 0;
    double fx = call_cost(o, x, g);
    pf[0] = fx;
    for (int i = 0; i < n; ++i) d[i] = -g[i];
    double gn = maxabs(g, n), xn = maxabs(x, n);
    if (gn / (xn > 1.0 ? xn : 1.0) < pr->g_epsilon) {
        ret = LB_CONVERGENCE;
    } else {
        double step = 1.0 / sqrt(dot(d, d, n));
        for (;;) {
            memcpy(xp, x, sizeof(double) * n);
            memcpy(gp, g, sizeof(double) * n);
            const int ls = search(o, n, x, &fx, g, &step, d, xp, gp, pr->min_step, pr->max_step, pr);
            if (ls < 0) {
                memcpy(x, xp, sizeof(double) * n);
                memcpy(g, gp, sizeof(double) * n);
                ret = ls;
                break;
            }
            gn = maxabs(g, n);
            xn = maxabs(x, n);
            if (gn / (xn > 1.0 ? xn : 1.0) < pr->g_epsilon) { ret = LB_CONVERGENCE; break; }
            if (pr->past > 0) {
                if (pr->past <= k) {
                    const double rate = fabs(pf[k % pr->past] - fx) / (fabs(fx) > 1.0 ? fabs(fx) : 1.0);
                    if (rate < pr->delta) { ret = LB_STOP; break; }
                }
                pf[k % pr->past] = fx;
            }
            if ((pr->max_iterations != 0 && pr->max_iterations <= k) || (iter_cap > 0 && iter_cap <= k)) {
                ret = LBE_MAXIMUMITERATION;
                break;
            }
            ++k;
            double *sk = S + (size_t)end * n, *yk = Y + (size_t)end * n;
            for (int i = 0; i < n; ++i) { sk[i] = x[i] - xp[i]; yk[i] = g[i] - gp[i]; }
            const double ys = dot(yk, sk, n), yy = dot(yk, yk, n);
            ys_[end] = ys;
            for (int i = 0; i < n; ++i) d[i] = -g[i];
            const double cau = dot(sk, sk, n) * sqrt(dot(gp, gp, n)) * pr->cautious_factor;
            if (ys > cau) {
                ++bound;
                if (bound > m) bound = m;
                end = (end + 1) % m;
                int j = end;
                for (int i = 0; i < bound; ++i) {
                    j = (j + m - 1) % m;
                    alpha[j] = dot(S + (size_t)j * n, d, n) / ys_[j];
                    for (int q = 0; q < n; ++q) d[q] += (-alpha[j]) * Y[(size_t)j * n + q];
                }
                for (int q = 0; q < n; ++q) d[q] *= ys / yy;
                for (int i = 0; i < bound; ++i) {
                    const double beta = dot(Y + (size_t)j * n, d, n) / ys_[j];
                    for (int q = 0; q < n; ++q) d[q] += (alpha[j] - beta) * S[(size_t)j * n + q];
                    j = (j + 1) % m;
                }
            }
            step = 1.0;
        }
    }
    *f_out = fx;
    if (n_iter) *n_iter = k;
    free(xp);
    return ret;
}

int be_lbfgs_run(const be_config *c, const be_map *map, const be_problem *p, int stage, double *x, double *cost,
                 const double lam[2], const double rho[2], double safe_dis, double time_weight, int max_iter, int *n_iter,
                 int *n_eval, double xy_err[2])
{
    opt_ctx o = {c, map, p, stage, lam, rho, safe_dis, time_weight, {0, 0}, 0};
    be_lbfgs_param pr = stage == 1 ? c->path_lbfgs : c->lbfgs;
    if (stage == 1 && fabs(p->tail[1][0]) < c->shot_path_horizon) pr.past = c->shot_path_past;
    const int ret = lbfgs(&o, 3 * p->M - 1, x, cost, &pr, max_iter, n_iter);
    if (n_eval) *n_eval = o.evals;
    if (xy_err) { xy_err[0] = o.xy_err[0]; xy_err[1] = o.xy_err[1]; }
    return ret;
}

int be_optimize(const be_config *c, const be_map *map, const be_problem *p, double safe_dis, double time_weight, be_result *r)
{
    const int M = p->M, n = 3 * M - 1;
    double *x = (double *)malloc(sizeof(double) * n);
    double lam[2], rho[2];
    const double *rmax = p->if_cut ? c->cut_rho_max : c->rho_max, *gam = p->if_cut ? c->cut_gamma : c->gamma;
    const double tol = p->if_cut ? c->cut_tol : c->tol;
    for (int q = 0; q < 2; ++q) { lam[q] = p->if_cut ? c->cut_lam0[q] : c->lam0[q]; rho[q] = p->if_cut ? c->cut_rho0[q] : c->rho0[q]; }
    memcpy(x, p->inner, sizeof(double) * 2 * (M - 1));
    x[2 * (M - 1)] = p->tail[1][0];
    for (int i = 0; i < M; ++i) x[2 * (M - 1) + 1 + i] = tau_of_t(p->init_T);
    opt_ctx o = {c, map, p, 1, lam, rho, safe_dis, time_weight, {0, 0}, 0};
    be_lbfgs_param pp = c->path_lbfgs;
    if (fabs(p->tail[1][0]) < c->shot_path_horizon) pp.past = c->shot_path_past;
    double cost = 0.0;
    r->path_ret = lbfgs(&o, n, x, &cost, &pp, 0, NULL);
    o.stage = 2;
    r->alm_rounds = 0;
    for (;;) {
        r->lbfgs_ret = lbfgs(&o, n, x, &cost, &c->lbfgs, 0, NULL);
        ++r->alm_rounds;
        if (sqrt(o.xy_err[0] * o.xy_err[0] + o.xy_err[1] * o.xy_err[1]) < tol) break;
        if (r->alm_rounds >= c->max_alm_rounds) break;
        for (int q = 0; q < 2; ++q) {
            lam[q] += rho[q] * o.xy_err[q];
            rho[q] = fmin((1 + gam[q]) * rho[q], rmax[q]);
        }
    }
    r->cost = cost;
    r->evals = o.evals;
    r->xy_err[0] = o.xy_err[0];
    r->xy_err[1] = o.xy_err[1];
    r->tail_s = x[2 * (M - 1)];
    memcpy(r->inner, x, sizeof(double) * 2 * (M - 1));
    double tail[2][3];
    memcpy(tail, p->tail, sizeof(tail));
    tail[1][0] = r->tail_s;
    for (int i = 0; i < M; ++i) r->T[i] = t_of_tau(x[2 * (M - 1) + 1 + i]);
    be_spline(M, r->T, r->inner, p->head, tail, r->coef);
    free(x);
    return 0;
}

/* dense Simpson resample of the final trajectory against the ESDF */
int be_final_collision(const be_config *c, const be_map *map, int M, const double *T, const double *coef,
                       const double start_xy[2], double *min_dist)
{
    const int R = c->final_check_num, S = 2 * R;
    const double xv = c->standard_diff ? 0.0 : c->icr_xv;
    double pos[2] = {start_xy[0], start_xy[1]}, mind = 1.79769313486231570815e+308;
    double *inc = (double *)malloc(sizeof(double) * 2 * R * M);
    for (int i = 0; i < M; ++i) {
        const double half = T[i] / R / 2.0, cint = T[i] / R / 6.0;
        const double *ci = coef + 12 * i;
        double *pi = inc + 2 * R * i;
        for (int q = 0; q < 2 * R; ++q) pi[q] = 0.0;
        double s1 = 0.0;
        for (int j = 0; j <= S; ++j) {
            double sg[2] = {0, 0}, d1[2] = {0, 0}, tn = 1.0, tv = 1.0;
            for (int q = 0; q < 6; ++q) {
                for (int d = 0; d < 2; ++d) {
                    sg[d] += ci[2 * q + d] * tn;
                    if (q >= 1) d1[d] += q * ci[2 * q + d] * tv;
                }
                if (q >= 1) tv *= s1;
                tn *= s1;
            }
            s1 += half;
            const double fx = d1[1] * cos(sg[0]) + d1[0] * xv * sin(sg[0]), fy = d1[1] * sin(sg[0]) - d1[0] * xv * cos(sg[0]);
            if (j % 2 == 0) {
                if (j != 0) { pi[2 * (j / 2 - 1)] += cint * fx; pi[2 * (j / 2 - 1) + 1] += cint * fy; }
                if (j != S) { pi[2 * (j / 2)] += cint * fx; pi[2 * (j / 2) + 1] += cint * fy; }
            } else {
                pi[2 * (j / 2)] += 4.0 * cint * fx;
                pi[2 * (j / 2) + 1] += 4.0 * cint * fy;
            }
        }
    }
    int hit = 0;
    for (int q = 0; q < R * M && !hit; ++q) {
        pos[0] += inc[2 * q];
        pos[1] += inc[2 * q + 1];
        const double sd = be_esdf(map, pos[0], pos[1], NULL, 2, 0.0);
        if (sd < mind) mind = sd;
        if (sd < c->final_min_safe_dis) hit = 1;
    }
    free(inc);
    if (min_dist) *min_dist = mind;
    return hit;
}

int be_minco_plan(const be_config *c, const be_map *map, const be_problem *p, be_result *r)
{
    const double start_safe = esdf_nearest(map, p->start_xy[0], p->start_xy[1]) * 0.85;
    const double safe = start_safe < c->safe_dis ? start_safe : c->safe_dis;
    double tw = c->w_time;
    int evals = 0;
    r->attempts = 0;
    r->collision = 1;
    while (r->attempts < c->safe_replan_max) {
        be_optimize(c, map, p, safe, tw, r);
        evals += r->evals;
        ++r->attempts;
        r->collision = be_final_collision(c, map, p->M, r->T, r->coef, p->start_xy, &r->min_dist);
        if (!r->collision) break;
        tw *= 0.75;
    }
    r->evals = evals;
    return r->collision ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* ESDF construction: SDFmap::updateESDF2d + fillESDF (P/utils/plan_env/src/sdf_map.cpp:618-714).                      */
/* Two-pass Felzenszwalb distance transform (rows, then columns) of the occupied cells (positive part) and of the      */
/* non-occupied cells (negative part) inside the window odom +- detection_range, combined as pos - neg + res inside     */
/* obstacles.  The reference indexes its (X+1) x (Y+1) scratch buffers with stride Y: the last element of a row is the   */
/* first of the next.  Executed sequentially that has a defined outcome, restated here statement by statement (the     */
/* combine step leaves the last row and column of the window untouched, :671-679).                                     */
/* grid states: 0 Unknown, 1 Unoccupied, 2 Occupied (sdf_map.h:98)                                                     */
/* ------------------------------------------------------------------------------------------------------------------ */
#include <float.h>
#include <math.h>
#include <stdlib.h>

static void fill_esdf(const double *in, int in_stride, double *out, int out_stride, int end, double scale_sqrt, int *v, double *z)
{
    /* f_get_val(q) = in[q * in_stride];  f_set_val(q, val): out[q * out_stride] = scale_sqrt > 0 ? scale_sqrt * sqrt(val) : val */
    int k = 0, q;
    v[0] = 0;
    z[0] = -DBL_MAX;
    z[1] = DBL_MAX;
    for (q = 1; q <= end; q++) {
        double s;
        k++;
        do {
            k--;
            s = ((in[q * in_stride] + q * q) - (in[v[k] * in_stride] + v[k] * v[k])) / (2 * q - 2 * v[k]);
        } while (s <= z[k]);
        k++;
        v[k] = q;
        z[k] = s;
        z[k + 1] = DBL_MAX;
    }
    k = 0;
    for (q = 0; q <= end; q++) {
        double val;
        while (z[k + 1] < q) k++;
        val = (q - v[k]) * (q - v[k]) + in[v[k] * in_stride];
        out[q * out_stride] = scale_sqrt > 0.0 ? scale_sqrt * sqrt(val) : val;
    }
}

/* dist_all [GLX][GLY] is updated in place inside the window (cells outside keep their values; the reference starts from
 * DBL_MAX everywhere, sdf_map.h:160).  Returns 0, or -1 when the window is empty. */
int be_update_esdf2d(const unsigned char *grid, int GLX, int GLY, double res, double x_lo, double y_lo, double odom_x, double odom_y,
                     double range, double *dist_all)
{
    const double inv = 1.0 / res, gx_hi = x_lo + GLX * res, gy_hi = y_lo + GLY * res;
    const int min_x = (int)floor(fmax(0.0, odom_x - range - x_lo) * inv), min_y = (int)floor(fmax(0.0, odom_y - range - y_lo) * inv);
    const int max_x = (int)ceil(fmin(gx_hi - x_lo, odom_x + range - x_lo) * inv) - 1;
    const int max_y = (int)ceil(fmin(gy_hi - y_lo, odom_y + range - y_lo) * inv) - 1;
    const int X = max_x - min_x, Y = max_y - min_y;
    int x, y, pass;
    size_t total;
    double *tmp, *pos, *neg, *line_in, *z;
    int *v;
    if (X < 1 || Y < 1) return -1;
    total = (size_t)(X + 1) * (Y + 1);
    tmp = calloc(total, sizeof(double)); pos = calloc(total, sizeof(double)); neg = calloc(total, sizeof(double));
    line_in = malloc(sizeof(double) * (size_t)((X > Y ? X : Y) + 2));
    z = malloc(sizeof(double) * (size_t)((X > Y ? X : Y) + 3));
    v = malloc(sizeof(int) * (size_t)((X > Y ? X : Y) + 2));
    for (pass = 0; pass < 2; ++pass) {
        double *dst = pass == 0 ? pos : neg;
        for (x = 0; x <= X; x++) {
            for (y = 0; y <= Y; y++) {
                const int st = grid[(size_t)(x + min_x) * GLY + (y + min_y)];
                const int seed = pass == 0 ? (st == 2) : (st == 1 || st == 0);
                line_in[y] = seed ? 0.0 : DBL_MAX;
            }
            fill_esdf(line_in, 1, tmp + (size_t)x * Y, 1, Y, 0.0, v, z);
        }
        for (y = 0; y <= Y; y++) fill_esdf(tmp + y, Y, dst + y, Y, X, res, v, z);
    }
    for (x = 0; x < X; x++)
        for (y = 0; y < Y; y++) {
            const size_t g = (size_t)(x + min_x) * GLY + y + min_y, i = (size_t)x * Y + y;
            dist_all[g] = pos[i];
            if (neg[i] > 0.0) dist_all[g] += (-neg[i] + res);
        }
    free(tmp); free(pos); free(neg); free(line_in); free(z); free(v);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* MSPlanner::get_the_predicted_state / get_the_predicted_state_and_path (optimizer.cpp:1108-1188, 1190-1262): the pose   */
/* reached at `time` by Simpson integration of the flat trajectory in steps of trajPredictResolution from start_time,   */
/* and the flat derivatives there.  coef [(6 i + k) * 2 + d] (ascending powers), T [M].                                 */
/* ------------------------------------------------------------------------------------------------------------------ */
static int traj_locate(const double *T, int M, double *t)
{ /* Trajectory::locatePieceIdx (trajectory.hpp:472-490) */
    int idx;
    double d = 0.0;
    for (idx = 0; idx < M && *t > (d = T[idx]); ++idx) *t -= d;
    if (idx == M) { --idx; *t += T[idx]; }
    return idx;
}
static void traj_eval(const double *T, const double *coef, int M, double t, double p[2], double v[2], double a[2], double j[2])
{
    double tl = t;
    const int i = traj_locate(T, M, &tl);
    int d;
    for (d = 0; d < 2; ++d) {
        const double c0 = coef[(6 * i + 0) * 2 + d], c1 = coef[(6 * i + 1) * 2 + d], c2 = coef[(6 * i + 2) * 2 + d];
        const double c3 = coef[(6 * i + 3) * 2 + d], c4 = coef[(6 * i + 4) * 2 + d], c5 = coef[(6 * i + 5) * 2 + d];
        if (p) p[d] = ((((c5 * tl + c4) * tl + c3) * tl + c2) * tl + c1) * tl + c0;
        if (v) v[d] = (((5.0 * c5 * tl + 4.0 * c4) * tl + 3.0 * c3) * tl + 2.0 * c2) * tl + c1;
        if (a) a[d] = ((20.0 * c5 * tl + 12.0 * c4) * tl + 6.0 * c3) * tl + 2.0 * c2;
        if (j) j[d] = (60.0 * c5 * tl + 24.0 * c4) * tl + 6.0 * c3;
    }
}
static void simpson_add(int standard_diff, double xv, double w6, const double p1[2], const double v1[2], const double p2[2], const double v2[2],
                        const double p3[2], const double v3[2], double xyt[3])
{
    if (standard_diff) {
        xyt[0] += w6 * (v1[1] * cos(p1[0]) + 4.0 * v2[1] * cos(p2[0]) + v3[1] * cos(p3[0]));
        xyt[1] += w6 * (v1[1] * sin(p1[0]) + 4.0 * v2[1] * sin(p2[0]) + v3[1] * sin(p3[0]));
    } else {
        const double x1 = v1[1] * cos(p1[0]) + v1[0] * xv * sin(p1[0]), x2 = v2[1] * cos(p2[0]) + v2[0] * xv * sin(p2[0]),
                     x3 = v3[1] * cos(p3[0]) + v3[0] * xv * sin(p3[0]);
        const double y1 = v1[1] * sin(p1[0]) - v1[0] * xv * cos(p1[0]), y2 = v2[1] * sin(p2[0]) - v2[0] * xv * cos(p2[0]),
                     y3 = v3[1] * sin(p3[0]) - v3[0] * xv * cos(p3[0]);
        xyt[0] += w6 * (x1 + 4.0 * x2 + x3);
        xyt[1] += w6 * (y1 + 4.0 * y2 + y3);
    }
    xyt[2] = p3[0];
}
/* returns if_forward of the _and_path variant; xyt in: start pose, out: predicted pose */
int be_predicted_state(const double *T, const double *coef, int M, int standard_diff, double xv, double step, double start_time, double time,
                       double xyt[3], double vaj[3], double oaj[3])
{
    double total = 0.0, check = time, p1[2], v1[2], p2[2], v2[2], p3[2], v3[2], a3[2], j3[2], pe[2], ps[2];
    int i, n;
    double left;
    for (i = 0; i < M; ++i) total += T[i];
    if (time > total) check = total;
    n = (int)floor((check - start_time) / step);
    left = check - n * step - start_time;
    traj_eval(T, coef, M, start_time, p3, v3, 0, 0);
    for (i = 0; i < n; ++i) {
        p1[0] = p3[0]; p1[1] = p3[1]; v1[0] = v3[0]; v1[1] = v3[1];
        traj_eval(T, coef, M, start_time + i * step + step / 2.0, p2, v2, 0, 0);
        traj_eval(T, coef, M, start_time + i * step + step, p3, v3, 0, 0);
        simpson_add(standard_diff, xv, step / 6.0, p1, v1, p2, v2, p3, v3, xyt);
    }
    p1[0] = p3[0]; p1[1] = p3[1]; v1[0] = v3[0]; v1[1] = v3[1];
    traj_eval(T, coef, M, check - left / 2.0, p2, v2, 0, 0);
    traj_eval(T, coef, M, check, p3, v3, a3, j3);
    simpson_add(standard_diff, xv, left / 6.0, p1, v1, p2, v2, p3, v3, xyt);
    oaj[0] = v3[0]; oaj[1] = a3[0]; oaj[2] = j3[0];
    vaj[0] = v3[1]; vaj[1] = a3[1]; vaj[2] = j3[1];
    traj_eval(T, coef, M, time, pe, 0, 0, 0);
    traj_eval(T, coef, M, start_time, ps, 0, 0, 0);
    return pe[1] - ps[1] > 0.0 ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* MSPlanner::mincoPointPub (optimizer.cpp:1714-1826): the optimised trajectory as the marker points the ALORE FSM      */
/* follows.  Per piece, `res` Simpson panels of the planar velocity (the same even / odd node walk and the same order of  */
/* additions as the reference: start term, 4 x mid term, end term), accumulated from the start position; the last point  */
/* of every piece is pushed twice.  res = (int)(sparseResolution_ * 0.4).  xy [M (res + 1)][2], yaw [M res] (the         */
/* reference fills `Yaw` and does not publish it) or NULL.  Returns the number of points.                                */
/* ------------------------------------------------------------------------------------------------------------------ */
int be_path_points(const double *T, const double *coef, int M, int standard_diff, double xv, int res, const double start_xy[2],
                   double *xy, double *yaw)
{
    double sumT = 0.0, px = start_xy[0], py = start_xy[1];
    int i, j, n = 0;
    double *ix = (double *)calloc((size_t)(res > 0 ? res : 1), sizeof(double)), *iy = (double *)calloc((size_t)(res > 0 ? res : 1), sizeof(double));
    for (i = 0; i < M; ++i) {
        const double step = T[i] / res, halfstep = step / 2.0, C = T[i] / res / 6.0;
        double s1 = 0.0;
        for (j = 0; j < res; ++j) ix[j] = iy[j] = 0.0;
        for (j = 0; j <= 2 * res; ++j) {
            double p[2], v[2], fx, fy, cy, sy;
            traj_eval(T, coef, M, s1 + sumT, p, v, 0, 0);
            s1 += halfstep;
            cy = cos(p[0]); sy = sin(p[0]);
            if (j % 2 == 0) {
                if (standard_diff) { fx = C * v[1] * cy; fy = C * v[1] * sy; }
                else { fx = C * (v[1] * cy + v[0] * xv * sy); fy = C * (v[1] * sy - v[0] * xv * cy); }
                if (j != 0) { ix[j / 2 - 1] += fx; iy[j / 2 - 1] += fy; if (yaw) yaw[i * res + j / 2 - 1] = p[0]; }
                if (j != 2 * res) { ix[j / 2] += fx; iy[j / 2] += fy; }
            } else {
                if (standard_diff) { fx = 4.0 * C * v[1] * cy; fy = 4.0 * C * v[1] * sy; }
                else { fx = 4 * C * (v[1] * cy + v[0] * xv * sy); fy = 4 * C * (v[1] * sy - v[0] * xv * cy); }
                ix[j / 2] += fx; iy[j / 2] += fy;
            }
        }
        for (j = 0; j < res; ++j) {
            px += ix[j]; py += iy[j];
            xy[2 * n] = px; xy[2 * n + 1] = py; ++n;
            if (j == res - 1) { xy[2 * n] = px; xy[2 * n + 1] = py; ++n; }
        }
        sumT += T[i];
    }
    free(ix); free(iy);
    return n;
}
