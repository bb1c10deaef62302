/*
 * oracle/nmpc_oracle.c -- CPU restatement of the reference NMPC numerics.
 *
 * TEST INFRASTRUCTURE ONLY (see nmpc_oracle.h).  Horizon-generic float32
 * restatement of the reference's ACADO real-time-iteration solver for the
 * planar differential-drive (ICR skid-steer) model, and of the box-QP path of
 * its vendored qpOASES.  Every function cites the reference lines it follows;
 * paths are relative to /root/reference/planning_ddr_opt/nmpc_controller/ with
 *   CG  = UAV_CAR_model/build/quadrotor_mpc_codegen
 *   QPO = externals/qpoases
 * The reference is fully unrolled generated code for N = 50; here the same
 * arithmetic is written as loops over the horizon, keeping the reference's
 * operation order so that at N = 50 the intermediate results agree to the last
 * bits (tests/test_oracle_golden.py pins this against oracle/_ref and the
 * committed fixtures).
 */
#include "nmpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef float real_t; /* CG/acado_qpoases_interface.hpp:50 */

enum { NX = ORC_NX, NU = ORC_NU, NOD = ORC_NOD, NY = ORC_NY, NYN = ORC_NYN };

/* qpOASES constants, QPO/INCLUDE/Constants.hpp:64-95 with the overrides of
 * CG/acado_qpoases_interface.hpp:44-48.  NB: ZERO = (float)1e-50 underflows to
 * exactly 0.0f, so "> ZERO" is "> 0". */
#define QP_EPS ((real_t)1.193e-07)
#define QP_ZERO ((real_t)1.0e-50)
#define QP_INFTY ((real_t)1.0e12)
#define QP_BOUNDTOL ((real_t)1.0e-10)
#define QP_BOUNDRELAXATION ((real_t)1.0e3)
#define QP_NWSRMAX 300

/* qpOASES returnValue numbers (QPO/INCLUDE/MessageHandling.hpp:64-170) */
enum {
    RET_OK = 0,
    RET_DIV_BY_ZERO = 1,
    RET_INVALID_ARGUMENTS = 3,
    RET_INIT_FAILED = 29,
    RET_INIT_FAILED_CHOLESKY = 31,
    RET_INIT_FAILED_HOTSTART = 32,
    RET_INIT_FAILED_INFEASIBILITY = 33,
    RET_QP_INFEASIBLE = 41,
    RET_STEPDIRECTION_DETERMINATION_FAILED = 51,
    RET_OPTIMAL_SOLUTION_FOUND = 53,
    RET_HOTSTART_STOPPED_INFEASIBILITY = 55,
    RET_MAX_NWSR_REACHED = 58,
    RET_STEPDIRECTION_FAILED_CHOLESKY = 63,
    RET_REMOVE_FROM_ACTIVESET_FAILED = 78,
    RET_ADD_TO_ACTIVESET_FAILED = 79,
    RET_HESSIAN_NOT_SPD = 93
};

struct orc_nmpc {
    int N;
    real_t Ah[4]; /* dt * Butcher matrix, row = stage (CG/acado_integrator.c:255-257) */
    real_t hb;    /* dt * b_s = dt/2 for both stages (CG/acado_integrator.c:393) */
    /* ACADOvariables (CG/acado_common.h:104-162) */
    real_t *x, *u, *od, *y, *yN, *W, *WN, *x0, *lbValues, *ubValues;
    /* ACADOworkspace (CG/acado_common.h:170-257) */
    real_t *d, *Dy, *DyN, *evGx, *evGu, *Q1, *Q2, *R1, *R2, *QN1, *QN2, *sbar, *Dx0, *E, *QDy, *H,
        *g, *lb, *ub, *dx, *dual;
    /* integrator memory: rk_kkk is a never-reset file-scope global in the
     * reference (CG/acado_integrator.c:37); here it lives in the handle */
    real_t rk_kkk[6];
    real_t rk_A[36];
    int rk_perm[6];
    int nWSR;
    real_t *pool;
};

/* ------------------------------------------------------------------------- */
/* A1  model: CG/acado_integrator.c:62-123 (spec UAV_CAR_model.cpp:38-40)      */
/* ------------------------------------------------------------------------- */

/* in = [x y psi | vr vl | xv yr yl] */
void orc_rhs(const real_t *in, real_t *out)
{
    const real_t *xd = in, *u = in + 3, *od = in + 5;
    /* libm double sin/cos of the float heading, rounded back to float */
    const real_t c = (real_t)cos(xd[2]);
    const real_t s = (real_t)sin(xd[2]);
    const real_t den = od[2] - od[1];
    const real_t lon = ((u[0] * od[2]) - (u[1] * od[1])) / den; /* body-x speed */
    const real_t lat = ((u[0] - u[1]) * od[0]) / den;           /* ICR slip term */
    out[0] = (lon * c) + (lat * s);
    out[1] = (lon * s) - (lat * c);
    out[2] = (u[0] - u[1]) / den;
}

/* 3 x 5 Jacobian wrt (x, y, psi, vr, vl), row-major */
void orc_diffs(const real_t *in, real_t *out)
{
    const real_t *xd = in, *u = in + 3, *od = in + 5;
    const real_t c = (real_t)cos(xd[2]);
    const real_t s = (real_t)sin(xd[2]);
    const real_t ms = (real_t)((real_t)(-1.0) * sin(xd[2]));
    const real_t den = od[2] - od[1];
    const real_t inv = (real_t)1.0 / den;
    const real_t lon = ((u[0] * od[2]) - (u[1] * od[1])) / den;
    const real_t lat = ((u[0] - u[1]) * od[0]) / den;
    const real_t mod1 = (real_t)0.0 - od[1];
    const real_t mod0 = ((real_t)0.0 - (real_t)1.0) * od[0];
    int i;
    for (i = 0; i < 15; ++i) out[i] = (real_t)0.0;
    out[2] = (lon * ms) + (lat * c);
    out[3] = ((od[2] * inv) * c) + ((od[0] * inv) * s);
    out[4] = ((mod1 * inv) * c) + ((mod0 * inv) * s);
    out[7] = (lon * c) - (lat * ms);
    out[8] = ((od[2] * inv) * s) - ((od[0] * inv) * c);
    out[9] = ((mod1 * inv) * s) - ((mod0 * inv) * c);
    out[13] = inv;
    out[14] = ((real_t)0.0 - (real_t)1.0) * inv;
}

/* ------------------------------------------------------------------------- */
/* 6x6 pivoted LU: CG/acado_integrator.c:127-250                               */
/* ------------------------------------------------------------------------- */

static void lu6_back_substitute(const real_t *A, real_t *b)
{
    int i, k;
    for (i = 5; i >= 0; --i) {
        for (k = 5; k > i; --k) b[i] -= A[i * 6 + k] * b[k];
        b[i] = b[i] / A[i * 6 + i];
    }
}

/* factor in place (multipliers stored negated below the diagonal), apply to b,
 * return |det| */
static real_t lu6_factor_solve(real_t *A, real_t *b, int *perm)
{
    real_t det = (real_t)1.0, tmp;
    int i, j, k;
    for (i = 0; i < 6; ++i) perm[i] = i;
    for (i = 0; i < 5; ++i) {
        int imax = i;
        real_t vmax = (real_t)fabs(A[i * 6 + i]);
        for (j = i + 1; j < 6; ++j) {
            tmp = (real_t)fabs(A[j * 6 + i]);
            if (tmp > vmax) {
                imax = j;
                vmax = tmp;
            }
        }
        if (imax > i) {
            int ip;
            for (k = 0; k < 6; ++k) {
                tmp = A[i * 6 + k];
                A[i * 6 + k] = A[imax * 6 + k];
                A[imax * 6 + k] = tmp;
            }
            tmp = b[i];
            b[i] = b[imax];
            b[imax] = tmp;
            ip = perm[i];
            perm[i] = perm[imax];
            perm[imax] = ip;
        }
        det *= A[i * 6 + i];
        for (j = i + 1; j < 6; ++j) {
            A[j * 6 + i] = -A[j * 6 + i] / A[i * 6 + i];
            for (k = i + 1; k < 6; ++k) A[j * 6 + k] += A[j * 6 + i] * A[i * 6 + k];
            b[j] += A[j * 6 + i] * b[i];
        }
    }
    det *= A[35];
    det = (real_t)fabs(det);
    lu6_back_substitute(A, b);
    return det;
}

static void lu6_resolve(const real_t *A, real_t *b, const int *perm)
{
    real_t bp[6];
    int i, k;
    for (i = 0; i < 6; ++i) bp[i] = b[perm[i]];
    for (i = 1; i < 6; ++i)
        for (k = 0; k < i; ++k) bp[i] += A[i * 6 + k] * bp[k];
    lu6_back_substitute(A, bp);
    for (i = 0; i < 6; ++i) b[i] = bp[i];
}

/* ------------------------------------------------------------------------- */
/* A2  IRK-GL2 step with forward sensitivities: CG/acado_integrator.c:261-449  */
/* ------------------------------------------------------------------------- */

static void irk_stage_state(const orc_nmpc *s, const real_t *eta, int stage, real_t *xxx)
{
    int j;
    for (j = 0; j < NX; ++j) {
        xxx[j] = eta[j];
        xxx[j] += s->Ah[stage * 2] * s->rk_kkk[j * 2];
        xxx[j] += s->Ah[stage * 2 + 1] * s->rk_kkk[j * 2 + 1];
    }
}

/* Newton matrix rows of one stage: Ah[stage][s'] * df/dx - I (:298-309) */
static void irk_fill_newton_rows(orc_nmpc *s, int stage, const real_t *dT)
{
    int j, c;
    for (j = 0; j < NX; ++j) {
        real_t *row = &s->rk_A[(stage * NX + j) * 6];
        for (c = 0; c < NX; ++c) row[c] = s->Ah[stage * 2] * dT[j * 5 + c];
        if (stage == 0) row[j] -= (real_t)1.0;
        for (c = 0; c < NX; ++c) row[NX + c] = s->Ah[stage * 2 + 1] * dT[j * 5 + c];
        if (stage == 1) row[j + NX] -= (real_t)1.0;
    }
}

static void irk_residual(const orc_nmpc *s, int stage, const real_t *rhs, real_t *b)
{
    b[stage * 3] = s->rk_kkk[stage] - rhs[0];
    b[stage * 3 + 1] = s->rk_kkk[stage + 2] - rhs[1];
    b[stage * 3 + 2] = s->rk_kkk[stage + 4] - rhs[2];
}

static void irk_apply_update(orc_nmpc *s, const real_t *b)
{
    int j;
    for (j = 0; j < 2; ++j) {
        s->rk_kkk[j] += b[j * 3];
        s->rk_kkk[j + 2] += b[j * 3 + 1];
        s->rk_kkk[j + 4] += b[j * 3 + 2];
    }
}

/* rk_eta[23]: [0..2] x | [3..11] Gx | [12..17] Gu | [18..19] u | [20..22] od
 * (CG/acado_solver.c:44-75) */
int orc_integrate(orc_nmpc *s, real_t *eta, int reset)
{
    real_t xxx[8], rhs[3], b[6], dT[2][15], diffK[6], dNew[15];
    real_t det = (real_t)0.0;
    int stage, i, j, dir;

    for (i = 0; i < 5; ++i) xxx[3 + i] = eta[18 + i];

    if (reset) { /* one full Newton iteration from the carried-over stage guess (:284-321) */
        for (stage = 0; stage < 2; ++stage) {
            irk_stage_state(s, eta, stage, xxx);
            orc_diffs(xxx, dT[stage]);
            irk_fill_newton_rows(s, stage, dT[stage]);
            orc_rhs(xxx, rhs);
            irk_residual(s, stage, rhs, b);
        }
        det = lu6_factor_solve(s->rk_A, b, s->rk_perm);
        irk_apply_update(s, b);
    }
    for (i = 0; i < 5; ++i) { /* simplified Newton re-using the LU (:323-346) */
        for (stage = 0; stage < 2; ++stage) {
            irk_stage_state(s, eta, stage, xxx);
            orc_rhs(xxx, rhs);
            irk_residual(s, stage, rhs, b);
        }
        lu6_resolve(s->rk_A, b, s->rk_perm);
        irk_apply_update(s, b);
    }
    /* Jacobian at the converged stages (:347-366) */
    for (stage = 0; stage < 2; ++stage) {
        irk_stage_state(s, eta, stage, xxx);
        orc_diffs(xxx, dT[stage]);
        irk_fill_newton_rows(s, stage, dT[stage]);
    }
    /* dK/dx: 3 right-hand sides, the first one re-factors (:367-391) */
    for (dir = 0; dir < NX; ++dir) {
        for (i = 0; i < 2; ++i)
            for (j = 0; j < NX; ++j) b[i * 3 + j] = -dT[i][dir + j * 5];
        if (dir == 0)
            det = lu6_factor_solve(s->rk_A, b, s->rk_perm);
        else
            lu6_resolve(s->rk_A, b, s->rk_perm);
        for (i = 0; i < 2; ++i) {
            diffK[i] = b[i * 3];
            diffK[i + 2] = b[i * 3 + 1];
            diffK[i + 4] = b[i * 3 + 2];
        }
        for (i = 0; i < NX; ++i) {
            dNew[i * 5 + dir] = (real_t)(i == dir);
            dNew[i * 5 + dir] += diffK[i * 2] * s->hb + diffK[i * 2 + 1] * s->hb;
        }
    }
    /* dK/du: 2 right-hand sides (:392-418) */
    for (dir = 0; dir < NU; ++dir) {
        for (i = 0; i < 2; ++i)
            for (j = 0; j < NX; ++j) b[i * 3 + j] = -dT[i][dir + j * 5 + 3];
        lu6_resolve(s->rk_A, b, s->rk_perm);
        for (i = 0; i < 2; ++i) {
            diffK[i] = b[i * 3];
            diffK[i + 2] = b[i * 3 + 1];
            diffK[i + 4] = b[i * 3 + 2];
        }
        for (i = 0; i < NX; ++i)
            dNew[i * 5 + dir + 3] = diffK[i * 2] * s->hb + diffK[i * 2 + 1] * s->hb;
    }
    /* x+ = x + dt/2 (k1 + k2) (:419-421) */
    for (j = 0; j < NX; ++j) eta[j] += s->rk_kkk[j * 2] * s->hb + s->rk_kkk[j * 2 + 1] * s->hb;
    for (i = 0; i < NX; ++i) {
        for (j = 0; j < NX; ++j) eta[3 + i * 3 + j] = dNew[i * 5 + j];
        for (j = 0; j < NU; ++j) eta[12 + i * 2 + j] = dNew[i * 5 + 3 + j];
    }
    /* status from |det| of the last factorisation (:441-447) */
    if (det < 1e-12) return 2;
    if (det < 1e-6) return 1;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A3  multiple-shooting simulation: CG/acado_solver.c:35-78                   */
/* ------------------------------------------------------------------------- */

static void load_node(const orc_nmpc *s, int k, real_t *state)
{
    int i;
    for (i = 0; i < NX; ++i) state[i] = s->x[k * NX + i];
    for (i = 0; i < NU; ++i) state[18 + i] = s->u[k * NU + i];
    for (i = 0; i < NOD; ++i) state[20 + i] = s->od[k * NOD + i];
}

int orc_model_simulation(orc_nmpc *s)
{
    real_t state[23];
    int k, i, ret = 0;
    memset(state, 0, sizeof state);
    for (k = 0; k < s->N; ++k) {
        load_node(s, k, state);
        ret = orc_integrate(s, state, 1); /* only the last interval's code survives (:54,77) */
        for (i = 0; i < NX; ++i) s->d[k * NX + i] = state[i] - s->x[(k + 1) * NX + i];
        for (i = 0; i < NX * NX; ++i) s->evGx[k * 9 + i] = state[3 + i];
        for (i = 0; i < NX * NU; ++i) s->evGu[k * 6 + i] = state[12 + i];
    }
    return ret;
}

/* ------------------------------------------------------------------------- */
/* A4  objective: CG/acado_solver.c:80-211                                     */
/* ------------------------------------------------------------------------- */

void orc_evaluate_objective(orc_nmpc *s)
{
    int k, r, c;
    for (k = 0; k < s->N; ++k) {
        const real_t *Wk = &s->W[k * 25];
        /* h(x,u) = (x, y, psi, vr, vl) (:80-91) */
        for (r = 0; r < NX; ++r) s->Dy[k * NY + r] = s->x[k * NX + r];
        for (r = 0; r < NU; ++r) s->Dy[k * NY + NX + r] = s->u[k * NU + r];
        /* Q2 = W[0:3,:], Q1 = W[0:3,0:3]; R2 = W[3:5,:], R1 = W[3:5,3:5] (:103-152) */
        for (r = 0; r < NX; ++r)
            for (c = 0; c < NY; ++c) s->Q2[k * 15 + r * NY + c] = Wk[r * NY + c];
        for (r = 0; r < NX; ++r)
            for (c = 0; c < NX; ++c) s->Q1[k * 9 + r * NX + c] = Wk[r * NY + c];
        for (r = 0; r < NU; ++r)
            for (c = 0; c < NY; ++c) s->R2[k * 10 + r * NY + c] = Wk[(NX + r) * NY + c];
        for (r = 0; r < NU; ++r)
            for (c = 0; c < NU; ++c) s->R1[k * 4 + r * NU + c] = Wk[(NX + r) * NY + NX + c];
    }
    for (r = 0; r < NYN; ++r) s->DyN[r] = s->x[s->N * NX + r]; /* hN(x) = x (:93-101) */
    for (r = 0; r < 9; ++r) {
        s->QN2[r] = s->WN[r];
        s->QN1[r] = s->WN[r];
    }
}

/* ------------------------------------------------------------------------- */
/* A5  O(N^2) condensing: CG/acado_solver.c:213-363                            */
/* ------------------------------------------------------------------------- */

/* C(3x2) = A(3x3) * B(3x2)  (multGxGu :213-221) */
static void mul_33_32(const real_t *A, const real_t *B, real_t *C)
{
    int r, c;
    for (r = 0; r < 3; ++r)
        for (c = 0; c < 2; ++c)
            C[r * 2 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[2 + c] + A[r * 3 + 2] * B[4 + c];
}

/* C(3x2) = A'(3x3) * B(3x2)  (multGxTGu :250-258) */
static void mul_33t_32(const real_t *A, const real_t *B, real_t *C)
{
    int r, c;
    for (r = 0; r < 3; ++r)
        for (c = 0; c < 2; ++c)
            C[r * 2 + c] = A[r] * B[c] + A[3 + r] * B[2 + c] + A[6 + r] * B[4 + c];
}

/* 2x2 block of H at block (iRow, iCol) = Gu' * W1 (+ R)  (multBTW1 / multBTW1_R1 :234-248) */
static void put_h_block(orc_nmpc *s, const real_t *Gu, const real_t *W1, const real_t *R, int iRow,
                        int iCol)
{
    const int n = 2 * s->N;
    int r, c;
    for (r = 0; r < 2; ++r)
        for (c = 0; c < 2; ++c) {
            real_t v = Gu[r] * W1[c] + Gu[2 + r] * W1[2 + c] + Gu[4 + r] * W1[4 + c];
            if (R) v = v + R[r * 2 + c];
            s->H[(iRow * 2 + r) * n + iCol * 2 + c] = v;
        }
}

void orc_condense_prep(orc_nmpc *s)
{
    const int N = s->N, n = 2 * N;
    real_t W1[6], W2[6];
    int j, i, r, c;
    for (j = 0; j < N; ++j) {
        const int col0 = (j * (2 * N + 1 - j)) / 2; /* first block of column j of E */
        /* E_{j,j} = Gu_j ; E_{i,j} = Gx_i E_{i-1,j} */
        memcpy(&s->E[col0 * 6], &s->evGu[j * 6], 6 * sizeof(real_t));
        for (i = 1; i < N - j; ++i)
            mul_33_32(&s->evGx[(j + i) * 9], &s->E[(col0 + i - 1) * 6], &s->E[(col0 + i) * 6]);
        /* backward sweep over rows of H column block j */
        mul_33_32(s->QN1, &s->E[(col0 + N - 1 - j) * 6], W1);
        for (i = N - 1; i > j; --i) {
            put_h_block(s, &s->evGu[i * 6], W1, NULL, i, j);
            mul_33t_32(&s->evGx[i * 9], W1, W2);
            /* W1 = Q1_i * E_{i-1,j} + W2  (multQEW2 :260-268) */
            {
                const real_t *Q = &s->Q1[i * 9], *Ep = &s->E[(col0 + i - j - 1) * 6];
                for (r = 0; r < 3; ++r)
                    for (c = 0; c < 2; ++c)
                        W1[r * 2 + c] = Q[r * 3] * Ep[c] + Q[r * 3 + 1] * Ep[2 + c] +
                                        Q[r * 3 + 2] * Ep[4 + c] + W2[r * 2 + c];
            }
        }
        put_h_block(s, &s->evGu[j * 6], W1, &s->R1[j * 4], j, j);
    }
    /* mirror the strict lower block triangle (copyHTH :311-317, :351-357) */
    for (i = 0; i < N; ++i)
        for (j = 0; j < i; ++j)
            for (r = 0; r < 2; ++r)
                for (c = 0; c < 2; ++c)
                    s->H[(j * 2 + r) * n + i * 2 + c] = s->H[(i * 2 + c) * n + j * 2 + r];
    for (i = 0; i < NX * N; ++i) s->sbar[i + NX] = s->d[i]; /* :359-360 */
}

/* ------------------------------------------------------------------------- */
/* A6  feedback-side condensing: CG/acado_solver.c:365-891                     */
/* ------------------------------------------------------------------------- */

void orc_condense_fdb(orc_nmpc *s)
{
    const int N = s->N;
    real_t w1[3], w2[3];
    int k, r;
    for (r = 0; r < NX; ++r) s->Dx0[r] = s->x0[r] - s->x[r];
    for (k = 0; k < NY * N; ++k) s->Dy[k] -= s->y[k];
    for (r = 0; r < NYN; ++r) s->DyN[r] -= s->yN[r];
    /* g_k = R2_k Dy_k (multRDy :319-323) ; QDy_k = Q2_k Dy_k (multQDy :325-330) */
    for (k = 0; k < N; ++k) {
        const real_t *Dy = &s->Dy[k * NY];
        for (r = 0; r < NU; ++r) {
            const real_t *R2 = &s->R2[k * 10 + r * NY];
            s->g[k * NU + r] =
                R2[0] * Dy[0] + R2[1] * Dy[1] + R2[2] * Dy[2] + R2[3] * Dy[3] + R2[4] * Dy[4];
        }
        for (r = 0; r < NX; ++r) {
            const real_t *Q2 = &s->Q2[k * 15 + r * NY];
            s->QDy[k * NX + r] =
                Q2[0] * Dy[0] + Q2[1] * Dy[1] + Q2[2] * Dy[2] + Q2[3] * Dy[3] + Q2[4] * Dy[4];
        }
    }
    for (r = 0; r < NX; ++r)
        s->QDy[N * NX + r] = s->QN2[r * 3] * s->DyN[0] + s->QN2[r * 3 + 1] * s->DyN[1] +
                             s->QN2[r * 3 + 2] * s->DyN[2];
    /* sbar_0 = Dx0 ; sbar_{k+1} (= d_k from condensePrep) += Gx_k sbar_k (macASbar :297-302) */
    for (r = 0; r < NX; ++r) s->sbar[r] = s->Dx0[r];
    for (k = 0; k < N; ++k) {
        const real_t *Gx = &s->evGx[k * 9], *a = &s->sbar[k * NX];
        real_t *o = &s->sbar[(k + 1) * NX];
        for (r = 0; r < NX; ++r) o[r] += Gx[r * 3] * a[0] + Gx[r * 3 + 1] * a[1] + Gx[r * 3 + 2] * a[2];
    }
    /* backward adjoint sweep */
    {
        const real_t *sN = &s->sbar[N * NX];
        for (r = 0; r < NX; ++r)
            w1[r] = s->QN1[r * 3] * sN[0] + s->QN1[r * 3 + 1] * sN[1] + s->QN1[r * 3 + 2] * sN[2] +
                    s->QDy[N * NX + r];
    }
    for (k = N - 1; k >= 0; --k) {
        const real_t *Gu = &s->evGu[k * 6], *Gx = &s->evGx[k * 9];
        /* g_k += Gu_k' w1 (macBTw1 :284-288) */
        for (r = 0; r < NU; ++r) s->g[k * NU + r] += Gu[r] * w1[0] + Gu[2 + r] * w1[1] + Gu[4 + r] * w1[2];
        if (k == 0) break;
        /* w2 = Gx_k' w1 + QDy_k (macATw1QDy :277-282) */
        for (r = 0; r < NX; ++r)
            w2[r] = Gx[r] * w1[0] + Gx[3 + r] * w1[1] + Gx[6 + r] * w1[2] + s->QDy[k * NX + r];
        /* w1 = Q1_k sbar_k + w2 (macQSbarW2 :290-295) */
        {
            const real_t *Q = &s->Q1[k * 9], *sb = &s->sbar[k * NX];
            for (r = 0; r < NX; ++r)
                w1[r] = Q[r * 3] * sb[0] + Q[r * 3 + 1] * sb[1] + Q[r * 3 + 2] * sb[2] + w2[r];
        }
    }
    for (k = 0; k < NU * N; ++k) {
        s->lb[k] = s->lbValues[k] - s->u[k];
        s->ub[k] = s->ubValues[k] - s->u[k];
    }
}

/* ------------------------------------------------------------------------- */
/* A7  dense box-QP, qpOASES QProblemB online active-set strategy             */
/* ------------------------------------------------------------------------- */

enum { ST_INACTIVE = 0, ST_LOWER = 1, ST_UPPER = 2, ST_UNDEFINED = 3 };
enum { TY_BOUNDED = 0, TY_EQUALITY = 1, TY_UNBOUNDED = 2 };

typedef struct {
    int n;
    real_t *H, *R;                /* n x n row-major */
    real_t *g, *lb, *ub, *x, *y;  /* current (homotopy) QP data + primal/dual */
    int *status, *type;
    int *FR, *FX;                 /* ordered index lists (QPO/SRC/Indexlist.cpp:144-262) */
    int nFR, nFX;
    int noLower, noUpper, identityHessian, infeasible;
    real_t tau;
} qpb;

static real_t absr(real_t v) { return v >= (real_t)0.0 ? v : -v; }

static int list_index(const int *L, int len, int number)
{
    int i;
    for (i = 0; i < len; ++i)
        if (L[i] == number) return i;
    return -1;
}

static void list_remove(int *L, int *len, int number)
{
    int i = list_index(L, *len, number);
    if (i < 0) return;
    for (; i + 1 < *len; ++i) L[i] = L[i + 1];
    --*len;
}

/* R a = b or R' a = b with the leading nR x nR block (QProblemB.cpp:1411-1464) */
static int qpb_backsolve(const qpb *q, const real_t *b, int transposed, int nR, real_t *a)
{
    const int n = q->n;
    int i, j;
    real_t sum;
    if (nR <= 0) return RET_OK;
    if (!transposed) {
        for (i = nR - 1; i >= 0; --i) {
            sum = b[i];
            for (j = i + 1; j < nR; ++j) sum -= q->R[i * n + j] * a[j];
            if (absr(q->R[i * n + i]) > QP_ZERO)
                a[i] = sum / q->R[i * n + i];
            else
                return RET_DIV_BY_ZERO;
        }
    } else {
        for (i = 0; i < nR; ++i) {
            sum = b[i];
            for (j = 0; j < i; ++j) sum -= q->R[j * n + i] * a[j];
            if (absr(q->R[i * n + i]) > QP_ZERO)
                a[i] = sum / q->R[i * n + i];
            else
                return RET_DIV_BY_ZERO;
        }
    }
    return RET_OK;
}

/* Cholesky R'R = H projected onto the free variables (QProblemB.cpp:753-824) */
static int qpb_cholesky(qpb *q)
{
    const int n = q->n, nFR = q->nFR;
    int i, j, k;
    real_t sum, inv;
    for (i = 0; i < n * n; ++i) q->R[i] = (real_t)0.0;
    if (q->identityHessian) {
        for (i = 0; i < nFR; ++i) q->R[i * n + i] = (real_t)1.0;
        return RET_OK;
    }
    for (i = 0; i < nFR; ++i) {
        const int ii = q->FR[i];
        sum = q->H[ii * n + ii];
        for (k = i - 1; k >= 0; --k) sum -= q->R[k * n + i] * q->R[k * n + i];
        if (sum > (real_t)0.0) {
            q->R[i * n + i] = (real_t)sqrt(sum);
            inv = (real_t)1.0 / q->R[i * n + i];
        } else {
            return RET_HESSIAN_NOT_SPD;
        }
        for (j = i + 1; j < nFR; ++j) {
            const int jj = q->FR[j];
            sum = q->H[jj * n + ii];
            for (k = i - 1; k >= 0; --k) sum -= q->R[k * n + i] * q->R[k * n + j];
            q->R[i * n + j] = sum * inv;
        }
    }
    return RET_OK;
}

/* Givens rotation, QPO/SRC/QProblemB.ipp:369-414 */
static void givens(real_t xold, real_t yold, real_t *xnew, real_t *ynew, real_t *c, real_t *s)
{
    if (absr(yold) <= QP_ZERO) {
        *c = (real_t)1.0;
        *s = (real_t)0.0;
        *xnew = xold;
        *ynew = yold;
    } else {
        real_t t, mu = absr(xold);
        if (absr(yold) > mu) mu = absr(yold);
        t = mu * (real_t)sqrt((xold / mu) * (xold / mu) + (yold / mu) * (yold / mu));
        if (xold < (real_t)0.0) t = -t;
        *c = xold / t;
        *s = yold / t;
        *xnew = t;
        *ynew = (real_t)0.0;
    }
}

/* fix variable `number` at a bound; optionally down-date R (QProblemB.cpp:1269-1326) */
static int qpb_add_bound(qpb *q, int number, int st, int updateCholesky)
{
    const int n = q->n, nFR = q->nFR;
    if (updateCholesky) {
        const int idx = list_index(q->FR, nFR, number);
        real_t c, s;
        int i, j;
        for (i = idx + 1; i < nFR; ++i) {
            givens(q->R[(i - 1) * n + i], q->R[i * n + i], &q->R[(i - 1) * n + i], &q->R[i * n + i], &c, &s);
            for (j = i + 1; j < nFR; ++j) {
                const real_t xo = q->R[(i - 1) * n + j], yo = q->R[i * n + j];
                q->R[(i - 1) * n + j] = c * xo + s * yo;
                q->R[i * n + j] = -s * xo + c * yo;
            }
        }
        for (i = 0; i < nFR - 1; ++i)
            for (j = idx + 1; j < nFR; ++j) q->R[i * n + j - 1] = q->R[i * n + j];
        for (i = 0; i < nFR; ++i) q->R[i * n + nFR - 1] = (real_t)0.0;
    }
    list_remove(q->FR, &q->nFR, number);
    q->FX[q->nFX++] = number;
    q->status[number] = st;
    return RET_OK;
}

/* free variable `number`; optionally append its column to R (QProblemB.cpp:1329-1393) */
static int qpb_remove_bound(qpb *q, int number, int updateCholesky, real_t *wrk)
{
    const int n = q->n, nFR = q->nFR; /* count BEFORE the move, as in the reference */
    real_t *rhs = wrk, *r = wrk + n;
    real_t r0;
    int i;
    list_remove(q->FX, &q->nFX, number);
    q->FR[q->nFR++] = number;
    q->status[number] = ST_INACTIVE;
    if (!updateCholesky) return RET_OK;
    r0 = q->H[number * n + number];
    for (i = 0; i < nFR; ++i) rhs[i] = q->H[number * n + q->FR[i]];
    if (qpb_backsolve(q, rhs, 1, nFR, r) != RET_OK) return 75; /* RET_REMOVEBOUND_FAILED */
    for (i = 0; i < nFR; ++i) r0 -= r[i] * r[i];
    for (i = 0; i < nFR; ++i) q->R[i * n + nFR] = r[i];
    if (r0 > (real_t)0.0)
        q->R[nFR * n + nFR] = (real_t)sqrt(r0);
    else
        return RET_HESSIAN_NOT_SPD;
    return RET_OK;
}

/* homotopy from the current (auxiliary) QP to (g_new, lb_new, ub_new)
 * (QProblemB::hotstart, QProblemB.cpp:315-506 with its helpers :1470-1524,
 * :1637-2015) */
static int qpb_hotstart(qpb *q, const real_t *g_new, const real_t *lb_new, const real_t *ub_new,
                        int *nWSR, real_t *wrk)
{
    const int n = q->n;
    real_t *delta_g = wrk, *delta_lb = wrk + n, *delta_ub = wrk + 2 * n, *delta_xFR = wrk + 3 * n,
           *delta_xFX = wrk + 4 * n, *delta_yFX = wrk + 5 * n, *HMX = wrk + 6 * n, *tmp = wrk + 7 * n,
           *rhsz = wrk + 8 * n, *upd = wrk + 9 * n /* 2n */;
    int l, i, j;
    q->infeasible = 0;
    for (l = 0; l < *nWSR; ++l) {
        const int nFR = q->nFR, nFX = q->nFX;
        int isZero = 1, BC_idx = 0, BC_status = ST_UNDEFINED, rv;
        real_t tau_new = (real_t)1.0, tau_tmp;

        /* data shift (:1470-1524) */
        for (i = 0; i < n; ++i) {
            delta_g[i] = g_new[i] - q->g[i];
            delta_lb[i] = lb_new[i] - q->lb[i];
            delta_ub[i] = ub_new[i] - q->ub[i];
        }
        for (i = 0; i < nFX; ++i) {
            const int ii = q->FX[i];
            if (absr(delta_lb[ii]) > QP_EPS || absr(delta_ub[ii]) > QP_EPS) {
                isZero = 0;
                break;
            }
        }

        /* step direction (:1637-1747) */
        for (i = 0; i < nFR; ++i) HMX[i] = (real_t)0.0;
        for (i = 0; i < nFX; ++i) {
            const int ii = q->FX[i];
            delta_xFX[i] = (q->status[ii] == ST_LOWER) ? delta_lb[ii] : delta_ub[ii];
        }
        if (nFR > 0) {
            if (!isZero)
                for (i = 0; i < nFR; ++i)
                    for (j = 0; j < nFX; ++j) HMX[i] += q->H[q->FR[i] * n + q->FX[j]] * delta_xFX[j];
            for (j = 0; j < nFR; ++j)
                rhsz[j] = isZero ? delta_g[q->FR[j]] : (delta_g[q->FR[j]] + HMX[j]);
            for (i = 0; i < nFR; ++i) delta_xFR[i] = -rhsz[i];
            if (qpb_backsolve(q, delta_xFR, 1, nFR, tmp) != RET_OK ||
                qpb_backsolve(q, tmp, 0, nFR, delta_xFR) != RET_OK) {
                *nWSR = l;
                return RET_STEPDIRECTION_FAILED_CHOLESKY;
            }
        }
        for (i = 0; i < nFX; ++i) {
            const int ii = q->FX[i];
            delta_yFX[i] = (real_t)0.0;
            for (j = 0; j < nFR; ++j) delta_yFX[i] += q->H[ii * n + q->FR[j]] * delta_xFR[j];
            if (!isZero)
                for (j = 0; j < nFX; ++j) delta_yFX[i] += q->H[ii * n + q->FX[j]] * delta_xFX[j];
            delta_yFX[i] += delta_g[ii];
        }

        /* step length (:1753-1898): dual ratio test, then primal ratio tests */
        for (i = 0; i < nFX; ++i) {
            const int ii = q->FX[i];
            if (q->type[ii] == TY_EQUALITY) continue;
            if (q->status[ii] == ST_LOWER) {
                if (delta_yFX[i] < -QP_ZERO && q->y[ii] >= (real_t)0.0) {
                    tau_tmp = q->y[ii] / (-delta_yFX[i]);
                    if (tau_tmp < tau_new && tau_tmp >= (real_t)0.0) {
                        tau_new = tau_tmp;
                        BC_idx = ii;
                        BC_status = ST_INACTIVE;
                    }
                }
            } else {
                if (delta_yFX[i] > QP_ZERO && q->y[ii] <= (real_t)0.0) {
                    tau_tmp = q->y[ii] / (-delta_yFX[i]);
                    if (tau_tmp < tau_new && tau_tmp >= (real_t)0.0) {
                        tau_new = tau_tmp;
                        BC_idx = ii;
                        BC_status = ST_INACTIVE;
                    }
                }
            }
        }
        if (!q->noLower)
            for (i = 0; i < nFR; ++i) {
                const int ii = q->FR[i];
                if (q->type[ii] == TY_UNBOUNDED) continue;
                if (delta_lb[ii] > delta_xFR[i]) {
                    if (q->x[ii] > q->lb[ii])
                        tau_tmp = (q->x[ii] - q->lb[ii]) / (delta_lb[ii] - delta_xFR[i]);
                    else
                        tau_tmp = (real_t)0.0;
                    if (tau_tmp < tau_new && tau_tmp >= (real_t)0.0) {
                        tau_new = tau_tmp;
                        BC_idx = ii;
                        BC_status = ST_LOWER;
                    }
                }
            }
        if (!q->noUpper)
            for (i = 0; i < nFR; ++i) {
                const int ii = q->FR[i];
                if (q->type[ii] == TY_UNBOUNDED) continue;
                if (delta_ub[ii] < delta_xFR[i]) {
                    if (q->x[ii] < q->ub[ii])
                        tau_tmp = (q->x[ii] - q->ub[ii]) / (delta_ub[ii] - delta_xFR[i]);
                    else
                        tau_tmp = (real_t)0.0;
                    if (tau_tmp < tau_new && tau_tmp >= (real_t)0.0) {
                        tau_new = tau_tmp;
                        BC_idx = ii;
                        BC_status = ST_UPPER;
                    }
                }
            }
        q->tau = tau_new;

        /* perform step (:1904-2015) */
        for (i = 0; i < n; ++i) /* areBoundsConsistent (:1530-1542) */
            if (q->lb[i] > q->ub[i] - QP_BOUNDTOL && delta_lb[i] > delta_ub[i] + QP_EPS) {
                q->infeasible = 1;
                q->tau = (real_t)0.0;
                *nWSR = l;
                return RET_HOTSTART_STOPPED_INFEASIBILITY;
            }
        if (q->tau > QP_ZERO) {
            for (i = 0; i < nFR; ++i) q->x[q->FR[i]] += q->tau * delta_xFR[i];
            for (i = 0; i < nFX; ++i) {
                const int ii = q->FX[i];
                q->x[ii] += q->tau * delta_xFX[i];
                q->y[ii] += q->tau * delta_yFX[i];
            }
            for (i = 0; i < n; ++i) {
                q->g[i] += q->tau * delta_g[i];
                q->lb[i] += q->tau * delta_lb[i];
                q->ub[i] += q->tau * delta_ub[i];
            }
        }
        if (BC_status == ST_UNDEFINED) { /* no blocking bound: optimal (KKT re-check is
                                            compiled out in the reference, :2077) */
            *nWSR = l;
            return RET_OK;
        }
        if (BC_status == ST_INACTIVE) {
            rv = qpb_remove_bound(q, BC_idx, 1, upd);
            if (rv != RET_OK) {
                *nWSR = l;
                return RET_REMOVE_FROM_ACTIVESET_FAILED;
            }
            q->y[BC_idx] = (real_t)0.0;
        } else {
            rv = qpb_add_bound(q, BC_idx, BC_status, 1);
            if (rv != RET_OK) {
                *nWSR = l;
                return RET_ADD_TO_ACTIVESET_FAILED;
            }
        }
    }
    return RET_MAX_NWSR_REACHED;
}

/* QProblemB::init -> solveInitialQP (QProblemB.cpp:285-296, 830-937) */
int orc_qpb_solve(int n, const real_t *H, const real_t *g, const real_t *lb, const real_t *ub,
                  const real_t *yOpt, int *nWSR, real_t *x_out, real_t *y_out)
{
    qpb q;
    real_t *fbuf, *g0, *lb0, *ub0, *wrk;
    int *ibuf;
    int i, j, rv, nFV = 0;
    if (n <= 0 || !H || !g) return RET_INVALID_ARGUMENTS;
    fbuf = (real_t *)calloc((size_t)(2 * n * n + 20 * n), sizeof(real_t));
    ibuf = (int *)calloc((size_t)(4 * n), sizeof(int));
    if (!fbuf || !ibuf) {
        free(fbuf);
        free(ibuf);
        return RET_INIT_FAILED;
    }
    q.n = n;
    q.H = fbuf;
    q.R = fbuf + n * n;
    q.g = fbuf + 2 * n * n;
    q.lb = q.g + n;
    q.ub = q.lb + n;
    q.x = q.ub + n;
    q.y = q.x + n;
    g0 = q.y + n;
    lb0 = g0 + n;
    ub0 = lb0 + n;
    wrk = ub0 + n; /* 11 n */
    q.status = ibuf;
    q.type = ibuf + n;
    q.FR = ibuf + 2 * n;
    q.FX = ibuf + 3 * n;
    q.infeasible = 0;
    q.tau = (real_t)0.0;

    /* setupQPdata (:1548-1631): missing bounds mean -/+ INFTY */
    memcpy(q.H, H, (size_t)(n * n) * sizeof(real_t));
    for (i = 0; i < n; ++i) {
        q.g[i] = g[i];
        q.lb[i] = lb ? lb[i] : -QP_INFTY;
        q.ub[i] = ub ? ub[i] : QP_INFTY;
    }
    /* checkForIdentityHessian (:659-688) */
    q.identityHessian = 1;
    for (i = 0; i < n && q.identityHessian; ++i)
        if (absr(q.H[i * n + i] - (real_t)1.0) > QP_EPS) q.identityHessian = 0;
    for (i = 0; i < n && q.identityHessian; ++i)
        for (j = 0; j < i; ++j)
            if (absr(q.H[i * n + j]) > QP_EPS || absr(q.H[j * n + i]) > QP_EPS) {
                q.identityHessian = 0;
                break;
            }
    /* setupSubjectToType (:694-745) */
    q.noLower = 1;
    q.noUpper = 1;
    for (i = 0; i < n; ++i) {
        if (q.lb[i] > -QP_INFTY) q.noLower = 0;
        if (q.ub[i] < QP_INFTY) q.noUpper = 0;
    }
    for (i = 0; i < n; ++i) {
        if (q.lb[i] < -QP_INFTY + QP_BOUNDTOL && q.ub[i] > QP_INFTY - QP_BOUNDTOL)
            q.type[i] = TY_UNBOUNDED;
        else if (q.lb[i] > q.ub[i] - QP_BOUNDTOL) {
            q.type[i] = TY_EQUALITY;
            ++nFV;
        } else
            q.type[i] = TY_BOUNDED;
    }
    (void)nFV;
    /* all free, in index order (Bounds::setupAllFree) */
    q.nFR = n;
    q.nFX = 0;
    for (i = 0; i < n; ++i) {
        q.FR[i] = i;
        q.status[i] = ST_INACTIVE;
    }
    /* auxiliary QP solution x = 0, y = yOpt (:1127-1165) */
    for (i = 0; i < n; ++i) {
        q.x[i] = (real_t)0.0;
        q.y[i] = yOpt ? yOpt[i] : (real_t)0.0;
    }
    /* working set guessed from the sign of yOpt (:1010-1036, 1038-1060), fixed
     * in index order without touching R (:1064-1122 with setupAfresh) */
    for (i = 0; i < n; ++i) {
        int want = ST_INACTIVE;
        if (yOpt && yOpt[i] > QP_ZERO)
            want = ST_LOWER;
        else if (yOpt && yOpt[i] < -QP_ZERO)
            want = ST_UPPER;
        else if (q.type[i] == TY_EQUALITY)
            want = ST_LOWER;
        if (want != ST_INACTIVE) qpb_add_bound(&q, i, want, 0);
    }
    rv = qpb_cholesky(&q);
    if (rv != RET_OK) {
        rv = RET_INIT_FAILED_CHOLESKY;
        goto done;
    }
    for (i = 0; i < n; ++i) {
        g0[i] = q.g[i];
        lb0[i] = q.lb[i];
        ub0[i] = q.ub[i];
    }
    /* auxiliary gradient g = y - H x (:1170-1189) */
    for (i = 0; i < n; ++i) {
        q.g[i] = q.y[i];
        for (j = 0; j < n; ++j) q.g[i] -= q.H[i * n + j] * q.x[j];
    }
    /* auxiliary bounds with relaxation (:1195-1257) */
    for (i = 0; i < n; ++i) {
        const int eq = (q.type[i] == TY_EQUALITY);
        switch (q.status[i]) {
        case ST_INACTIVE:
            q.lb[i] = eq ? q.x[i] : q.x[i] - QP_BOUNDRELAXATION;
            q.ub[i] = eq ? q.x[i] : q.x[i] + QP_BOUNDRELAXATION;
            break;
        case ST_LOWER:
            q.lb[i] = q.x[i];
            q.ub[i] = eq ? q.x[i] : q.x[i] + QP_BOUNDRELAXATION;
            break;
        default:
            q.ub[i] = q.x[i];
            q.lb[i] = eq ? q.x[i] : q.x[i] - QP_BOUNDRELAXATION;
            break;
        }
    }
    rv = qpb_hotstart(&q, g0, lb0, ub0, nWSR, wrk);
    if (q.infeasible)
        rv = RET_INIT_FAILED_INFEASIBILITY;
    else if (rv != RET_OK && rv != RET_MAX_NWSR_REACHED)
        rv = RET_INIT_FAILED_HOTSTART;
done:
    /* getPrimalSolution / getDualSolution copy whatever the state is */
    for (i = 0; i < n; ++i) {
        if (x_out) x_out[i] = q.x[i];
        if (y_out) y_out[i] = q.y[i];
    }
    free(fbuf);
    free(ibuf);
    return rv;
}

/* CG/acado_qpoases_interface.cpp:39-60 */
int orc_solve_qp(orc_nmpc *s)
{
    const int n = NU * s->N;
    int rv;
    s->nWSR = QP_NWSRMAX;
    rv = orc_qpb_solve(n, s->H, s->g, s->lb, s->ub, s->dual, &s->nWSR, s->dx, s->dual);
    return rv;
}

/* ------------------------------------------------------------------------- */
/* A8  expansion: CG/acado_solver.c:893-1055                                   */
/* ------------------------------------------------------------------------- */

void orc_expand(orc_nmpc *s)
{
    const int N = s->N;
    int k, r;
    for (k = 0; k < NU * N; ++k) s->u[k] += s->dx[k];
    for (r = 0; r < NX; ++r) s->sbar[r] = s->Dx0[r];
    for (k = 0; k < NX * N; ++k) s->sbar[k + NX] = s->d[k];
    for (k = 0; k < N; ++k) { /* expansionStep :304-313 */
        const real_t *Gx = &s->evGx[k * 9], *Gu = &s->evGu[k * 6], *U = &s->dx[k * NU],
                     *a = &s->sbar[k * NX];
        real_t *o = &s->sbar[(k + 1) * NX];
        for (r = 0; r < NX; ++r) o[r] += Gx[r * 3] * a[0] + Gx[r * 3 + 1] * a[1] + Gx[r * 3 + 2] * a[2];
        for (r = 0; r < NX; ++r) o[r] += Gu[r * 2] * U[0] + Gu[r * 2 + 1] * U[1];
    }
    for (k = 0; k < NX * (N + 1); ++k) s->x[k] += s->sbar[k];
}

/* ------------------------------------------------------------------------- */
/* A9  RTI API: CG/acado_solver.c:1057-1450                                    */
/* ------------------------------------------------------------------------- */

int orc_preparation_step(orc_nmpc *s)
{
    int ret = orc_model_simulation(s);
    orc_evaluate_objective(s);
    orc_condense_prep(s);
    return ret;
}

int orc_feedback_step(orc_nmpc *s)
{
    int tmp;
    orc_condense_fdb(s);
    tmp = orc_solve_qp(s);
    orc_expand(s);
    return tmp;
}

static size_t ws_floats(int N)
{
    const size_t n = 2 * (size_t)N;
    return 3 * N + 5 * N + 3 + 9 * N + 6 * N + 9 * N + 15 * N + 4 * N + 10 * N + 9 + 9 + 3 * (N + 1) + 3 +
           (size_t)N * (N + 1) / 2 * 6 + 3 * (N + 1) + n * n + 5 * n;
}

/* zero the workspace, bake the +-3 wheel-speed bounds (:1079-1290; the bounds
 * come from UAV_CAR_model.cpp:97-101).  acadoVariables is NOT touched, like
 * the reference. */
int orc_initialize_solver(orc_nmpc *s)
{
    int k;
    memset(s->d, 0, ws_floats(s->N) * sizeof(real_t));
    for (k = 0; k < NU * s->N; ++k) {
        s->lbValues[k] = (real_t)-3.0;
        s->ubValues[k] = (real_t)3.0;
    }
    return 0;
}

/* :1292-1312 -- note the integrator is reset only on the first interval */
void orc_initialize_nodes_by_forward_simulation(orc_nmpc *s)
{
    real_t state[23];
    int k, i;
    memset(state, 0, sizeof state);
    for (k = 0; k < s->N; ++k) {
        load_node(s, k, state);
        orc_integrate(s, state, k == 0);
        for (i = 0; i < NX; ++i) s->x[(k + 1) * NX + i] = state[i];
    }
}

/* :1314-1355 */
void orc_shift_states(orc_nmpc *s, int strategy, const real_t *xEnd, const real_t *uEnd)
{
    const int N = s->N;
    int k;
    for (k = 0; k < NX * N; ++k) s->x[k] = s->x[k + NX];
    if (strategy == 1 && xEnd) {
        for (k = 0; k < NX; ++k) s->x[N * NX + k] = xEnd[k];
    } else if (strategy == 2) {
        real_t state[23];
        memset(state, 0, sizeof state);
        for (k = 0; k < NX; ++k) state[k] = s->x[N * NX + k];
        for (k = 0; k < NU; ++k) state[18 + k] = uEnd ? uEnd[k] : s->u[(N - 1) * NU + k];
        for (k = 0; k < NOD; ++k) state[20 + k] = s->od[N * NOD + k];
        orc_integrate(s, state, 1);
        for (k = 0; k < NX; ++k) s->x[N * NX + k] = state[k];
    }
}

/* :1357-1371 */
void orc_shift_controls(orc_nmpc *s, const real_t *uEnd)
{
    const int N = s->N;
    int k;
    for (k = 0; k < NU * (N - 1); ++k) s->u[k] = s->u[k + NU];
    if (uEnd)
        for (k = 0; k < NU; ++k) s->u[(N - 1) * NU + k] = uEnd[k];
}

/* :1373-1391 */
real_t orc_get_kkt(orc_nmpc *s)
{
    const int n = NU * s->N;
    real_t kkt = (real_t)0.0, prd;
    int i;
    for (i = 0; i < n; ++i) kkt = (i == 0) ? s->g[0] * s->dx[0] : kkt + s->g[i] * s->dx[i];
    kkt = (real_t)fabs(kkt);
    for (i = 0; i < n; ++i) {
        prd = s->dual[i];
        if (prd > 1e-12)
            kkt += (real_t)fabs(s->lb[i] * prd);
        else if (prd < -1e-12)
            kkt += (real_t)fabs(s->ub[i] * prd);
    }
    return kkt;
}

/* :1393-1450 -- overwrites Dy / DyN like the reference, and uses only the
 * diagonal of WN in the terminal term (:1442-1444) */
real_t orc_get_objective(orc_nmpc *s)
{
    const int N = s->N;
    real_t obj = (real_t)0.0, t[5];
    int k, r, c;
    for (k = 0; k < N; ++k) {
        for (r = 0; r < NX; ++r) s->Dy[k * NY + r] = s->x[k * NX + r] - s->y[k * NY + r];
        for (r = 0; r < NU; ++r) s->Dy[k * NY + NX + r] = s->u[k * NU + r] - s->y[k * NY + NX + r];
    }
    for (r = 0; r < NYN; ++r) s->DyN[r] = s->x[N * NX + r] - s->yN[r];
    for (k = 0; k < N; ++k) {
        const real_t *Dy = &s->Dy[k * NY], *Wk = &s->W[k * 25];
        for (c = 0; c < NY; ++c)
            t[c] = Dy[0] * Wk[c] + Dy[1] * Wk[5 + c] + Dy[2] * Wk[10 + c] + Dy[3] * Wk[15 + c] +
                   Dy[4] * Wk[20 + c];
        obj += Dy[0] * t[0] + Dy[1] * t[1] + Dy[2] * t[2] + Dy[3] * t[3] + Dy[4] * t[4];
    }
    t[0] = s->DyN[0] * s->WN[0];
    t[1] = s->DyN[1] * s->WN[4];
    t[2] = s->DyN[2] * s->WN[8];
    obj += s->DyN[0] * t[0] + s->DyN[1] * t[1] + s->DyN[2] * t[2];
    obj *= (real_t)0.5;
    return obj;
}

int orc_get_nwsr(const orc_nmpc *s) { return s->nWSR; }
int orc_N(const orc_nmpc *s) { return s->N; }

/* ------------------------------------------------------------------------- */
/* handle management                                                          */
/* ------------------------------------------------------------------------- */

orc_nmpc *orc_create(int N, double dt)
{
    orc_nmpc *s;
    real_t *p;
    size_t nvar, nws;
    const size_t n = 2 * (size_t)N;
    if (N < 1 || !(dt > 0.0)) return NULL;
    s = (orc_nmpc *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->N = N;
    /* 2-stage Gauss-Legendre, stages in the order the generated code uses
     * (CG/acado_integrator.c:255-257: Ah = dt * [[1/4, 1/4+sqrt3/6],[1/4-sqrt3/6, 1/4]]) */
    s->Ah[0] = (real_t)(dt * 0.25);
    s->Ah[1] = (real_t)(dt * (0.25 + 0.28867513459481290));
    s->Ah[2] = (real_t)(dt * (0.25 - 0.28867513459481287));
    s->Ah[3] = (real_t)(dt * 0.25);
    s->hb = (real_t)(dt * 0.5);
    nvar = 3 * (N + 1) + 2 * N + 3 * (N + 1) + 5 * N + 3 + 25 * N + 9 + 3 + 2 * n;
    nws = ws_floats(N);
    s->pool = p = (real_t *)calloc(nvar + nws, sizeof(real_t));
    if (!p) {
        free(s);
        return NULL;
    }
#define TAKE(field, count) \
    s->field = p;          \
    p += (count)
    TAKE(x, 3 * (N + 1));
    TAKE(u, 2 * N);
    TAKE(od, 3 * (N + 1));
    TAKE(y, 5 * N);
    TAKE(yN, 3);
    TAKE(W, 25 * N);
    TAKE(WN, 9);
    TAKE(x0, 3);
    TAKE(lbValues, n);
    TAKE(ubValues, n);
    /* workspace: contiguous from d (orc_initialize_solver zeroes ws_floats from d) */
    TAKE(d, 3 * N);
    TAKE(Dy, 5 * N);
    TAKE(DyN, 3);
    TAKE(evGx, 9 * N);
    TAKE(evGu, 6 * N);
    TAKE(Q1, 9 * N);
    TAKE(Q2, 15 * N);
    TAKE(R1, 4 * N);
    TAKE(R2, 10 * N);
    TAKE(QN1, 9);
    TAKE(QN2, 9);
    TAKE(sbar, 3 * (N + 1));
    TAKE(Dx0, 3);
    TAKE(E, (size_t)N * (N + 1) / 2 * 6);
    TAKE(QDy, 3 * (N + 1));
    TAKE(H, n * n);
    TAKE(g, n);
    TAKE(lb, n);
    TAKE(ub, n);
    TAKE(dx, n);
    TAKE(dual, n);
#undef TAKE
    return s;
}

void orc_destroy(orc_nmpc *s)
{
    if (!s) return;
    free(s->pool);
    free(s);
}

real_t *orc_ptr(orc_nmpc *s, const char *name, int *len)
{
    const int N = s->N, n = 2 * N;
    int dummy;
    if (!len) len = &dummy;
#define F(nm, field, count)        \
    if (strcmp(name, nm) == 0) {   \
        *len = (int)(count);       \
        return s->field;           \
    }
    F("x", x, 3 * (N + 1))
    F("u", u, 2 * N)
    F("od", od, 3 * (N + 1))
    F("y", y, 5 * N)
    F("yN", yN, 3)
    F("W", W, 25 * N)
    F("WN", WN, 9)
    F("x0", x0, 3)
    F("lbValues", lbValues, n)
    F("ubValues", ubValues, n)
    F("d", d, 3 * N)
    F("Dy", Dy, 5 * N)
    F("DyN", DyN, 3)
    F("evGx", evGx, 9 * N)
    F("evGu", evGu, 6 * N)
    F("Q1", Q1, 9 * N)
    F("Q2", Q2, 15 * N)
    F("R1", R1, 4 * N)
    F("R2", R2, 10 * N)
    F("QN1", QN1, 9)
    F("QN2", QN2, 9)
    F("sbar", sbar, 3 * (N + 1))
    F("Dx0", Dx0, 3)
    F("E", E, N * (N + 1) / 2 * 6)
    F("QDy", QDy, 3 * (N + 1))
    F("H", H, n * n)
    F("g", g, n)
    F("lb", lb, n)
    F("ub", ub, n)
    F("dx", dx, n)
    F("dual", dual, n)
    F("rk_kkk", rk_kkk, 6)
#undef F
    *len = 0;
    return NULL;
}

double orc_time_rti(orc_nmpc *s, int iters)
{
    const int N = s->N, n = 2 * N;
    const size_t nx = 3 * (size_t)(N + 1), nu = 2 * (size_t)N;
    real_t *save = (real_t *)malloc((nx + nu + n) * sizeof(real_t));
    struct timespec t0, t1;
    int i;
    if (!save) return -1.0;
    memcpy(save, s->x, nx * sizeof(real_t));
    memcpy(save + nx, s->u, nu * sizeof(real_t));
    memcpy(save + nx + nu, s->dual, n * sizeof(real_t));
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < iters; ++i) {
        memcpy(s->x, save, nx * sizeof(real_t));
        memcpy(s->u, save + nx, nu * sizeof(real_t));
        memcpy(s->dual, save + nx + nu, n * sizeof(real_t));
        orc_preparation_step(s);
        orc_feedback_step(s);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(save);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
