/*
 * alore_acado_compat.h -- single-instance compatibility ABI: the symbols the reference's
 * Tracked_nmpc::MpcWrapper links against, served by the GPU engine (B = 1).
 *
 * The reference solver is a set of `extern "C"` functions working on two process-global structs that
 * the CALLER defines (planning_ddr_opt/nmpc_controller/src/mpc_wrapper.cpp:29-30) and the solver
 * `extern`s (UAV_CAR_model/build/quadrotor_mpc_codegen/acado_common.h:340-341).  libalore_acado_compat.so
 * exports the same function symbols and expects the caller to define the same two data symbols with
 * the layout below, which restates member order and sizes of acado_common.h:104-162 (ACADOvariables)
 * and :170-257 (ACADOworkspace) for the generated horizon N = 50 (acado_common.h:65).
 *
 * Served from the GPU: acado_initializeSolver, acado_initializeNodesByForwardSimulation,
 * acado_preparationStep, acado_feedbackStep, acado_shiftStates, acado_shiftControls, acado_getKKT,
 * acado_getObjective, acado_getNWSR, acado_integrate, acado_rhs, acado_diffs, acado_getErrorString.
 * acadoWorkspace members kept up to date (what Tracked_nmpc::MpcWrapper and acado_solve() read):
 *   acado_preparationStep: d, evGx, evGu (acado_modelSimulation, CG/acado_solver.c:35-78), H and the state-independent part of g of
 *     the condensed QP (formed on the GPU by alore_nmpc_condense: what acado_condensePrep leaves, :365-891);
 *   acado_feedbackStep: Dx0, lb, ub, g (with the Dx0 terms: acado_condenseFdb, :893-960), x (= delta u), y (= dual);
 *   acado_solve(): solves the dense QP that is in H, g, lb, ub (alore_nmpc_dense_qp), primal into x, dual into y.
 * NOT refreshed -- the intermediates of the reference's O(N^2) condensing, which the stage-wise solver never forms and the
 * reference's wrapper never reads: E (d x_k / d u_j blocks), Q1, Q2, R1, R2, QN1, QN2 (weight slices, :171-211), Dy, DyN, QDy
 * (stacked residuals, :327-363), sbar, w1, w2, QGx, QGu and the state pieces of the integrator (rk_*).  They keep whatever the
 * caller put there.  ALORE_ACADO_DENSE_WORKSPACE=0 switches the H / g refresh off (then acado_solve() has nothing to solve).
 */
#ifndef ALORE_ACADO_COMPAT_H
#define ALORE_ACADO_COMPAT_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ACADO_N
#define ACADO_N 50
#endif
#define ACADO_NX 3
#define ACADO_NU 2
#define ACADO_NOD 3
#define ACADO_NY 5
#define ACADO_NYN 3
#define ACADO_QP_NV (ACADO_NU * ACADO_N)

typedef float real_t; /* acado_qpoases_interface.hpp:50 */

typedef struct ACADOvariables_ {
    int dummy;
    real_t x[(ACADO_N + 1) * ACADO_NX];
    real_t u[ACADO_N * ACADO_NU];
    real_t od[(ACADO_N + 1) * ACADO_NOD];
    real_t y[ACADO_N * ACADO_NY];
    real_t yN[ACADO_NYN];
    real_t W[ACADO_N * ACADO_NY * ACADO_NY];
    real_t WN[ACADO_NYN * ACADO_NYN];
    real_t x0[ACADO_NX];
    real_t lbValues[ACADO_QP_NV];
    real_t ubValues[ACADO_QP_NV];
} ACADOvariables;

typedef struct ACADOworkspace_ {
    real_t rhs_aux[13];
    real_t d[ACADO_N * ACADO_NX];
    real_t Dy[ACADO_N * ACADO_NY];
    real_t DyN[ACADO_NYN];
    real_t evGx[ACADO_N * ACADO_NX * ACADO_NX];
    real_t evGu[ACADO_N * ACADO_NX * ACADO_NU];
    real_t objValueIn[8];
    real_t objValueOut[5];
    real_t Q1[ACADO_N * ACADO_NX * ACADO_NX];
    real_t Q2[ACADO_N * ACADO_NX * ACADO_NY];
    real_t R1[ACADO_N * ACADO_NU * ACADO_NU];
    real_t R2[ACADO_N * ACADO_NU * ACADO_NY];
    real_t QN1[ACADO_NX * ACADO_NX];
    real_t QN2[ACADO_NX * ACADO_NYN];
    real_t sbar[(ACADO_N + 1) * ACADO_NX];
    real_t Dx0[ACADO_NX];
    real_t W1[6];
    real_t W2[6];
    real_t E[ACADO_N * (ACADO_N + 1) / 2 * ACADO_NX * ACADO_NU];
    real_t QDy[(ACADO_N + 1) * ACADO_NX];
    real_t w1[3];
    real_t w2[3];
    real_t H[ACADO_QP_NV * ACADO_QP_NV];
    real_t g[ACADO_QP_NV];
    real_t lb[ACADO_QP_NV];
    real_t ub[ACADO_QP_NV];
    real_t x[ACADO_QP_NV];
    real_t y[ACADO_QP_NV];
} ACADOworkspace;

/* defined by the caller, exactly like with the generated solver */
extern ACADOworkspace acadoWorkspace;
extern ACADOvariables acadoVariables;

int acado_initializeSolver(void);                       /* acado_solver.c:1079 */
void acado_initializeNodesByForwardSimulation(void);    /* :1292 */
int acado_preparationStep(void);                        /* :1057 */
int acado_feedbackStep(void);                           /* :1067 */
void acado_shiftStates(int strategy, real_t *const xEnd, real_t *const uEnd); /* :1314 */
void acado_shiftControls(real_t *const uEnd);           /* :1357 */
real_t acado_getKKT(void);                              /* :1373 */
real_t acado_getObjective(void);                        /* :1393 */
int acado_integrate(real_t *const rk_eta, int resetIntegrator); /* acado_integrator.c:261 */
int acado_solve(void);                                  /* acado_qpoases_interface.cpp:39 */
/* The model's symbolic functions (host, pure): in = [x y theta | u_r u_l | xv yr yl];
 * acado_rhs -> out[3] = (xdot, ydot, thetadot); acado_diffs -> out[15] = d rhs / d (x, y, theta, u_r, u_l), row major. */
void acado_rhs(const real_t *in, real_t *out);          /* acado_integrator.c:62 */
void acado_diffs(const real_t *in, real_t *out);        /* acado_integrator.c:84 */
int acado_getNWSR(void);                                /* :62 */
const char *acado_getErrorString(int error);            /* :67 */

/* not part of the reference ABI: choose the GPU (before acado_initializeSolver), release the engine */
void alore_acado_set_device(int device);
void alore_acado_shutdown(void);
/* acado_preparationStep / acado_feedbackStep leave the condensed QP (acadoWorkspace.H, g) behind like the reference's do: one
 * more launch and 4 (2N)^2 bytes over the bus per call.  A caller that never reads them (MpcWrapper does not) can switch that
 * off: 0 = H / g are not refreshed (acado_solve() then works on whatever the caller puts there), 1 = refreshed (default; the
 * environment variable ALORE_ACADO_DENSE_WORKSPACE=0 has the same effect as a call with 0). */
void alore_acado_dense_workspace(int enable);

#ifdef __cplusplus
}
#endif
#endif
