/*
 * oracle/ref_glue.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Glue that turns the reference's own ACADO-generated solver + vendored qpOASES
 * (compiled from where they lie under /root/reference by oracle/Makefile, never
 * copied into this repo) into a shared object `oracle/_ref/libacado_ref.so`
 * that tests can drive through ctypes.
 *
 * The reference solver `extern`s two process-global structs which the *caller*
 * must define (reference: planning_ddr_opt/nmpc_controller/src/mpc_wrapper.cpp:29-30,
 * declared at .../quadrotor_mpc_codegen/acado_common.h:340-341).  This file is
 * that caller: it defines the two globals and exports plain accessors so that
 * Python does not need to mirror the struct layouts.
 */
#include "acado_common.h"
#include "acado_auxiliary_functions.h"

ACADOvariables acadoVariables;
ACADOworkspace acadoWorkspace;

/* integrator stage memory: a never-reset global of the reference
 * (acado_integrator.c:37); exposed so a test can put it into a known state. */
extern real_t rk_kkk[6];

#define ACC(name, expr) real_t *ref_##name(void) { return (expr); }
ACC(x, acadoVariables.x)
ACC(u, acadoVariables.u)
ACC(od, acadoVariables.od)
ACC(y, acadoVariables.y)
ACC(yN, acadoVariables.yN)
ACC(W, acadoVariables.W)
ACC(WN, acadoVariables.WN)
ACC(x0, acadoVariables.x0)
ACC(lbValues, acadoVariables.lbValues)
ACC(ubValues, acadoVariables.ubValues)
ACC(ws_d, acadoWorkspace.d)
ACC(ws_Dy, acadoWorkspace.Dy)
ACC(ws_DyN, acadoWorkspace.DyN)
ACC(ws_evGx, acadoWorkspace.evGx)
ACC(ws_evGu, acadoWorkspace.evGu)
ACC(ws_Q1, acadoWorkspace.Q1)
ACC(ws_Q2, acadoWorkspace.Q2)
ACC(ws_R1, acadoWorkspace.R1)
ACC(ws_R2, acadoWorkspace.R2)
ACC(ws_QN1, acadoWorkspace.QN1)
ACC(ws_QN2, acadoWorkspace.QN2)
ACC(ws_sbar, acadoWorkspace.sbar)
ACC(ws_Dx0, acadoWorkspace.Dx0)
ACC(ws_E, acadoWorkspace.E)
ACC(ws_QDy, acadoWorkspace.QDy)
ACC(ws_H, acadoWorkspace.H)
ACC(ws_g, acadoWorkspace.g)
ACC(ws_lb, acadoWorkspace.lb)
ACC(ws_ub, acadoWorkspace.ub)
ACC(ws_x, acadoWorkspace.x)
ACC(ws_y, acadoWorkspace.y)
ACC(rk_kkk, rk_kkk)

int ref_N(void) { return ACADO_N; }
int ref_NX(void) { return ACADO_NX; }
int ref_NU(void) { return ACADO_NU; }
int ref_NOD(void) { return ACADO_NOD; }
int ref_NY(void) { return ACADO_NY; }
int ref_NYN(void) { return ACADO_NYN; }
int ref_sizeof_variables(void) { return (int)sizeof(ACADOvariables); }
int ref_sizeof_workspace(void) { return (int)sizeof(ACADOworkspace); }

/* zero both structs and the integrator memory (what MpcWrapper's ctor does for
 * acadoVariables, mpc_wrapper.cpp:36-41, plus rk_kkk for reproducibility) */
void ref_reset(void)
{
    int i;
    memset(&acadoVariables, 0, sizeof(acadoVariables));
    memset(&acadoWorkspace, 0, sizeof(acadoWorkspace));
    for (i = 0; i < 6; ++i) rk_kkk[i] = 0;
}

/* time `iters` RTI ticks (preparation + feedback) on the loaded problem,
 * restoring x,u,dual before every tick so each tick does identical work.
 * Returns seconds (monotonic clock). */
#include <time.h>
double ref_time_rti(int iters)
{
    static ACADOvariables v0;
    static real_t y0[ACADO_QP_NV];
    struct timespec t0, t1;
    int i, k;
    v0 = acadoVariables;
    for (k = 0; k < ACADO_QP_NV; ++k) y0[k] = acadoWorkspace.y[k];
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < iters; ++i) {
        acadoVariables = v0;
        for (k = 0; k < ACADO_QP_NV; ++k) acadoWorkspace.y[k] = y0[k];
        acado_preparationStep();
        acado_feedbackStep();
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
