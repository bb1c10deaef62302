/*
 * oracle/nmpc_oracle.h -- CPU restatement of the reference NMPC numerics.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product path
 * (alore_legged_manipulator_amd/) never does.
 *
 * Restates, horizon-generic (the reference is code-generated for N = 50 only)
 * and in IEEE float32 like the reference (acado_qpoases_interface.hpp:50),
 * the algorithm of
 *   P/nmpc_controller/UAV_CAR_model/build/quadrotor_mpc_codegen/acado_integrator.c
 *   P/nmpc_controller/UAV_CAR_model/build/quadrotor_mpc_codegen/acado_solver.c
 *   P/nmpc_controller/UAV_CAR_model/build/quadrotor_mpc_codegen/acado_qpoases_interface.cpp
 *   P/nmpc_controller/externals/qpoases/SRC/QProblemB.cpp  (box-QP path only)
 * with P = /root/reference/planning_ddr_opt.  Parity pinned: checked against
 * the compiled reference (oracle/_ref) and against tests/golden/ fixtures that
 * were generated from it (tests/test_oracle_golden.py, oracle/gen_golden.py).
 */
#ifndef NMPC_ORACLE_H
#define NMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 3
#define ORC_NU 2
#define ORC_NOD 3
#define ORC_NY 5
#define ORC_NYN 3

typedef struct orc_nmpc orc_nmpc;

/* one single-instance solver with horizon N and sampling time dt; all state
 * lives in the handle (the reference keeps it in process globals). */
orc_nmpc *orc_create(int N, double dt);
void orc_destroy(orc_nmpc *s);
int orc_N(const orc_nmpc *s);

/* named access to every member of the reference's ACADOvariables /
 * ACADOworkspace (acado_common.h:104-257): "x","u","od","y","yN","W","WN",
 * "x0","lbValues","ubValues","d","Dy","DyN","evGx","evGu","Q1","Q2","R1","R2",
 * "QN1","QN2","sbar","Dx0","E","QDy","H","g","lb","ub","dx"(=workspace x),
 * "dual"(=workspace y),"rk_kkk".  Returns NULL for unknown names; *len gets
 * the element count. */
float *orc_ptr(orc_nmpc *s, const char *name, int *len);

/* model + integrator (acado_integrator.c:62-123, 261-449) */
void orc_rhs(const float *in8, float *out3);
void orc_diffs(const float *in8, float *out15);
int orc_integrate(orc_nmpc *s, float *rk_eta23, int reset);

/* RTI phases (acado_solver.c) */
int orc_model_simulation(orc_nmpc *s);    /* :35-78   */
void orc_evaluate_objective(orc_nmpc *s); /* :171-211 */
void orc_condense_prep(orc_nmpc *s);      /* :327-363 */
void orc_condense_fdb(orc_nmpc *s);       /* :365-891 */
int orc_solve_qp(orc_nmpc *s);            /* acado_qpoases_interface.cpp:39-60 */
void orc_expand(orc_nmpc *s);             /* :893-1055 */

/* RTI API (acado_solver.c:1057-1450) */
int orc_initialize_solver(orc_nmpc *s);
void orc_initialize_nodes_by_forward_simulation(orc_nmpc *s);
int orc_preparation_step(orc_nmpc *s);
int orc_feedback_step(orc_nmpc *s);
void orc_shift_states(orc_nmpc *s, int strategy, const float *xEnd, const float *uEnd);
void orc_shift_controls(orc_nmpc *s, const float *uEnd);
float orc_get_kkt(orc_nmpc *s);
float orc_get_objective(orc_nmpc *s);
int orc_get_nwsr(const orc_nmpc *s);

/* stand-alone dense box-QP  min 1/2 x'Hx + g'x  s.t. lb <= x <= ub  by the
 * online active-set homotopy of qpOASES' QProblemB::init (QProblemB.cpp:285-296).
 * H is n x n row-major; yOpt (may be NULL) seeds the initial working set by
 * sign (QProblemB.cpp:1010-1036).  *nWSR: in = max working-set changes, out =
 * performed.  Returns the qpOASES returnValue as int (0 = SUCCESSFUL_RETURN). */
int orc_qpb_solve(int n, const float *H, const float *g, const float *lb, const float *ub,
                  const float *yOpt, int *nWSR, float *x_out, float *y_out);

/* bench helper: `iters` x (restore x,u,dual; preparation; feedback) on the
 * loaded problem; returns seconds. */
double orc_time_rti(orc_nmpc *s, int iters);

#ifdef __cplusplus
}
#endif
#endif
