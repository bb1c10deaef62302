/*
 * alore_wb.h -- C ABI of the whole-body NMPC class: B2 quadruped + Z1 arm, floating base + 18 actuated joints
 * (BASELINE.json configs[2]; SURVEY.md section 8(a) row A-RB).
 *
 * THE REFERENCE HAS NO IMPLEMENTATION OF THIS CLASS: no RNEA / ABA / CRBA and no whole-body OCP anywhere in the
 * repository; its only whole-body artefact is the URDF planning_ddr_opt/utils/simulator/urdf/b2_z1/urdf/
 * b2_plus_z1.urdf (12 leg joints :256-1416, arm joint1..6 :1527-1705, base mass :13-14), which
 * tools/gen_b2z1_model.py turns into the number table the kernels are compiled with.  PARITY UNPINNED: the checker is
 * oracle/wb_oracle.py (float64, 6-D spatial algebra), itself pinned by physics identities (tests/test_wb_oracle.py).
 * What would bind here in the reference is therefore a NEW controller class next to nmpc_controller's MpcWrapper
 * (include/nmpc_controller/mpc_wrapper.h:35-87: setReferencePose / setTrajectory / solve / update / getStates /
 * getInputs) -- the entry points below keep that shape: weights, references, measured state in; one real-time
 * iteration per call; predicted states and inputs out.
 *
 * The OCP (builder-defined, stated here because no reference text exists):
 *   state    x = [p (3, world) | rpy (3, ZYX) | joint angles (18) | omega_base (3) | v_base (3, base frame) | joint rates (18)]   nx = 48
 *   input    u = [joint torques (18) | foot forces (4 x 3, world frame, at the foot points FL FR RL RR)]                        nu = 30
 *   dynamics semi-implicit Euler:  v+ = v + dt a(q, v, u),  q+ = q + dt G(q) v+,   a = M(q)^-1 ([0; tau] - RNEA(q, v, 0, f))
 *   cost     sum_k 1/2 (x_k - xref_k)' Q (..) + 1/2 (u_k - uref_k)' R (..)  + 1/2 (x_N - xref_N)' QN (..),  Q, R, QN diagonal
 *   bounds   |tau_i| <= effort_i of the URDF, handled in the Riccati sweep as in control-limited DDP (Tassa et al. 2014,
 *            one projection per stage): where the feed-forward step of a stage would leave the limit, that input is
 *            clamped, the others are re-solved against it and its feedback row is zero; the applied inputs are clipped
 *            as well.  Contact consistency of the feet is the caller's job through xref / uref.
 * One real-time iteration: linearise the dynamics about the current iterate (A_k, B_k, defects), solve the LQ problem
 * by a Riccati sweep (float32 MFMA on the 48 x 48 / 48 x 30 blocks), apply the full step.
 */
#ifndef ALORE_WB_H
#define ALORE_WB_H

#ifdef __cplusplus
extern "C" {
#endif

#define ALORE_WB_OK 0
#define ALORE_WB_E_INVALID (-1)
#define ALORE_WB_E_NO_DEVICE (-2)
#define ALORE_WB_E_HIP (-3)
#define ALORE_WB_E_NOMEM (-4)

#define ALORE_WB_NQ 24
#define ALORE_WB_NV 24
#define ALORE_WB_NX 48
#define ALORE_WB_NU 30
#define ALORE_WB_NJ 18

typedef struct alore_wb_solver *alore_wb_handle;

typedef struct alore_wb_config {
    int horizon;      /* N (<= 32) */
    double dt;
    int device;
    int max_problems; /* B */
} alore_wb_config;

void alore_wb_default_config(alore_wb_config *c); /* N = 20, dt = 0.01 */
int alore_wb_create(const alore_wb_config *cfg, alore_wb_handle *out); /* fails without a GPU: there is no CPU path */
int alore_wb_destroy(alore_wb_handle h);
const char *alore_wb_last_error(alore_wb_handle h);
/* the model table the library was compiled with: masses [19], joint limits lower / upper / effort [18]; any may be NULL */
int alore_wb_model_info(double *masses, double *lower, double *upper, double *effort);

/* LDS bytes per workgroup of the two kernels (residency: 8 stage wavefronts / 3 Riccati workgroups per CU need
 * <= 20480 / <= 53760 bytes; a test guards both) */
int alore_wb_kernel_info(int *stage_lds_bytes, int *riccati_lds_bytes);

/* ---- rigid-body dynamics for n independent evaluation points (HOST pointers; synchronous) ----
 * q [n][24], v [n][24], a [n][24], f [n][12] (NULL = no foot forces), gravity 0/1 -> tau [n][24] */
int alore_wb_rnea(alore_wb_handle h, int n, const double *q, const double *v, const double *a, const double *f, int gravity,
                  double *tau);
/* mass matrix M [n][24][24] (from unit accelerations) and forward dynamics a [n][24] for inputs u [n][30]; either
 * output may be NULL */
int alore_wb_forward_dynamics(alore_wb_handle h, int n, const double *q, const double *v, const double *u, double *M,
                              double *a);

/* the same accelerations by the articulated-body algorithm (O(n), no mass matrix): u [n][30] -> a [n][24] */
int alore_wb_aba(alore_wb_handle h, int n, const double *q, const double *v, const double *u, double *a);

/* ---- the OCP ---- */
/* diagonal weights shared by the batch: Q [48], R [30], QN [48] */
int alore_wb_set_weights(alore_wb_handle h, const double *Q, const double *R, const double *QN);
/* torque limits inside the Riccati sweep (default 1); 0: the sweep is unconstrained and only the applied inputs are clipped */
int alore_wb_set_torque_limits(alore_wb_handle h, int enable);
/* measured states x0 [B][48], references xref [B][N+1][48], uref [B][N][30] (HOST pointers) */
int alore_wb_set_problem(alore_wb_handle h, int B, const double *x0, const double *xref, const double *uref);
/* what one control tick moves: the measured states in (x0 [B][48]), the iterate shifted by one stage on the device (the
 * last stage repeated, asynchronous on `stream`), the first inputs out (u0 [B][30]) */
int alore_wb_set_x0(alore_wb_handle h, int B, const double *x0);
int alore_wb_shift_iterate(alore_wb_handle h, int B, void *stream);
int alore_wb_get_first_input(alore_wb_handle h, int B, double *u0);
/* iterate x [B][N+1][48], u [B][N][30] */
int alore_wb_set_iterate(alore_wb_handle h, int B, const double *x, const double *u);
int alore_wb_get_iterate(alore_wb_handle h, int B, double *x, double *u);
/* linearisation about the current iterate, copied out for inspection: A [B][N][48][48], Bm [B][N][48][30],
 * next [B][N][48] = f(x_k, u_k); any may be NULL */
int alore_wb_linearize(alore_wb_handle h, int B, double *A, double *Bm, double *next);
/* Contact constraints of the 12 foot-force inputs (world frame, flat ground), the "constraint Jacobians" of this class:
 *   - a foot that is not in contact at a stage carries no force there (f = 0): alore_wb_set_contact_schedule,
 *     stance [B][N][4] bytes (1 = in contact), HOST pointer, NULL = every foot at every stage (the default); the
 *     schedule belongs to the horizon as set and is not moved by alore_wb_shift_iterate;
 *   - a stance foot can only push, inside the linearised friction cone (pyramid): fz >= 0, |fx| <= mu fz, |fy| <= mu fz.
 * They are handled INSIDE the Riccati sweep exactly like the torque boxes (control-limited DDP, one projection per
 * stage): the unconstrained feed-forward step of the stage is computed, the force components that leave their bounds
 * are clamped -- the tangential bounds from the projected normal force of the same stage --, the free inputs are
 * re-solved against the clamped ones, the gain rows of clamped inputs are zero; the applied forces satisfy the
 * constraints exactly.  Off by default (enable = 0: forces are free inputs, as in round 2).  The kinematic side -- stance
 * feet do not move -- is alore_wb_set_contact_penalty below (velocity level, as a penalty on J_c v) or, as hard equality
 * rows, alore_wb_set_contact_rows. */
int alore_wb_set_contact_constraints(alore_wb_handle h, int enable, double mu);
/* How the inequality constraints -- torque boxes (alore_wb_set_torque_limits), and with alore_wb_set_contact_constraints the friction
 * pyramid of the stance feet (|fx|, |fy| <= mu fz, hence fz >= 0) and f = 0 of the feet in the air -- are handled:
 *   mode 0 (default)  one projection per stage inside the Riccati sweep, as described above: one sweep, an approximation of the
 *                     constrained LQ solution (exact when nothing is clamped);
 *   mode 1            EXACT: a primal active-set iteration (Nocedal & Wright, Alg. 16.3) around the unconstrained sweep.  The rows of the
 *                     working set are eliminated from the stage data (an input at a bound is a constant, a tangential force on a face
 *                     of the pyramid follows the normal force: its column of B joins the column of fz, its weight mu^2 R the weight of
 *                     fz), the sweep solves the reduced problem -- the minimiser on the face of the working set --, the current feasible
 *                     step moves towards it along the straight line as far as the first row outside the working set allows (ratio test,
 *                     float64; the state step is the same combination of the two closed-loop state steps, nothing is rolled out
 *                     open-loop), that row joins the working set; at a face minimiser the gradient of the cost with respect to every
 *                     input (float64, from the costates) gives the multipliers of the held rows and the worst wrong-signed one leaves.
 *                     One row in or out per sweep and problem, until every problem has a face minimiser with correct signs or
 *                     `max_sweeps` (1 .. 4096) sweeps have run: as many sweeps as the working set of the slowest problem of the batch
 *                     differs from its starting guess (the rows active at the iterate).  This is the reference mode -- the minimiser of
 *                     the inequality-constrained LQ problem (tests/test_wb_gpu.py: against a float64 active-set solve of the condensed QP
 *                     built from the oracle's own linearisation) --, not the control-rate one.  The working set is kept from one real-time
 *                     iteration to the next and reset by alore_wb_set_iterate / alore_wb_shift_iterate (a shifted horizon starts
 *                     from the rows active at its iterate again: the set is not shifted with it).  alore_wb_rti then BLOCKS THE HOST
 *                     -- one device-to-host read of the open-problem counter and one stream synchronisation per sweep, up to
 *                     `max_sweeps` of them, all inside the `riccati` span of alore_wb_last_timing -- and CANNOT be captured into a
 *                     hipGraph; it runs without contact rows, contact penalty and refinement.
 * alore_wb_working_set_info: sweeps of the last real-time iteration; per problem 1 if it was still open when the sweeps ran out
 * (0 = solved); the working set itself, codes [B][N][32] (0 free, 1 lower, 2 upper bound, 3 / 4 on the + / - face of the pyramid, 5 zero
 * force); the input gradients [B][N][30] at the last face minimiser (the multipliers).  Any pointer may be NULL. */
int alore_wb_set_constraint_mode(alore_wb_handle h, int mode, int max_sweeps);
int alore_wb_working_set_info(alore_wb_handle h, int B, int *sweeps, int *changed, unsigned char *ws, double *input_gradients);
int alore_wb_set_contact_schedule(alore_wb_handle h, int B, const unsigned char *stance);
/* Contact consistency of the stance feet through the contact Jacobian J_c(q) (12 x 24: world-frame velocity of the four foot
 * points per unit generalized velocity -- the transpose of the map that takes the foot forces into the dynamics): the stage
 * cost gains  1/2 rho | J_c(q_k) v_k |^2  over the feet in contact at stage k (contact schedule above), i.e. the constraint
 * J_c v = 0 (stance feet do not move) as a quadratic penalty, Gauss-Newton: rho J_c' J_c on the velocity block of the stage
 * Hessian (added to the cost-to-go on the matrix cores), rho J_c' (J_c v_k) on its gradient; J_c is evaluated at the current
 * iterate by the linearisation kernel (its foot-force columns).  rho = 0 (default): off, the feet are held by the posture cost
 * only.  The violation |J_c v| of the iterate shrinks like 1 / rho; rho of 1e3 .. 1e4 keeps it below a millimetre per second
 * on the test motions without hurting the conditioning of the float32 sweep. */
int alore_wb_set_contact_penalty(alore_wb_handle h, double rho);
/* Hard contact rows (default off): a foot in contact at stage k (contact schedule) does not move at the end of that step,
 *     J_c(q_k) v_{k+1} = 0
 * as EQUALITY rows of the LQ problem of every real-time iteration -- the velocity-level form of J_c qdd + Jdot_c qd = 0 under
 * the semi-implicit Euler step (J_c qdd + J_c v / dt = 0: the acceleration row with its stabilisation term), J_c evaluated at
 * the current iterate by the linearisation kernel.  The block of the rows with respect to the 12 foot forces,
 * J_c (dt M^-1 J_c'), is symmetric positive definite, so the rows are solved for the forces: df = F_x dx + F_tau dtau + f0,
 * the forces leave the dynamics (A + B_f F_x, B_tau + B_f F_tau: the null space of the rows, parametrised by the joint
 * torques), the Riccati sweep solves an ordinary LQ problem in (dx, dtau) and the forces follow from its solution.  In this
 * mode the foot forces are what the contacts need: their entries of R and uref are not used; a foot in the air carries f = 0.
 * Replaces the contact penalty (its rho is ignored while the rows are on) and the force constraints of
 * alore_wb_set_contact_constraints (unilateral / friction limits on the eliminated forces are not enforced); torque limits
 * keep working.  oracle/wb_oracle.py: solve_lq_contact_rows is the float64 KKT restatement (tests/test_wb_gpu.py). */
int alore_wb_set_contact_rows(alore_wb_handle h, int enable);
/* `steps` (0 .. 4, default 0 = off) steps of iterative refinement of the LQ solution.  The Riccati sweep runs in float32 on the matrix cores
 * and loses digits where the stage Hessian is stiff -- a contact penalty of rho = 2000 next to velocity weights of 1 leaves
 * 1.5e-3 of relative error in the step.  With refinement the residuals of the LQ optimality system at that solution are
 * formed in float64 (defect residuals, stage and terminal gradients, the penalty Hessian included) and go through the same
 * float32 sweep as right-hand sides; the sum of the solutions is applied (every step multiplies the error by the one-pass error, at one more
 * Riccati sweep).  Acts only while nothing is clamped inside the sweep (alore_wb_set_torque_limits(h, 0), no contact
 * constraints): a clamped sweep would have to be repeated with its clamp set held, which is not built. */
int alore_wb_set_refinement(alore_wb_handle h, int steps);
/* n_iter real-time iterations (linearise + Riccati + step) for B problems; asynchronous on `stream`.
 * Stream contract of this header: only alore_wb_rti and alore_wb_shift_iterate enqueue on the caller's stream (which
 * may be a non-blocking one).  Every entry point that moves data between host and device (set_problem, set_x0,
 * set_iterate, get_first_input, get_iterate, last_step, status, linearize) first waits for ALL work on the device
 * (hipDeviceSynchronize) and has completed its copy when it returns, so a tick
 *   set_x0 -> rti(stream) -> get_first_input -> shift_iterate(stream)
 * is ordered on any stream without further synchronisation by the caller. */
int alore_wb_rti(alore_wb_handle h, int B, int n_iter, void *stream);
/* the LQ step of the last iteration: dx [B][N+1][48], du [B][N][30] (before clipping), HOST pointers; synchronises */
int alore_wb_last_step(alore_wb_handle h, int B, double *dx, double *du);
/* per problem: 0 the last iteration was applied, 1 its step was not finite (indefinite weights, overflow) and the iterate
 * was left as it was; synchronises */
int alore_wb_status(alore_wb_handle h, int B, int *status);
/* kernel times of the last alore_wb_rti call in ms (linearisation, Riccati), measured with HIP events */
int alore_wb_last_times(alore_wb_handle h, float *linearize_ms, float *riccati_ms);

#ifdef __cplusplus
}
#endif
#endif
