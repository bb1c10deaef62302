/*
 * alore_ltv_mpc.h -- C ABI of the batched linear time-varying MPC (the controller planner_sim.launch actually
 * starts: planning_ddr_opt/mpc_controller, node `mpc`).
 *
 * One reference control tick (MpcController::CmdCallback, mpc_controller/src/mpc.cpp:131-216) is
 *   getRefPoints + smooth_yaw (:634-690, 538-567)  ->  getCmd (:569-614): repeat { predictMotion (:259-269),
 *   solveMPCV (:304-535: a QP over 3 (T - d) states and 2 (T - d) inputs with unicycle dynamics linearised about
 *   the rollout, input boxes and rate limits, handed to OSQP) }  ->  cmd = output(:, delay_num).
 * This library runs that tick for B robots in one launch: one GPU thread per robot walks the stages, the per-stage
 * records of all robots interleaved in memory so that the wavefront's accesses coalesce.  The QP is solved EXACTLY
 * (float64) by a working-set Riccati method: for a guess of which inputs sit on a box bound or on a rate limit the
 * equality-constrained problem is one backward / forward sweep over the stages (state augmented by the previous
 * input, so that rate terms and rate limits are stage-local); multipliers and violations update the guess until
 * it reproduces itself (4 - 25 sweeps cold, 1 - 3 warm).  The reference stops its relinearisation loop on a 9.7 ms
 * wall clock, i.e. after a machine-dependent number of passes; here the count is an argument.
 * Parity: the QP arithmetic of the reference lives in OSQP (external, un-vendored, version unpinned): PARITY
 * UNPINNED; checked against an independent dense solve of the same matrices (oracle/ltv_mpc_oracle.py).
 */
#ifndef ALORE_LTV_MPC_H
#define ALORE_LTV_MPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define ALORE_LTV_OK 0
#define ALORE_LTV_E_INVALID (-1)
#define ALORE_LTV_E_NO_DEVICE (-2)
#define ALORE_LTV_E_HIP (-3)
#define ALORE_LTV_E_NOMEM (-4)

typedef struct alore_ltv_solver *alore_ltv_handle;

/* /mpc/... parameters (mpc.cpp:10-30; values: mpc_controller/config/mpc3ms.yaml, plan_manager/config/car3ms.yaml) */
typedef struct alore_ltv_config {
    double dt;
    int predict_steps; /* T  (<= 64) */
    int delay_num;     /* d  (< T) */
    double matrix_q[4];  /* x, y, v, theta */
    double matrix_r[2];  /* v, omega */
    double matrix_rd[2]; /* rate of v, omega */
    double max_vel, min_vel, max_omega, max_acc, max_domega;
    int max_sweeps; /* cap on working-set sweeps per QP (<= 0: 64) */
} alore_ltv_config;

void alore_ltv_default_config(alore_ltv_config *c);
int alore_ltv_create(const alore_ltv_config *cfg, int device, int max_robots, alore_ltv_handle *out);
int alore_ltv_destroy(alore_ltv_handle h);
const char *alore_ltv_last_error(alore_ltv_handle h);

/* References of the tick for B robots, HOST pointers: xref [B][T][3] (x, y, theta after smooth_yaw),
 * dref [B][T][2] (v, omega).  (getRefPoints output; alore_ltv_refs_from_store fills them on the device instead.) */
int alore_ltv_set_refs(alore_ltv_handle h, int B, const double *xref, const double *dref, void *stream);
/* The same from the device trajectory store of an NMPC handle (include/alore_nmpc.h: alore_nmpc_refs_*), i.e.
 * getRefPoints + smooth_yaw on the GPU: est [B][3] HOST (x, y, theta); at_goal [B] HOST or NULL.
 * `nmpc` is an alore_nmpc_handle on the same device (passed as void* to keep the headers independent). */
int alore_ltv_refs_from_store(alore_ltv_handle h, void *nmpc, int B, double now, const double *est, int *at_goal, void *stream);

/* getCmd for B robots: now_state [B][3] HOST (x, y, theta; the reference zeroes v, omega in its odometry callback),
 * n_relin passes of predictMotion + solveMPCV.  The previous output (2 x T per robot, warm start of the rollout) and
 * the delay buffer live in the handle, as in the reference's members `output`, `output_buff`; reset = 1 zeroes them
 * first (a freshly constructed controller).  Asynchronous on `stream`. */
int alore_ltv_get_cmd(alore_ltv_handle h, int B, const double *now_state, int n_relin, int reset, void *stream);
/* results: output [B][T][2] (v, omega per step; the command is row delay_num), xopt [B][T+1][3] (predictMotion(xopt),
 * :271-302), sweeps [B] (working-set sweeps of the last QP), status [B] (0 ok, 1 sweep cap reached); any may be NULL.
 * Synchronises the stream. */
int alore_ltv_results(alore_ltv_handle h, int B, double *output, double *xopt, int *sweeps, int *status, void *stream);
/* what a control tick needs: cmd [B][2] = output(:, delay_num) per robot (mpc.cpp:169-172) and the status; 16 bytes per robot
 * cross the bus instead of the whole prediction.  Synchronises the stream. */
int alore_ltv_commands(alore_ltv_handle h, int B, double *cmd, int *status, void *stream);
/* one control tick in one call (CmdCallback, mpc.cpp:142-175): states in, getCmd, cmd [B][2] and status [B] (may be NULL)
 * out -- one upload, one launch, one download, one wait on the stream */
int alore_ltv_tick(alore_ltv_handle h, int B, const double *now_state, int n_relin, int reset, double *cmd, int *status,
                   void *stream);
/* overwrite the stored previous output / delay buffer (tests): output [B][T][2], buff [B][d][2] */
int alore_ltv_set_state(alore_ltv_handle h, int B, const double *output, const double *buff, void *stream);

#ifdef __cplusplus
}
#endif
#endif
