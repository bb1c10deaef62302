/*
 * alore_backend.h -- C ABI of the MI355X-native batched back_end trajectory optimiser.
 *
 * Drop-in boundary for `bool MSPlanner::minco_plan(const FlatTrajData&)`
 * (reference: planning_ddr_opt/back_end/include/back_end/optimizer.h:207, src/optimizer.cpp:169-220; input
 * struct planning_ddr_opt/front_end/include/front_end/traj_representation.h:46-58; results read back through
 * get_current_iniState / finState / Innerpoints / finalpieceTime / iniStateXYTheta, optimizer.h:253-268, the
 * fields plan_manager serialises into carstatemsgs/Polynome, plan_manager.hpp:784-831).
 * One call plans `count` independent FlatTrajData problems -- Monte-Carlo start poses x object classes --
 * in one kernel launch, one wavefront per problem: differential-flat (yaw, arc length) minimum-jerk
 * spline, Simpson-integrated pose, smoothed-L1 penalties, ESDF clearance, L-BFGS with the Lewis-Overton
 * line search, augmented-Lagrangian terminal constraint, final collision check and retry, all in float64
 * like the reference.  Plain pointers and sizes only; nothing throws across this boundary; there is no CPU
 * path behind it (alore_backend_create fails without a GPU).
 *
 * Functions return 0 or a negative ALORE_BE_E_* code.  A handle is bound to one GPU and is not
 * thread-safe; `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef ALORE_BACKEND_H
#define ALORE_BACKEND_H

#ifdef __cplusplus
extern "C" {
#endif

#define ALORE_BE_OK 0
#define ALORE_BE_E_INVALID (-1)
#define ALORE_BE_E_NO_DEVICE (-2)
#define ALORE_BE_E_HIP (-3)
#define ALORE_BE_E_NOMEM (-4)
#define ALORE_BE_E_UNSUPPORTED (-5) /* more pieces than the handle was created for */

typedef struct alore_backend_planner *alore_backend_handle;

/* lbfgs::lbfgs_parameter_t (planning_ddr_opt/back_end/include/gcopter/lbfgs.hpp:14-129) */
typedef struct alore_lbfgs_param {
    int mem_size, past, max_iterations, max_linesearch;
    double g_epsilon, delta, min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
} alore_lbfgs_param;

/* What MSPlanner reads from the parameter server (optimizer.cpp:17-166: back_end/config/global_planning3ms.yaml,
 * planner_sim.launch:41-46) and from plan_manager's Config (plan_manager/config/car3ms.yaml).
 * alore_backend_default_config fills the values of those files. */
typedef struct alore_backend_config {
    double max_vel, min_vel, max_acc, max_omega, max_domega, max_cen_acc;
    int direct_v_omega;                                                            /* if_directly_constrain_v_omega */
    double w_time, w_acc, w_domega, w_collision, w_moment, w_mean_time, w_cen_acc; /* penaltyWeights */
    double p_time, p_bigpath, p_moment, p_mean_time, p_acc, p_domega;              /* PathpenaltyWeights */
    double energy_w[2];                                                            /* energyWeights (yaw, s) */
    double smooth_eps, safe_dis, final_min_safe_dis;
    int final_check_num, safe_replan_max;
    double mean_lo, mean_hi;
    int sparse_res;  /* sparseResolution; this build supports 8 (17 Simpson nodes per piece) */
    int n_check;     /* body check points (<= 8) */
    double check_pts[8][2];
    double icr_xv;
    int standard_diff;
    double lam0[2], rho0[2], rho_max[2], gamma[2], tol;
    double cut_lam0[2], cut_rho0[2], cut_rho_max[2], cut_gamma[2], cut_tol;
    alore_lbfgs_param path_lbfgs;
    int shot_path_past;
    double shot_path_horizon;
    alore_lbfgs_param lbfgs;
    int max_alm_rounds; /* cap on the reference's unbounded `while (ros::ok())` ALM loop (optimizer.cpp:376) */
} alore_backend_config;

/* FlatTrajData, host pointers; M = n_pieces = len(UnOccupied_traj_pts) + 1 */
typedef struct alore_flat_traj {
    int n_pieces;
    const double *traj_pts;  /* (M-1) x 3: yaw, s, t (t unused by the optimiser) */
    double init_T;           /* UnOccupied_initT */
    const double *positions; /* (M-1) x 3: x, y, yaw of UnOccupied_positions */
    double start_state[2][3], final_state[2][3]; /* [yaw | s][p v a] */
    double start_xytheta[3], final_xytheta[3];
    int if_cut;
} alore_flat_traj;

/* per-problem outcome; `ok` is minco_plan's return value */
typedef struct alore_backend_status {
    int ok;         /* 1: planned and collision-free */
    int attempts;   /* optimiser passes (collision retries with 0.75 x time weight, optimizer.cpp:179-200) */
    int alm_rounds; /* of the last pass */
    int evals;      /* cost-callback evaluations, all passes */
    int lbfgs_ret;  /* lbfgs return code of the last stage-2 call (lbfgs.hpp:135-184) */
    int path_ret;   /* of the pre-processing stage */
    int collision;  /* final check of the last pass */
    int n_pieces;
    double cost, xy_err[2], min_dist, tail_s;
} alore_backend_status;

void alore_backend_default_config(alore_backend_config *c);
int alore_backend_create(const alore_backend_config *cfg, int device, int max_pieces, int max_problems,
                         alore_backend_handle *out);
int alore_backend_destroy(alore_backend_handle h);
const char *alore_backend_last_error(alore_backend_handle h);

/* plan_env::SDFmap as the optimiser queries it (sdf_map.cpp:760-863): double grid, dist[ix * ny + iy], cell
 * centres at ((i + 0.5) res + lo); copied to the device and kept until replaced */
int alore_backend_set_map(alore_backend_handle h, const double *dist, int nx, int ny, double x_lo, double y_lo, double res);
/* The same map built on the device from the occupancy grid: SDFmap::updateESDF2d (sdf_map.cpp:618-681, fillESDF
 * :683-714).  grid [nx][ny] holds the reference's cell states (0 unknown, 1 unoccupied, 2 occupied; sdf_map.h:98); the
 * field is recomputed inside the window (odom_x, odom_y) +- detection_range, cells outside it keep their values
 * (DBL_MAX after the first call with a new geometry, as in sdf_map.h:160).  dist_out (HOST, [nx][ny]) may be NULL. */
int alore_backend_build_esdf(alore_backend_handle h, const unsigned char *grid, int nx, int ny, double x_lo, double y_lo, double res,
                             double odom_x, double odom_y, double detection_range, double *dist_out);

/* upload `count` problems (<= max_problems) into the handle's device slots 0..count-1 */
int alore_backend_set_problems(alore_backend_handle h, int count, const alore_flat_traj *problems, void *stream);
/* MSPlanner::minco_plan for every uploaded problem, one launch; results stay on the device */
int alore_backend_plan(alore_backend_handle h, int count, void *stream);
/* results of the last plan: status[count]; inner [count][(P-1)*2] (yaw, s), T [count][P],
 * coef [count][P*12] (coefficient of t^q of piece i, dimension d at ((6 i + q) * 2 + d)); any pointer may be NULL;
 * P = max_pieces.  Synchronises the stream. */
int alore_backend_results(alore_backend_handle h, int count, alore_backend_status *status, double *inner, double *T,
                          double *coef, void *stream);

/* device pointers of the result slabs (for consumers on the same GPU, e.g. alore_nmpc_refs_set_from_backend):
 * same layouts as alore_backend_results; valid until the handle is destroyed */
typedef struct alore_backend_device_view {
    int max_pieces;
    const int *n_pieces;        /* [B] */
    const double *inner;        /* [B][(P-1)*2] */
    const double *T;            /* [B][P] */
    const double *coef;         /* [B][P*12] */
    const double *head;         /* [B][6]  start_state as uploaded: [d][p v a] */
    const double *tail;         /* [B][6]  final_state with the optimised arc length */
    const double *start_xytheta; /* [B][3] */
    const int *ok;              /* [B] */
} alore_backend_device_view;
int alore_backend_device_results(alore_backend_handle h, alore_backend_device_view *out);

/* ---- pieces of the optimiser, exposed for parity tests against the CPU oracle ------------------------ */
/* one cost-callback evaluation per problem: stage 1 = costFunctionCallbackPath (optimizer.cpp:1272-1317),
 * 2 = costFunctionCallback (:631-692).  x, grad: [count][3P-1 stride = 3 * max_pieces]; lam, rho: [count][2] or
 * NULL (config values); cost [count]; xy_err [count][2].  Host pointers; synchronises. */
int alore_backend_eval(alore_backend_handle h, int count, int stage, const double *x, const double *lam, const double *rho,
                       double safe_dis, double time_weight, double *cost, double *grad, double *xy_err, void *stream);
/* lbfgs_optimize on one stage from x (in/out), at most max_iter iterations (0 = the configured limit);
 * ret/iters/evals [count] */
int alore_backend_lbfgs(alore_backend_handle h, int count, int stage, double *x, const double *lam, const double *rho,
                        double safe_dis, double time_weight, int max_iter, double *cost, int *ret, int *iters, int *evals,
                        double *xy_err, void *stream);

/* duration of the last alore_backend_plan launch in ms (HIP events on its stream; synchronises) */
int alore_backend_last_plan_ms(alore_backend_handle h, float *ms);

/* MSPlanner::get_the_predicted_state (optimizer.cpp:1108-1188) and get_the_predicted_state_and_path (:1190-1262) on the
 * plans of the last alore_backend_plan: the pose reached at time[b] by Simpson integration in steps of `resolution`
 * (trajPredictResolution) from start_time[b] (NULL: 0) and start_xytheta[b] (NULL: the plan's start pose), the flat
 * derivatives there (vaj = v, a, j of the arc length; oaj = omega, alpha, jerk of the yaw) and the reference's
 * if_forward flag.  HOST pointers, [count] / [count][3]; outputs may be NULL. */
int alore_backend_predicted_state(alore_backend_handle h, int count, double resolution, const double *start_time, const double *time,
                                  const double *start_xytheta, double *xytheta, double *vaj, double *oaj, int *forward);

/* MSPlanner::mincoPointPub (optimizer.cpp:1714-1826) on the plans of the last alore_backend_plan: the optimised path as the
 * marker points the reference publishes for the ALORE FSM to follow -- per piece `panels_per_piece` Simpson panels of the
 * planar velocity ((int)(sparseResolution * 0.4) in the reference), accumulated from the plan's start position, the last
 * point of every piece twice.  HOST pointers: xy [count][max_pieces (panels_per_piece + 1)][2] (z = 0.15 in the marker),
 * yaw [count][max_pieces panels_per_piece] or NULL (the heading at the end of every panel: the reference computes it and
 * does not publish it), n_points [count] (n_pieces (panels_per_piece + 1); 0 for a slot whose plan the optimiser rejected).
 * Synchronises. */
int alore_backend_path_points(alore_backend_handle h, int count, int panels_per_piece, double *xy, double *yaw, int *n_points);

#ifdef __cplusplus
}
#endif
#endif /* ALORE_BACKEND_H */
