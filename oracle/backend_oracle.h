/* oracle/backend_oracle.h -- TEST INFRASTRUCTURE ONLY (see backend_oracle.c).
 *
 * float64 CPU restatement of the reference's back_end trajectory optimiser
 * (planning_ddr_opt/back_end, class MSPlanner + gcopter headers).  PARITY UNPINNED against the
 * reference binary: back_end needs Eigen + ROS + PCL, none of which exists in this image, so the
 * reference cannot be compiled here and it ships no tests or golden vectors.  The restatement is pinned
 * by the self-checks SURVEY.md section 8(c) prescribes (tests/test_backend_oracle.py): dense NumPy solve
 * of the spline system, C4 continuity, finite-difference gradients, Simpson against adaptive quadrature,
 * bilinear ESDF against an analytic field.
 */
#ifndef ALORE_BACKEND_ORACLE_H
#define ALORE_BACKEND_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* plan_env::SDFmap as the optimiser sees it: a double ESDF grid, cell (ix, iy) at dist[ix * ny + iy],
 * cell centre ((ix + 0.5) res + x_lo, (iy + 0.5) res + y_lo)   (sdf_map.cpp:453-458, 739-757) */
typedef struct be_map {
    const double *dist;
    int nx, ny;
    double x_lo, y_lo, x_hi, y_hi, res;
} be_map;

/* lbfgs::lbfgs_parameter_t (back_end/include/gcopter/lbfgs.hpp:14-129) */
typedef struct be_lbfgs_param {
    int mem_size, past, max_iterations, max_linesearch;
    double g_epsilon, delta, min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
} be_lbfgs_param;

/* everything MSPlanner reads from the parameter server (optimizer.cpp:17-166) and from
 * plan_manager's Config (car3ms.yaml) */
typedef struct be_config {
    double max_vel, min_vel, max_acc, max_omega, max_domega, max_cen_acc; /* car3ms.yaml */
    int direct_v_omega;                                                    /* if_directly_constrain_v_omega */
    double w_time, w_acc, w_domega, w_collision, w_moment, w_mean_time, w_cen_acc; /* penaltyWeights */
    double p_time, p_bigpath, p_moment, p_mean_time, p_acc, p_domega;             /* PathpenaltyWeights */
    double energy_w[2];                                                           /* energyWeights (yaw, s) */
    double smooth_eps, safe_dis, final_min_safe_dis;
    int final_check_num, safe_replan_max;
    double mean_lo, mean_hi;
    int sparse_res;              /* sparseResolution: Simpson panels per piece */
    int n_check;                 /* body check points */
    double check_pts[8][2];
    double icr_xv;               /* ICR_.z() */
    int standard_diff;           /* if_standard_diff */
    double lam0[2], rho0[2], rho_max[2], gamma[2], tol;             /* Equal*           */
    double cut_lam0[2], cut_rho0[2], cut_rho_max[2], cut_gamma[2], cut_tol; /* CutEqual* */
    be_lbfgs_param path_lbfgs;   /* path_lbfgs_params */
    int shot_path_past;
    double shot_path_horizon;
    be_lbfgs_param lbfgs;        /* lbfgs_params */
    int max_alm_rounds;          /* safety cap on the reference's unbounded while(ros::ok()) loop */
    int exact_chain;             /* 0 = the reference's gradient, including its two inexact terms: the chain rule
                                    for obstacle terms counts the node itself with its full Simpson weight
                                    (optimizer.cpp:944-947) and, with if_standard_diff = false, one sign in
                                    d(dy)/d(yaw coefficients) (optimizer.cpp:822, 990); 1 = the exact
                                    derivatives, used only to pin the rest of the algebra by finite differences */
} be_config;

/* front_end's FlatTrajData (front_end/include/front_end/traj_representation.h:46-58), M = pieces */
typedef struct be_problem {
    int M;
    const double *inner;      /* (M-1) x 2: (yaw, s) of UnOccupied_traj_pts */
    double init_T;            /* UnOccupied_initT */
    const double *positions;  /* M x 2: UnOccupied_positions (x, y) then final_state_XYTheta (x, y) */
    double head[2][3];        /* start_state: [yaw | s][p v a] */
    double tail[2][3];        /* final_state */
    double start_xy[2];       /* start_state_XYTheta.head(2) */
    double final_xy[2];       /* final_state_XYTheta.head(2) */
    int if_cut;
} be_problem;

typedef struct be_result {
    double *inner;   /* (M-1) x 2 final inner points */
    double *T;       /* M final piece durations */
    double *coef;    /* 6M x 2 coefficients (row 6 i + k = power k of piece i; col 0 yaw, col 1 s) */
    double tail_s;   /* optimised finState(1, 0) */
    double cost;     /* final L-BFGS cost */
    double xy_err[2];
    int lbfgs_ret, path_ret; /* lbfgs return codes of the last stage-2 call / of stage 1 */
    int evals;       /* cost evaluations, both stages */
    int alm_rounds;
    int attempts;    /* minco_plan passes (collision retries) */
    int collision;   /* final check still colliding */
    double min_dist; /* minimum ESDF value seen by the final check of the last attempt */
} be_result;

void be_default_config(be_config *c);
void be_spline(int M, const double *T, const double *inner, const double head[2][3], const double tail[2][3],
               double *coef);
double be_energy(int M, const double *T, const double *coef, const double w[2], double *gdC, double *gdT);
void be_spline_adjoint(int M, const double *T, const double *coef, const double *gdC, const double *gdT,
                       double *grad_pts, double *grad_T, double grad_tail[2]);
double be_esdf(const be_map *m, double x, double y, double grad[2], int mode, double mindis);
/* MSPlanner::get_the_predicted_state[_and_path] (optimizer.cpp:1108-1262); xyt: start pose in, predicted pose out */
int be_predicted_state(const double *T, const double *coef, int M, int standard_diff, double xv, double step, double start_time, double time,
                       double xyt[3], double vaj[3], double oaj[3]);
/* SDFmap::updateESDF2d (sdf_map.cpp:618-681): grid states 0 unknown / 1 free / 2 occupied, dist_all updated in the window */
int be_update_esdf2d(const unsigned char *grid, int GLX, int GLY, double res, double x_lo, double y_lo, double odom_x, double odom_y,
                     double range, double *dist_all);

/* one cost-callback evaluation.  stage 1 = costFunctionCallbackPath, 2 = costFunctionCallback.
 * x, g: 3M-1 (inner points, tail s, virtual times).  lam/rho: ALM state (stage 2).  xy_err out. */
double be_eval(const be_config *c, const be_map *map, const be_problem *p, int stage, const double *x, double *g,
               const double lam[2], const double rho[2], double safe_dis, double time_weight, double xy_err[2],
               double *xy_nodes /* optional: M*(2*sparse_res+1) x 2 node positions (even nodes valid) */);

int be_optimize(const be_config *c, const be_map *map, const be_problem *p, double safe_dis, double time_weight,
                be_result *r);
int be_minco_plan(const be_config *c, const be_map *map, const be_problem *p, be_result *r);
int be_final_collision(const be_config *c, const be_map *map, int M, const double *T, const double *coef,
                       const double start_xy[2], double *min_dist);

/* lbfgs_optimize on one of the two cost functions, capped at max_iter iterations (tests: step-for-step
 * comparison with the device).  Returns the lbfgs code; x is updated in place. */
int be_lbfgs_run(const be_config *c, const be_map *map, const be_problem *p, int stage, double *x, double *cost,
                 const double lam[2], const double rho[2], double safe_dis, double time_weight, int max_iter,
                 int *n_iter, int *n_eval, double xy_err[2]);

#ifdef __cplusplus
}
#endif
/* MSPlanner::mincoPointPub (optimizer.cpp:1714-1826): the marker points of an optimised trajectory; returns their number */
int be_path_points(const double *T, const double *coef, int M, int standard_diff, double xv, int res, const double start_xy[2],
                   double *xy, double *yaw);

#endif
