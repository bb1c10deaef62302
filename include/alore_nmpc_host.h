/*
 * alore_nmpc_host.h -- C entry points of the host layer (libalore_nmpc_host.so): B instances of the reference's
 * `nmpc` node without ROS.  Each robot b is one MpcController
 * (planning_ddr_opt/nmpc_controller/include/nmpc_controller/mpc.h:83-200, src/mpc.cpp): its subscriber callbacks
 * become calls (odom <- ~odom, icr <- ~EKF_ICR, traj <- ~traj, emergency_stop <- /planner/emergency_stop,
 * mpc.cpp:22-53, 112-171, 279-294), its 100 Hz timer callback becomes alore_host_controller_tick(now)
 * (mpc.cpp:173-240), and what it publishes on ~wheel_cmd (carstatemsgs/CarControl) and ~cmd
 * (carstatemsgs/CarState) comes back in alore_host_command.  All numerics run on the GPU through
 * include/alore_nmpc.h; creation fails (NULL) without one.
 */
#ifndef ALORE_NMPC_HOST_H
#define ALORE_NMPC_HOST_H

#ifdef __cplusplus
extern "C" {
#endif

/* the node's private parameters (mpc.cpp:11-20, 32-33, 69-70); alore_host_default_params gives the nh.param
 * defaults with the weights of nmpc_controller/config/mpc3ms.yaml */
typedef struct alore_host_mpc_params {
    double max_omega, max_domega, max_vel, min_vel, max_acc, cmd_timer_rate, max_mpc_time;
    int if_mpc, delay_num;
    double state_seq_res;
    int Integral_appr_resInt;
    double matrix_q[3], matrix_r[2];
} alore_host_mpc_params;

typedef struct alore_host_command {
    double right_wheel_ome, left_wheel_ome; /* ~wheel_cmd */
    double v, omega, a, alpha;              /* ~cmd */
    int wheel_published, state_published;   /* which of the two topics the robot published on this tick */
} alore_host_command;

void alore_host_default_params(alore_host_mpc_params *p);
void *alore_host_controller_create(int B, int N, double dt, const alore_host_mpc_params *p, int device, int max_pieces,
                                   int max_checkpoints);
void alore_host_controller_destroy(void *c);
int alore_host_controller_odom(void *c, int b, double x, double y, double yaw);
int alore_host_controller_icr(void *c, int b, double yr, double yl, double xv); /* PointStamped x, y, z */
/* one carstatemsgs/Polynome: innerpoints (n_pieces-1) x 2 (theta, s), t_pts n_pieces, init_pva / tail_pva =
 * [p.x p.y v.x v.y a.x a.y], start_position (x, y, theta), ICR as sent */
int alore_host_controller_traj(void *c, int b, double traj_start_time, int n_pieces, const double *innerpoints,
                               const double *t_pts, const double *init_pva, const double *tail_pva,
                               const double *start_position, const double *ICR);
int alore_host_controller_emergency_stop(void *c, int b);
int alore_host_controller_robot_state(void *c, int b, int *at_goal, int *receive_traj, int *has_odom);
int alore_host_controller_tick(void *c, double now, alore_host_command *cmd /* [B] */);
/* the references the solver saw on the last tick (device copies): y B x N x 5, yN B x 3, od B x (N+1) x 3, x0 B x 3 */
int alore_host_controller_references(void *c, float *y, float *yN, float *od, float *x0);
int alore_host_controller_prediction(void *c, int b, double *states /* 3 x (N+1) */, double *inputs /* 2 x N */, int *status);

#ifdef __cplusplus
}
#endif
#endif
