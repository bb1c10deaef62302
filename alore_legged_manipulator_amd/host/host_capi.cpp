// host_capi.cpp -- C entry points (include/alore_nmpc_host.h) over the C++ host layer (mpc_controller.hpp) so that
// tests and Python callers can drive the batched controller through ctypes.  Plain pointers and sizes only.
#include <new>

#include "../../include/alore_nmpc_host.h"
#include "mpc_controller.hpp"

using namespace alore;

namespace {
BatchedMpcController* C(void* c) { return static_cast<BatchedMpcController*>(c); }
}

extern "C" {

void alore_host_default_params(alore_host_mpc_params* p)
{
    const MpcParams d;
    p->max_omega = d.max_omega; p->max_domega = d.max_domega; p->max_vel = d.max_vel; p->min_vel = d.min_vel; p->max_acc = d.max_acc;
    p->cmd_timer_rate = d.cmd_timer_rate; p->max_mpc_time = d.max_mpc_time; p->if_mpc = d.if_mpc ? 1 : 0; p->delay_num = d.delay_num;
    p->state_seq_res = d.state_seq_res; p->Integral_appr_resInt = d.Integral_appr_resInt;
    for (int i = 0; i < 3; ++i) p->matrix_q[i] = d.matrix_q[i];
    for (int i = 0; i < 2; ++i) p->matrix_r[i] = d.matrix_r[i];
}

void* alore_host_controller_create(int B, int N, double dt, const alore_host_mpc_params* p, int device, int max_pieces,
                                   int max_checkpoints)
{
    try {
        MpcParams m;
        m.max_omega = p->max_omega; m.max_domega = p->max_domega; m.max_vel = p->max_vel; m.min_vel = p->min_vel; m.max_acc = p->max_acc;
        m.cmd_timer_rate = p->cmd_timer_rate; m.max_mpc_time = p->max_mpc_time; m.if_mpc = p->if_mpc != 0; m.delay_num = p->delay_num;
        m.state_seq_res = p->state_seq_res; m.Integral_appr_resInt = p->Integral_appr_resInt;
        for (int i = 0; i < 3; ++i) m.matrix_q[i] = p->matrix_q[i];
        for (int i = 0; i < 2; ++i) m.matrix_r[i] = p->matrix_r[i];
        return new BatchedMpcController(B, N, dt, m, device, max_pieces, max_checkpoints);
    } catch (...) { return nullptr; }
}
void alore_host_controller_destroy(void* c) { delete C(c); }

int alore_host_controller_odom(void* c, int b, double x, double y, double yaw)
{
    try { C(c)->robots.at((size_t)b).OdomCallback(x, y, yaw); return 0; } catch (...) { return -1; }
}
int alore_host_controller_icr(void* c, int b, double yr, double yl, double xv)
{
    try { C(c)->robots.at((size_t)b).ICRCallback(yr, yl, xv); return 0; } catch (...) { return -1; }
}
int alore_host_controller_traj(void* c, int b, double traj_start_time, int n_pieces, const double* innerpoints, const double* t_pts,
                               const double* init_pva, const double* tail_pva, const double* start_position, const double* ICR)
{
    try {
        Polynome m;
        m.traj_start_time = traj_start_time;
        for (int i = 0; i < n_pieces - 1; ++i) m.innerpoints.push_back({innerpoints[i * 2], innerpoints[i * 2 + 1]});
        m.t_pts.assign(t_pts, t_pts + n_pieces);
        for (int d = 0; d < 2; ++d) {
            m.init_p[d] = init_pva[d]; m.init_v[d] = init_pva[2 + d]; m.init_a[d] = init_pva[4 + d];
            m.tail_p[d] = tail_pva[d]; m.tail_v[d] = tail_pva[2 + d]; m.tail_a[d] = tail_pva[4 + d];
        }
        for (int i = 0; i < 3; ++i) { m.start_position[i] = start_position[i]; m.ICR[i] = ICR[i]; }
        C(c)->robots.at((size_t)b).TrajCallback(m);
        return 0;
    } catch (...) { return -1; }
}
int alore_host_controller_emergency_stop(void* c, int b)
{
    try { C(c)->emergencyStop(b); return 0; } catch (...) { return -1; }
}
int alore_host_controller_robot_state(void* c, int b, int* at_goal, int* receive_traj, int* has_odom)
{
    try {
        const RobotNode& r = C(c)->robots.at((size_t)b);
        if (at_goal) *at_goal = r.at_goal ? 1 : 0;
        if (receive_traj) *receive_traj = r.receive_traj_ ? 1 : 0;
        if (has_odom) *has_odom = r.has_odom ? 1 : 0;
        return 0;
    } catch (...) { return -1; }
}
int alore_host_controller_tick(void* c, double now, alore_host_command* cmd)
{
    try {
        BatchedMpcController* k = C(c);
        std::vector<RobotCommand> out((size_t)k->mpc_wrapper_.B);
        k->tick(now, out.data());
        for (size_t b = 0; b < out.size(); ++b) {
            cmd[b].right_wheel_ome = out[b].right_wheel_ome; cmd[b].left_wheel_ome = out[b].left_wheel_ome;
            cmd[b].v = out[b].v; cmd[b].omega = out[b].omega; cmd[b].a = out[b].a; cmd[b].alpha = out[b].alpha;
            cmd[b].wheel_published = out[b].wheel_published ? 1 : 0;
            cmd[b].state_published = out[b].state_published ? 1 : 0;
        }
        return 0;
    } catch (...) { return -1; }
}
int alore_host_controller_references(void* c, float* y, float* yN, float* od, float* x0)
{
    try { C(c)->mpc_wrapper_.downloadReferences(y, yN, od, x0); return 0; } catch (...) { return -1; }
}
int alore_host_controller_prediction(void* c, int b, double* states, double* inputs, int* status)
{
    try {
        C(c)->mpc_wrapper_.getStates(b, states);
        C(c)->mpc_wrapper_.getInputs(b, inputs);
        *status = C(c)->mpc_wrapper_.getStatus(b);
        return 0;
    } catch (...) { return -1; }
}
}
