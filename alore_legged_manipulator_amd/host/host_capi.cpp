// host_capi.cpp -- C entry points over the C++ host layer (traj_anal.hpp, mpc_controller.hpp) so that
// tests and Python callers can drive it through ctypes.  Plain pointers and sizes only.
#include <new>

#include "mpc_controller.hpp"

using namespace alore;

namespace {
Polynome make_polynome(double traj_start_time, int n_pieces, const double* innerpoints /* (n-1) x 2 */,
                       const double* t_pts, const double* init_pva /* [p0 p1 v0 v1 a0 a1] */,
                       const double* tail_pva, const double* start_position, const double* ICR)
{
    Polynome m;
    m.traj_start_time = traj_start_time;
    for (int i = 0; i < n_pieces - 1; ++i) m.innerpoints.push_back({innerpoints[i * 2], innerpoints[i * 2 + 1]});
    m.t_pts.assign(t_pts, t_pts + n_pieces);
    for (int d = 0; d < 2; ++d) {
        m.init_p[d] = init_pva[d]; m.init_v[d] = init_pva[2 + d]; m.init_a[d] = init_pva[4 + d];
        m.tail_p[d] = tail_pva[d]; m.tail_v[d] = tail_pva[2 + d]; m.tail_a[d] = tail_pva[4 + d];
    }
    for (int i = 0; i < 3; ++i) { m.start_position[i] = start_position[i]; m.ICR[i] = ICR[i]; }
    return m;
}
} // namespace

extern "C" {

// ---- RefSampler (host only, no GPU) ------------------------------------------------------------
void* alore_host_sampler_create(int N, double dt, double state_seq_res, double integral_res_int)
{
    try { return new RefSampler(N, dt, state_seq_res, integral_res_int); } catch (...) { return nullptr; }
}
void alore_host_sampler_destroy(void* s) { delete static_cast<RefSampler*>(s); }
int alore_host_sampler_traj(void* s, double traj_start_time, int n_pieces, const double* innerpoints, const double* t_pts,
                            const double* init_pva, const double* tail_pva, const double* start_position, const double* ICR)
{
    try {
        static_cast<RefSampler*>(s)->TrajCallback(
            make_polynome(traj_start_time, n_pieces, innerpoints, t_pts, init_pva, tail_pva, start_position, ICR));
        return 0;
    } catch (...) { return -1; }
}
void alore_host_sampler_odom(void* s, double x, double y, double yaw) { static_cast<RefSampler*>(s)->OdomCallback(x, y, yaw); }
void alore_host_sampler_icr(void* s, double yr, double yl, double xv) { static_cast<RefSampler*>(s)->ICRCallback(yr, yl, xv); }
// one CmdCallback worth of reference handling: swap in the pending trajectory, sample, unwrap yaw
int alore_host_sampler_refs(void* s_, double now, int do_smooth, double* ref_states /* 3 x (N+1) */,
                            double* ref_inputs /* 2 x (N+1) */, int* at_goal)
{
    RefSampler* s = static_cast<RefSampler*>(s_);
    try {
        s->swapInNewTraj(now);
        s->getRefPoints(now);
        if (do_smooth) s->smooth_yaw();
    } catch (...) { return -1; }
    for (size_t i = 0; i < s->reference_states_.size(); ++i) ref_states[i] = s->reference_states_[i];
    for (size_t i = 0; i < s->reference_inputs_.size(); ++i) ref_inputs[i] = s->reference_inputs_[i];
    *at_goal = s->at_goal ? 1 : 0;
    return 0;
}
int alore_host_sampler_at_goal(void* s) { return static_cast<RefSampler*>(s)->at_goal ? 1 : 0; }
double alore_host_sampler_duration(void* s) { return static_cast<RefSampler*>(s)->new_traj_.get_traj_duration(); }
// direct TrajAnal queries on the pending (latest) trajectory
int alore_host_sampler_state(void* s_, double t, double* p3, double* v2, double* a2)
{
    RefSampler* s = static_cast<RefSampler*>(s_);
    try {
        s->new_traj_.getPstate(t, p3);
        s->new_traj_.getVstate(t, v2);
        s->new_traj_.getAstate(t, a2);
        return 0;
    } catch (...) { return -1; }
}
int alore_host_sampler_flat(void* s_, double t, double* pos2, double* vel2, double* acc2)
{
    RefSampler* s = static_cast<RefSampler*>(s_);
    s->new_traj_.trajectory().getPos(t, pos2);
    s->new_traj_.trajectory().getVel(t, vel2);
    s->new_traj_.trajectory().getAcc(t, acc2);
    return 0;
}
int alore_host_sampler_sequence(void* s_, double* out4, int max_rows)
{
    const auto& seq = static_cast<RefSampler*>(s_)->new_traj_.get_state_sequence_();
    const int n = (int)seq.size() < max_rows ? (int)seq.size() : max_rows;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) out4[i * 4 + k] = seq[i][k];
    return (int)seq.size();
}
void alore_host_normlize_theta(double* th) { RefSampler::normlize_theta(*th); }

// ---- BatchedMpcController (needs the GPU library) ----------------------------------------------
void* alore_host_controller_create(int B, int N, double dt, const double* matrix_q, const double* matrix_r, int delay_num,
                                   double state_seq_res, double integral_res_int, int device)
{
    try {
        return new BatchedMpcController(B, N, dt, matrix_q, matrix_r, delay_num, state_seq_res, integral_res_int, device);
    } catch (...) { return nullptr; }
}
void alore_host_controller_destroy(void* c) { delete static_cast<BatchedMpcController*>(c); }
int alore_host_controller_device_refs(void* c, int max_pieces, int max_checkpoints, int build_on_device)
{
    try { static_cast<BatchedMpcController*>(c)->useDeviceReferences(max_pieces, max_checkpoints, build_on_device != 0); return 0; } catch (...) { return -1; }
}
void* alore_host_controller_robot(void* c, int b) { return &static_cast<BatchedMpcController*>(c)->robots.at(b); }
int alore_host_controller_tick(void* c, double now, double* cmd)
{
    try { static_cast<BatchedMpcController*>(c)->tick(now, cmd); return 0; } catch (...) { return -1; }
}
// the references the solver saw on the last tick (device copies): y B x N x 5, yN B x 3, od B x (N+1) x 3, x0 B x 3
int alore_host_controller_references(void* c_, float* y, float* yN, float* od, float* x0)
{
    try { static_cast<BatchedMpcController*>(c_)->mpc_wrapper_.downloadReferences(y, yN, od, x0); return 0; } catch (...) { return -1; }
}
void alore_host_controller_prediction(void* c_, int b, double* states /* 3 x (N+1) */, double* inputs /* 2 x N */, int* status)
{
    BatchedMpcController* c = static_cast<BatchedMpcController*>(c_);
    c->mpc_wrapper_.getStates(b, states);
    c->mpc_wrapper_.getInputs(b, inputs);
    *status = c->mpc_wrapper_.getStatus(b);
}
}
