// oracle/traj_oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY: C entry points over oracle/traj_oracle.hpp so that
// tests can drive the host restatement of the reference sampling through ctypes (oracle/liboracle_traj.so).
#include <new>

#include "traj_oracle.hpp"

using namespace alore_oracle;

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

void* orc_sampler_create(int N, double dt, double state_seq_res, double integral_res_int)
{
    try { return new RefSampler(N, dt, state_seq_res, integral_res_int); } catch (...) { return nullptr; }
}
void orc_sampler_destroy(void* s) { delete static_cast<RefSampler*>(s); }
int orc_sampler_traj(void* s, double traj_start_time, int n_pieces, const double* innerpoints, const double* t_pts,
                     const double* init_pva, const double* tail_pva, const double* start_position, const double* ICR)
{
    try {
        static_cast<RefSampler*>(s)->TrajCallback(
            make_polynome(traj_start_time, n_pieces, innerpoints, t_pts, init_pva, tail_pva, start_position, ICR));
        return 0;
    } catch (...) { return -1; }
}
void orc_sampler_odom(void* s, double x, double y, double yaw) { static_cast<RefSampler*>(s)->OdomCallback(x, y, yaw); }
void orc_sampler_icr(void* s, double yr, double yl, double xv) { static_cast<RefSampler*>(s)->ICRCallback(yr, yl, xv); }
// one CmdCallback worth of reference handling: swap in the pending trajectory, sample, unwrap yaw
int orc_sampler_refs(void* s_, double now, int do_smooth, double* ref_states /* 3 x (N+1) */, double* ref_inputs /* 2 x (N+1) */,
                     int* at_goal)
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
int orc_sampler_at_goal(void* s) { return static_cast<RefSampler*>(s)->at_goal ? 1 : 0; }
double orc_sampler_duration(void* s) { return static_cast<RefSampler*>(s)->new_traj_.get_traj_duration(); }
// direct TrajAnal queries on the pending (latest) trajectory
int orc_sampler_state(void* s_, double t, double* p3, double* v2, double* a2)
{
    RefSampler* s = static_cast<RefSampler*>(s_);
    try {
        s->new_traj_.getPstate(t, p3);
        s->new_traj_.getVstate(t, v2);
        s->new_traj_.getAstate(t, a2);
        return 0;
    } catch (...) { return -1; }
}
int orc_sampler_flat(void* s_, double t, int order, double* out2)
{
    RefSampler* s = static_cast<RefSampler*>(s_);
    const Trajectory5& tr = s->new_traj_.trajectory();
    double tl = t;
    const int i = tr.locatePieceIdx(tl);
    for (int d = 0; d < 2; ++d) { // derivative `order` of the quintic at the local time
        double v = 0.0;
        for (int k = order; k <= 5; ++k) {
            double f = 1.0;
            for (int q = 0; q < order; ++q) f *= (k - q);
            double tn = 1.0;
            for (int q = 0; q < k - order; ++q) tn *= tl;
            v += f * tr.pieces[i].c[d][k] * tn;
        }
        out2[d] = v;
    }
    return i;
}
int orc_sampler_coefficients(void* s_, double* dur, double* coef /* [piece][2][6] */, int max_pieces)
{
    const auto& pieces = static_cast<RefSampler*>(s_)->new_traj_.trajectory().pieces;
    const int n = (int)pieces.size() < max_pieces ? (int)pieces.size() : max_pieces;
    for (int i = 0; i < n; ++i) {
        dur[i] = pieces[i].duration;
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < 6; ++k) coef[(i * 2 + d) * 6 + k] = pieces[i].c[d][k];
    }
    return (int)pieces.size();
}
int orc_sampler_sequence(void* s_, double* out4, int max_rows)
{
    const auto& seq = static_cast<RefSampler*>(s_)->new_traj_.get_state_sequence_();
    const int n = (int)seq.size() < max_rows ? (int)seq.size() : max_rows;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) out4[i * 4 + k] = seq[i][k];
    return (int)seq.size();
}
void orc_normlize_theta(double* th) { RefSampler::normlize_theta(*th); }
}
