// traj_anal.hpp -- host-side reference sampling for the NMPC (float64, no Eigen, no ROS).
//
// Restates, for the controller side of the hot path:
//   * the banded LU of gcopter        P/back_end/include/gcopter/minco.hpp:43-200   (BandedSystem)
//   * the minimum-jerk spline solve    P/back_end/include/gcopter/minco.hpp:772-913  (MINCO_S3NU::
//                                       setConditions / setParameters / getTrajectory)
//   * quintic piece evaluation         P/back_end/include/gcopter/trajectory.hpp:75-135, 472-502
//   * TrajAnal                         P/nmpc_controller/include/nmpc_controller/traj_anal.hpp:11-139
// with P = /root/reference/planning_ddr_opt.  The trajectory lives in "flat" space: component 0 is the
// yaw theta(t), component 1 the arc length s(t); the Cartesian pose is recovered by Simpson
// integration of  x' = s' cos(theta) + theta' xv sin(theta),  y' = s' sin(theta) - theta' xv cos(theta).
//
// None of this can be compiled from the reference here (Eigen/ROS absent), so parity for these rows is
// UNPINNED; tests/test_host_layer.py pins the restatement with analytic known answers instead.
#pragma once

#include <array>
#include <cmath>
#include <stdexcept>
#include <vector>

#include "../csrc/minco_core.h"

namespace alore {

// One quintic piece in 2 flat coordinates.  c[d][i] multiplies t^i (ascending powers); the reference
// stores the same numbers with descending powers (trajectory.hpp:75-88, minco.hpp:900-913).
struct Piece5 {
    double duration = 0.0;
    double c[2][6] = {{0}};
    void pos(double t, double out[2]) const
    {
        for (int d = 0; d < 2; ++d) {
            double tn = 1.0, v = 0.0;
            for (int i = 0; i <= 5; ++i) { v += tn * c[d][i]; tn *= t; }
            out[d] = v;
        }
    }
    void vel(double t, double out[2]) const
    {
        for (int d = 0; d < 2; ++d) {
            double tn = 1.0, v = 0.0;
            for (int i = 1; i <= 5; ++i) { v += i * tn * c[d][i]; tn *= t; }
            out[d] = v;
        }
    }
    void acc(double t, double out[2]) const
    {
        for (int d = 0; d < 2; ++d) {
            double tn = 1.0, v = 0.0;
            for (int i = 2; i <= 5; ++i) { v += (i - 1) * i * tn * c[d][i]; tn *= t; }
            out[d] = v;
        }
    }
};

class Trajectory5 {
public:
    std::vector<Piece5> pieces;
    double totalDuration() const
    {
        double s = 0.0;
        for (const auto& p : pieces) s += p.duration;
        return s;
    }
    // trajectory.hpp:472-490: walks the pieces, t becomes the local time; past the end the last piece
    // is extrapolated
    int locatePieceIdx(double& t) const
    {
        const int N = (int)pieces.size();
        int idx;
        double dur;
        for (idx = 0; idx < N && t > (dur = pieces[idx].duration); ++idx) t -= dur;
        if (idx == N) {
            --idx;
            t += pieces[idx].duration;
        }
        return idx;
    }
    void getPos(double t, double out[2]) const { const int i = locatePieceIdx(t); pieces[i].pos(t, out); }
    void getVel(double t, double out[2]) const { const int i = locatePieceIdx(t); pieces[i].vel(t, out); }
    void getAcc(double t, double out[2]) const { const int i = locatePieceIdx(t); pieces[i].acc(t, out); }
};

// Minimum-jerk (s = 3) spline through inner points with given boundary P/V/A and piece times:
// the 6M x 6M banded system of minco.hpp:817-898.
class MincoS3NU {
public:
    // head / tail: [dim][p, v, a]
    void setConditions(const double head[2][3], const double tail[2][3], int pieceNum)
    {
        N_ = pieceNum;
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < 3; ++k) { head_[d][k] = head[d][k]; tail_[d][k] = tail[d][k]; }
    }
    // inPs: (N-1) inner points, [i][dim]; ts: N piece durations.  The system and its banded LU are
    // csrc/minco_core.h (shared with the device build of the same trajectory).
    void setParameters(const std::vector<std::array<double, 2>>& inPs, const std::vector<double>& ts)
    {
        if ((int)ts.size() != N_ || (int)inPs.size() != N_ - 1) throw std::invalid_argument("MincoS3NU sizes");
        T1_ = ts;
        std::vector<double> inner((size_t)2 * (N_ > 1 ? N_ - 1 : 1), 0.0);
        for (int i = 0; i < N_ - 1; ++i) { inner[2 * i] = inPs[i][0]; inner[2 * i + 1] = inPs[i][1]; }
        band_.assign((size_t)minco::band_doubles(N_), 0.0);
        b_.assign((size_t)minco::rhs_doubles(N_), 0.0);
        minco::spline_solve(N_, T1_.data(), inner.data(), head_, tail_, band_.data(), b_.data());
    }
    void getTrajectory(Trajectory5& traj) const
    {
        traj.pieces.clear();
        traj.pieces.reserve(N_);
        for (int i = 0; i < N_; ++i) {
            Piece5 p;
            p.duration = T1_[i];
            for (int d = 0; d < 2; ++d)
                for (int k = 0; k < 6; ++k) p.c[d][k] = b_[(size_t)(6 * i + k) * 2 + d];
            traj.pieces.push_back(p);
        }
    }

private:
    int N_ = 0;
    double head_[2][3] = {{0}}, tail_[2][3] = {{0}};
    std::vector<double> band_, b_;
    std::vector<double> T1_;
};

// the message the planner sends to the controller (P/utils/carstatemsgs/msg/Polynome.msg), ROS-free
struct Polynome {
    double traj_start_time = 0.0;
    std::vector<std::array<double, 2>> innerpoints; // (theta, s)
    std::vector<double> t_pts;
    double init_p[2] = {0, 0}, init_v[2] = {0, 0}, init_a[2] = {0, 0};
    double tail_p[2] = {0, 0}, tail_v[2] = {0, 0}, tail_a[2] = {0, 0};
    double start_position[3] = {0, 0, 0}; // x, y, theta
    double ICR[3] = {0, 0, 0};            // as sent: (yr, yl, xv) -> TrajAnal keeps it as a vector, .z = xv
};

class TrajAnal {
public:
    bool if_get_traj_ = false;

    void setRes(double state_seq_res, double Integral_appr_resInt)
    {
        state_seq_res_ = state_seq_res;
        Integral_appr_resInt_ = (int)Integral_appr_resInt;
    }
    // traj_anal.hpp:36-53
    void setTraj(const double start_state[3], const double initstate[2][3], const double finalstate[2][3],
                 const std::vector<std::array<double, 2>>& innerpoints, const std::vector<double>& pieceTimes,
                 const double ICR[3])
    {
        const int traj_num = (int)pieceTimes.size();
        if ((int)innerpoints.size() != traj_num - 1) throw std::invalid_argument("Innerpoints.cols() != pieceTimes.size()-1");
        minco_.setConditions(initstate, finalstate, traj_num);
        minco_.setParameters(innerpoints, pieceTimes);
        minco_.getTrajectory(traj_);
        for (int i = 0; i < 3; ++i) { start_state_[i] = start_state[i]; ICR_[i] = ICR[i]; }
        getSeq();
    }
    void setTraj(const Polynome& m)
    {
        const double init[2][3] = {{m.init_p[0], m.init_v[0], m.init_a[0]}, {m.init_p[1], m.init_v[1], m.init_a[1]}};
        const double fin[2][3] = {{m.tail_p[0], m.tail_v[0], m.tail_a[0]}, {m.tail_p[1], m.tail_v[1], m.tail_a[1]}};
        setTraj(m.start_position, init, fin, m.innerpoints, m.t_pts, m.ICR);
    }
    // traj_anal.hpp:55-95: composite Simpson, one checkpoint every state_seq_res_
    void getSeq()
    {
        state_sequence_.clear();
        const double res = state_seq_res_ / Integral_appr_resInt_;
        const double half = res / 2.0, sixth = res / 6.0;
        double cur[3] = {start_state_[0], start_state_[1], start_state_[2]};
        state_sequence_.push_back({cur[0], cur[1], cur[2], 0.0});
        const int sequence_num = (int)std::floor(traj_.totalDuration() / res);
        double p1[2], p2[2], p3[2], v1[2], v2[2], v3[2];
        traj_.getPos(0.0, p3);
        traj_.getVel(0.0, v3);
        for (int i = 0; i < sequence_num; ++i) {
            p1[0] = p3[0]; p1[1] = p3[1]; v1[0] = v3[0]; v1[1] = v3[1];
            traj_.getPos(i * res + half, p2);
            traj_.getVel(i * res + half, v2);
            traj_.getPos(i * res + res, p3);
            traj_.getVel(i * res + res, v3);
            cur[0] += sixth * (xdot(p1, v1) + 4.0 * xdot(p2, v2) + xdot(p3, v3));
            cur[1] += sixth * (ydot(p1, v1) + 4.0 * ydot(p2, v2) + ydot(p3, v3));
            if (i % Integral_appr_resInt_ == Integral_appr_resInt_ - 1)
                state_sequence_.push_back({cur[0], cur[1], p3[0], (i + 1) * res});
        }
    }
    const std::vector<std::array<double, 4>>& get_state_sequence_() const { return state_sequence_; }
    double get_traj_duration() const { return traj_.totalDuration(); }
    // traj_anal.hpp:105-130: nearest checkpoint + one Simpson panel
    void getPstate(double t, double out[3]) const
    {
        const int index = (int)std::floor(t / state_seq_res_);
        const double floor_t = index * state_seq_res_;
        const double diff_t = t - floor_t;
        const auto& st = state_sequence_.at((size_t)index);
        double p1[2], p2[2], p3[2], v1[2], v2[2], v3[2];
        traj_.getPos(floor_t, p1); traj_.getVel(floor_t, v1);
        traj_.getPos(floor_t + diff_t / 2.0, p2); traj_.getVel(floor_t + diff_t / 2.0, v2);
        traj_.getPos(t, p3); traj_.getVel(t, v3);
        out[0] = st[0] + diff_t / 6.0 * (xdot(p1, v1) + 4.0 * xdot(p2, v2) + xdot(p3, v3));
        out[1] = st[1] + diff_t / 6.0 * (ydot(p1, v1) + 4.0 * ydot(p2, v2) + ydot(p3, v3));
        out[2] = p3[0];
    }
    void getVstate(double t, double out[2]) const { traj_.getVel(t, out); } // (theta', s')
    void getAstate(double t, double out[2]) const { traj_.getAcc(t, out); }
    const Trajectory5& trajectory() const { return traj_; }
    double state_seq_res() const { return state_seq_res_; }
    double icr_xv() const { return ICR_[2]; }

private:
    double xdot(const double p[2], const double v[2]) const { return v[1] * std::cos(p[0]) + v[0] * ICR_[2] * std::sin(p[0]); }
    double ydot(const double p[2], const double v[2]) const { return v[1] * std::sin(p[0]) - v[0] * ICR_[2] * std::cos(p[0]); }
    MincoS3NU minco_;
    Trajectory5 traj_;
    double start_state_[3] = {0, 0, 0};
    double ICR_[3] = {0, 0, 0};
    std::vector<std::array<double, 4>> state_sequence_;
    double state_seq_res_ = 0.1;
    int Integral_appr_resInt_ = 4;
};

} // namespace alore
