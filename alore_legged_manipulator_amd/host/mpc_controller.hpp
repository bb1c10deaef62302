// mpc_controller.hpp -- C++ host layer above the C ABI: batched mirrors of the reference's
// Tracked_nmpc::MpcWrapper (A10) and of the ROS-free part of MpcController (A11).
//
//   MpcWrapper        P/nmpc_controller/include/nmpc_controller/mpc_wrapper.h:43-141,
//                     P/nmpc_controller/src/mpc_wrapper.cpp:33-410
//   MpcController     P/nmpc_controller/include/nmpc_controller/mpc.h:83-200,
//                     P/nmpc_controller/src/mpc.cpp:6-98 (parameters), :112-171 (callbacks),
//                     :173-240 (CmdCallback), :279-294 (emergencyStop), :296-350 (run), :502-509 (cmdPub)
// with P = /root/reference/planning_ddr_opt.  Same method names and argument meaning; every robot of
// the batch is one instance of the reference node.  ROS time is replaced by an explicit `now`
// argument, messages by plain structs (polynome.hpp).  All numerics run on the GPU through
// include/alore_nmpc.h: the solver AND what the reference does on the host before it -- turning a
// Polynome into a spline with Simpson checkpoints (TrajAnal::setTraj / getSeq), sampling it
// (getRefPoints), unwrapping the heading (smooth_yaw).  This layer only keeps the per-robot message
// state and the tick logic.  (A float64 host restatement of the sampling exists as a checker:
// oracle/traj_oracle.hpp -- test infrastructure, not included here.)
#pragma once

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/alore_nmpc.h"
#include "polynome.hpp"

namespace alore {

enum STATE { kX = 0, kY = 1, kPsi = 2 };       // mpc.h:57-61
enum CONTROL { kVr = 0, kVl = 1 };             // mpc.h:64-67
enum ONLINEDATA { kXv = 0, kYr = 1, kYl = 2 }; // mpc.h:70-74

struct CarICR { // mpc.h:76-81
    double yr = -0.2, yl = 0.2, xv = 0.0;
};

// The node's private parameters (mpc.cpp:11-20, 32-33, 69-70; values of
// P/nmpc_controller/config/mpc3ms.yaml where the file sets them, the nh.param defaults otherwise)
struct MpcParams {
    double max_omega = -1.0, max_domega = -1.0, max_vel = -1.0, min_vel = -1.0, max_acc = -1.0;
    double cmd_timer_rate = 100.0, max_mpc_time = 10.0;
    bool if_mpc = true;
    int delay_num = 0;
    double state_seq_res = 1.0;
    int Integral_appr_resInt = 10;
    double matrix_q[3] = {10.0, 10.0, 0.5};
    double matrix_r[2] = {0.1, 0.1};
};

// what one tick publishes for one robot: ~wheel_cmd (CarControl) and ~cmd (CarState)
struct RobotCommand {
    double right_wheel_ome = 0.0, left_wheel_ome = 0.0; // carstatemsgs/CarControl
    double v = 0.0, omega = 0.0, a = 0.0, alpha = 0.0;  // carstatemsgs/CarState (v, omega, a, alpha; js = jyaw = 0)
    bool wheel_published = false, state_published = false;
};

// ---- A10: batched MpcWrapper -------------------------------------------------------------------
class BatchedMpcWrapper {
public:
    static constexpr int kStateSize = ALORE_NMPC_NX, kRefSize = ALORE_NMPC_NY, kEndRefSize = ALORE_NMPC_NYN,
                         kInputSize = ALORE_NMPC_NU, kCostSize = ALORE_NMPC_NY - ALORE_NMPC_NU,
                         kOdSize = ALORE_NMPC_NOD;
    const int B, kSamples;

    BatchedMpcWrapper(int batch, int N, double dt = 0.01, int device = 0) : B(batch), kSamples(N), dt_(dt)
    {
        alore_nmpc_config cfg{N, (float)dt, device, 0, 0, -1};
        const int rc = alore_nmpc_create(&cfg, &h_);
        if (rc != ALORE_NMPC_OK) throw std::runtime_error("alore_nmpc_create failed (" + std::to_string(rc) + "): no CPU path");
        check(alore_nmpc_batch_alloc(h_, B, &dev_));
        check(alore_nmpc_batch_default_bounds(h_, &dev_, B, nullptr));
        x_.assign((size_t)B * 3 * (N + 1), 0.f); u_.assign((size_t)B * 2 * N, 0.f);
        od_.assign((size_t)B * 3 * (N + 1), 0.f); y_.assign((size_t)B * 5 * N, 0.f); yN_.assign((size_t)B * 3, 0.f);
        W_.assign((size_t)B * 25 * N, 0.f); WN_.assign((size_t)B * 9, 0.f); x0_.assign((size_t)B * 3, 0.f);
        status_.assign(B, 0);
        // mpc_wrapper.cpp:33-93: everything zero, default ICR (0, -0.2, 0.2), forward simulation, prepared
        const double icr[3] = {0.0, -0.2, 0.2};
        for (int b = 0; b < B; ++b) setICRParameters(b, icr);
        upload_all();
        check(alore_nmpc_forward_simulate(h_, &dev_, B, nullptr));
        acado_is_prepared_ = true;
    }
    ~BatchedMpcWrapper()
    {
        if (h_) {
            if (d_mask_) (void)hipFree(d_mask_);
            if (goal_pinned_) (void)alore_nmpc_host_free(goal_pinned_);
            alore_nmpc_batch_free(h_, &dev_);
            alore_nmpc_destroy(h_);
        }
    }
    BatchedMpcWrapper(const BatchedMpcWrapper&) = delete;
    BatchedMpcWrapper& operator=(const BatchedMpcWrapper&) = delete;

    // mpc_wrapper.cpp:106-138 (Q 3x3, R 2x2 row-major; the same weights for every robot)
    bool setCosts(const double Q[9], const double R[4], double state_cost_scaling = 0.0, double input_cost_scaling = 0.0)
    {
        if (state_cost_scaling < 0.0 || input_cost_scaling < 0.0) return false;
        const int N = kSamples;
        float state_scale = 1.0f, input_scale = 1.0f;
        std::vector<float> W((size_t)25 * N, 0.f);
        for (int i = 0; i < N; ++i) {
            state_scale = std::exp(-float(i) / float(N) * float(state_cost_scaling));
            input_scale = std::exp(-float(i) / float(N) * float(input_cost_scaling));
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) W[(size_t)i * 25 + r * 5 + c] = (float)Q[r * 3 + c] * state_scale;
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 2; ++c) W[(size_t)i * 25 + (3 + r) * 5 + 3 + c] = (float)R[r * 2 + c] * input_scale;
        }
        for (int b = 0; b < B; ++b) {
            std::memcpy(&W_[(size_t)b * 25 * N], W.data(), W.size() * sizeof(float));
            for (int k = 0; k < 9; ++k) WN_[(size_t)b * 9 + k] = (float)Q[k] * state_scale;
        }
        dirty_costs_ = true;
        return true;
    }
    // mpc_wrapper.cpp:200-207: (xv, yr, yl) to all N+1 nodes of robot b
    bool setICRParameters(int b, const double p_B_I[3])
    {
        for (int k = 0; k <= kSamples; ++k)
            for (int i = 0; i < 3; ++i) od_[((size_t)b * (kSamples + 1) + k) * 3 + i] = (float)p_B_I[i];
        return true;
    }
    // mpc_wrapper.cpp:242-264: states 3 x (N+1), inputs 2 x (N+1), column-major like the Eigen arguments
    bool setTrajectory(int b, const double* states, const double* inputs)
    {
        const int N = kSamples;
        for (int k = 0; k < N; ++k) {
            float* y = &y_[((size_t)b * N + k) * 5];
            y[0] = (float)states[k * 3]; y[1] = (float)states[k * 3 + 1]; y[2] = (float)states[k * 3 + 2];
            y[3] = (float)inputs[k * 2]; y[4] = (float)inputs[k * 2 + 1];
        }
        for (int i = 0; i < 3; ++i) yN_[(size_t)b * 3 + i] = (float)states[N * 3 + i];
        return true;
    }
    // mpc_wrapper.cpp:267-275: reset the iterate of robot b (x <- state replicated, u <- 0)
    void resetIterate(int b, const double state[3])
    {
        refreshStates(); // the whole mirror is uploaded with the next update: it must be current first
        refreshInputs();
        for (int k = 0; k <= kSamples; ++k)
            for (int i = 0; i < 3; ++i) x_[((size_t)b * (kSamples + 1) + k) * 3 + i] = (float)state[i];
        std::fill(u_.begin() + (size_t)b * 2 * kSamples, u_.begin() + (size_t)(b + 1) * 2 * kSamples, 0.f);
        dirty_iterate_ = true;
    }
    // ---- references sampled on the device (alore_nmpc_refs_*): od, y, yN and x0 never leave the GPU
    void enableDeviceRefs(int max_pieces, int max_checkpoints)
    {
        check(alore_nmpc_refs_init(h_, B, max_pieces, max_checkpoints));
        device_refs_ = true;
    }
    bool deviceRefs() const { return device_refs_; }
    // the same store filled on the device from the planner messages themselves
    void setDevicePolynomes(const std::vector<int>& robots, const std::vector<const Polynome*>& msgs, double state_seq_res,
                            int integral_res_int)
    {
        std::vector<alore_polynome> pm(msgs.size());
        std::vector<std::vector<double>> inner(msgs.size());
        for (size_t i = 0; i < msgs.size(); ++i) {
            const Polynome& m = *msgs[i];
            alore_polynome& q = pm[i];
            q.n_pieces = (int)m.t_pts.size();
            for (const auto& ip : m.innerpoints) { inner[i].push_back(ip[0]); inner[i].push_back(ip[1]); }
            q.innerpoints = inner[i].data();
            q.t_pts = m.t_pts.data();
            for (int d = 0; d < 2; ++d) {
                q.init_p[d] = m.init_p[d]; q.init_v[d] = m.init_v[d]; q.init_a[d] = m.init_a[d];
                q.tail_p[d] = m.tail_p[d]; q.tail_v[d] = m.tail_v[d]; q.tail_a[d] = m.tail_a[d];
            }
            for (int k = 0; k < 3; ++k) { q.start_position[k] = m.start_position[k]; q.ICR[k] = m.ICR[k]; }
            q.traj_start_time = m.traj_start_time;
        }
        check(alore_nmpc_refs_set_polynomes(h_, (int)pm.size(), robots.data(), pm.data(), state_seq_res, integral_res_int, nullptr));
    }
    // flat velocity / acceleration (theta', s', theta'', s'') of every robot's trajectory at now (the if_mpc = false
    // branch of CmdCallback, mpc.cpp:211-234); robots without a trajectory get zeros
    void flatStateAt(double now, double* out /* B x 4 */)
    {
        check(alore_nmpc_refs_eval(h_, B, now, out, nullptr));
    }
    // getRefPoints + smooth_yaw + setTrajectory + setICRParameters + the x0 of update() for all robots
    // at_goal is filled by the time the next update() returns (one synchronisation per tick)
    void sampleDeviceRefs(double now, const double* est /* B x 3 */, const double* icr /* B x 3: xv yr yl */, int* at_goal)
    {
        check(alore_nmpc_refs_sample(h_, &dev_, B, now, est, icr, 1, nullptr, nullptr));
        pending_goal_ = at_goal;
    }

    // which robots the next update() solves (1) and which it leaves exactly as they are (0): alore_nmpc_set_problem_mask.
    // The reference's CmdCallback returns early for a robot without odometry / trajectory, at its goal or stopped
    // (mpc.cpp:176-203): its solver state stays what its last real solve left.  nullptr: all.
    void setSolveMask(const unsigned char* solving)
    {
        bool all = solving == nullptr; // every robot solves: no mask (no 4 KB copy per tick, and the launch reads nothing extra)
        if (!all) {
            all = true;
            for (int b = 0; b < B && all; ++b) all = solving[b] != 0;
        }
        if (all) { check(alore_nmpc_set_problem_mask(h_, nullptr)); return; }
        if (!d_mask_ && hipMalloc((void**)&d_mask_, (size_t)B) != hipSuccess) throw std::runtime_error("hipMalloc (problem mask)");
        if (hipMemcpyAsync(d_mask_, solving, (size_t)B, hipMemcpyHostToDevice, nullptr) != hipSuccess) throw std::runtime_error("hipMemcpyAsync (problem mask)");
        check(alore_nmpc_set_problem_mask(h_, d_mask_));
    }

    // mpc_wrapper.cpp:279-373 for all robots at once: states = B x 3
    // cmd_node >= 0: only the inputs of that node (what the tick publishes) and the status come back with the launch;
    // the full input member follows on demand (getInputs / getInput of another node), like the predicted states
    bool update(const double* states, bool do_preparation = true, int cmd_node = -1)
    {
        if (!acado_is_prepared_) return false;
        for (size_t i = 0; i < (size_t)B * 3; ++i) x0_[i] = (float)states[i];
        alore_nmpc_batch host{};
        if (!device_refs_) { host.od = od_.data(); host.y = y_.data(); host.yN = yN_.data(); host.x0 = x0_.data(); }
        if (dirty_costs_) { host.W = W_.data(); host.WN = WN_.data(); }
        if (dirty_iterate_) { host.x = x_.data(); host.u = u_.data(); }
        check(alore_nmpc_batch_upload(h_, &dev_, &host, B, nullptr));
        dirty_costs_ = dirty_iterate_ = false;
        alore_nmpc_batch run = dev_; // the wrapper never asks for KKT value or objective (nor does the reference's)
        run.kkt = nullptr; run.obj = nullptr;
        check(alore_nmpc_rti(h_, &run, B, 1, nullptr));
        x_stale_ = true;
        // the at-goal flags come back through pinned memory (a copy into pageable memory blocks the host until the stream has drained to it)
        // and are handed over behind the tick's one synchronisation below
        int* goal_out = pending_goal_;
        pending_goal_ = nullptr;
        if (goal_out) {
            if (!goal_pinned_ && alore_nmpc_host_alloc(sizeof(int) * (size_t)B, (void**)&goal_pinned_) != ALORE_NMPC_OK) throw std::runtime_error("alore_nmpc_host_alloc (at-goal flags)");
            check(alore_nmpc_refs_at_goal(h_, B, goal_pinned_, nullptr));
        }
        if (cmd_node >= 0 && cmd_node < kSamples) {
            cmd_.resize((size_t)B * 2);
            check(alore_nmpc_input_column(h_, &dev_, B, cmd_node, cmd_.data(), status_.data(), nullptr)); // waits for the stream
            cmd_node_ = cmd_node;
            u_stale_ = true;
        } else {
            alore_nmpc_batch out{}; // the tick needs the inputs and the status; the predicted states come on demand
            out.u = u_.data(); out.status = status_.data();
            check(alore_nmpc_batch_download(h_, &dev_, &out, B, nullptr));
            cmd_node_ = -1;
            u_stale_ = false;
        }
        if (hipStreamSynchronize(nullptr) != hipSuccess) throw std::runtime_error("hipStreamSynchronize");
        if (goal_out) std::memcpy(goal_out, goal_pinned_, sizeof(int) * (size_t)B);
        acado_is_prepared_ = false;
        if (do_preparation) acado_is_prepared_ = true; // the preparation is fused into the next launch
        return true;
    }
    bool solve(const double* states) // mpc_wrapper.cpp:267-275 for every robot
    {
        for (int b = 0; b < B; ++b) resetIterate(b, states + (size_t)b * 3);
        return update(states);
    }
    bool prepare() { acado_is_prepared_ = true; return true; } // mpc_wrapper.cpp:377-383
    // what the solver saw on the last launch (device copies of the reference members)
    void downloadReferences(float* y, float* yN, float* od, float* x0)
    {
        alore_nmpc_batch out{};
        out.y = y; out.yN = yN; out.od = od; out.x0 = x0;
        check(alore_nmpc_batch_download(h_, &dev_, &out, B, nullptr));
        if (hipStreamSynchronize(nullptr) != hipSuccess) throw std::runtime_error("hipStreamSynchronize");
    }
    // mpc_wrapper.cpp:386-410 (double out)
    void getStates(int b, double* out /* 3 x (N+1) col-major */) const
    {
        refreshStates();
        for (int i = 0; i < 3 * (kSamples + 1); ++i) out[i] = x_[(size_t)b * 3 * (kSamples + 1) + i];
    }
    void getInputs(int b, double* out /* 2 x N col-major */) const
    {
        refreshInputs();
        for (int i = 0; i < 2 * kSamples; ++i) out[i] = u_[(size_t)b * 2 * kSamples + i];
    }
    double getInput(int b, int node, int which) const
    {
        if (u_stale_ && node == cmd_node_) return cmd_[(size_t)b * 2 + which];
        refreshInputs();
        return u_[((size_t)b * kSamples + node) * 2 + which];
    }
    int getStatus(int b) const { return status_[b]; }
    double getTimestep() const { return dt_; }

private:
    void refreshInputs() const
    {
        if (!u_stale_) return;
        alore_nmpc_batch out{};
        out.u = const_cast<float*>(u_.data());
        check(alore_nmpc_batch_download(h_, &dev_, &out, B, nullptr));
        if (hipStreamSynchronize(nullptr) != hipSuccess) throw std::runtime_error("hipStreamSynchronize");
        u_stale_ = false;
    }
    void refreshStates() const
    {
        if (!x_stale_) return;
        alore_nmpc_batch out{};
        out.x = const_cast<float*>(x_.data());
        check(alore_nmpc_batch_download(h_, &dev_, &out, B, nullptr));
        if (hipStreamSynchronize(nullptr) != hipSuccess) throw std::runtime_error("hipStreamSynchronize");
        x_stale_ = false;
    }
    void check(int rc) const
    {
        if (rc != ALORE_NMPC_OK) throw std::runtime_error(std::string("alore_nmpc: ") + alore_nmpc_last_error(h_));
    }
    void upload_all()
    {
        alore_nmpc_batch host{};
        host.x = x_.data(); host.u = u_.data(); host.od = od_.data(); host.y = y_.data(); host.yN = yN_.data();
        host.W = W_.data(); host.WN = WN_.data(); host.x0 = x0_.data();
        check(alore_nmpc_batch_upload(h_, &dev_, &host, B, nullptr));
        if (hipStreamSynchronize(nullptr) != hipSuccess) throw std::runtime_error("hipStreamSynchronize");
    }
    alore_nmpc_handle h_ = nullptr;
    alore_nmpc_batch dev_{};
    std::vector<float> x_, u_, od_, y_, yN_, W_, WN_, x0_;
    std::vector<float> cmd_;           // inputs of node cmd_node_ of the last launch
    int cmd_node_ = -1;
    mutable bool u_stale_ = false;     // u_ is behind the device (only cmd_ was fetched)
    std::vector<int> status_;
    bool acado_is_prepared_ = false, dirty_costs_ = true, dirty_iterate_ = true, device_refs_ = false;
    mutable bool x_stale_ = false; // x_ lags the device copy (downloaded on demand)
    int* pending_goal_ = nullptr;
    int* goal_pinned_ = nullptr; // pinned landing area of the at-goal flags
    unsigned char* d_mask_ = nullptr; // device copy of the solve mask of the tick (setSolveMask)
    const double dt_;
};

// ---- A11: per-robot message state + the tick ---------------------------------------------------------
// What MpcController keeps between callbacks, for one robot.
struct RobotNode {
    bool has_odom = false, receive_traj_ = false, at_goal = false;
    bool solve_from_scratch_ = true;          // mpc.cpp:317-320: the first solve of THIS robot resets its iterate
    bool pending = false;                     // new_traj_.if_get_traj_
    double new_traj_start_time_ = 0.0, start_time = -1.0, traj_duration = 0.0, new_duration_ = 0.0;
    Polynome msg_, new_msg_;
    unsigned traj_version = 0, uploaded_version = 0; // the device store follows traj_version
    CarICR car_icr_;
    double est_state_[3] = {0, 0, 0};
    double last_input_[2] = {0, 0};           // predicted_inputs_(., delay_num_) of the last solve

    void OdomCallback(double x, double y, double yaw) { has_odom = true; est_state_[0] = x; est_state_[1] = y; est_state_[2] = yaw; } // mpc.cpp:112-122
    void ICRCallback(double yr, double yl, double xv) { car_icr_.yr = yr; car_icr_.yl = yl; car_icr_.xv = xv; }                         // :124-128
    void TrajCallback(const Polynome& msg) // :130-171
    {
        if (pending) promote();
        new_msg_ = msg;
        new_duration_ = 0.0;
        for (double t : msg.t_pts) new_duration_ += t;
        new_traj_start_time_ = msg.traj_start_time;
        pending = true;
        receive_traj_ = true;
        at_goal = false;
    }
    void emergencyStop() { receive_traj_ = false; start_time = -1.0; } // :279-294 (the zero CarState is published by the controller)
    void swapInNewTraj(double now) { if (pending && now > new_traj_start_time_) promote(); } // :177-182
    void promote()
    {
        msg_ = new_msg_;
        traj_duration = new_duration_;
        ++traj_version;
        start_time = new_traj_start_time_;
        pending = false;
    }
};

// The ROS-free control tick of the reference node for B robots (mpc.cpp:173-240 CmdCallback + :296-350
// run + :502-509 cmdPub).
class BatchedMpcController {
public:
    BatchedMpcWrapper mpc_wrapper_;
    std::vector<RobotNode> robots;
    MpcParams params_;

    BatchedMpcController(int B, int N, double dt, const MpcParams& prm, int device = 0, int max_pieces = 64, int max_checkpoints = 1024)
        : mpc_wrapper_(B, N, dt, device), robots((size_t)B), params_(prm)
    {
        // mpc.cpp:67-85: diagonal weights from ~matrix_q / ~matrix_r
        const double Q[9] = {prm.matrix_q[0], 0, 0, 0, prm.matrix_q[1], 0, 0, 0, prm.matrix_q[2]};
        const double R[4] = {prm.matrix_r[0], 0, 0, prm.matrix_r[1]};
        mpc_wrapper_.setCosts(Q, R);
        mpc_wrapper_.enableDeviceRefs(max_pieces, max_checkpoints);
    }
    // /planner/emergency_stop (mpc.cpp:22, 279-294): the robot forgets its trajectory; `out` (optional) gets the
    // zero CarState the reference publishes from the callback
    void emergencyStop(int b, RobotCommand* out = nullptr)
    {
        robots.at((size_t)b).emergencyStop();
        if (out) { *out = RobotCommand{}; out->state_published = true; }
    }
    std::vector<double> tick_est_, tick_icr_;
    std::vector<int> tick_goal_, tick_fresh_robots_;
    std::vector<const Polynome*> tick_fresh_msgs_;
    std::vector<char> tick_solving_;
    // One CmdCallback for every robot.  cmd: B entries.  A robot without odometry or trajectory publishes nothing.
    void tick(double now, RobotCommand* cmd)
    {
        const int B = mpc_wrapper_.B, N = mpc_wrapper_.kSamples;
        const int node = params_.delay_num < N ? params_.delay_num : N - 1;
        // the tick's work arrays are members: nothing is allocated per tick
        std::vector<double>&est = tick_est_, &icr = tick_icr_;
        std::vector<int>&goal = tick_goal_, &fresh_robots = tick_fresh_robots_;
        std::vector<const Polynome*>& fresh_msgs = tick_fresh_msgs_;
        std::vector<char>& solving = tick_solving_;
        est.assign((size_t)B * 3, 0.0); icr.assign((size_t)B * 3, 0.0);
        goal.assign(B, 0); solving.assign(B, 0);
        fresh_robots.clear(); fresh_msgs.clear();
        for (int b = 0; b < B; ++b) {
            RobotNode& r = robots[b];
            cmd[b] = RobotCommand{};
            icr[(size_t)b * 3] = r.car_icr_.xv; icr[(size_t)b * 3 + 1] = r.car_icr_.yr; icr[(size_t)b * 3 + 2] = r.car_icr_.yl;
            for (int i = 0; i < 3; ++i) est[(size_t)b * 3 + i] = r.est_state_[i];
            if (!r.has_odom || !r.receive_traj_) continue;
            r.swapInNewTraj(now);
            if (r.traj_version != r.uploaded_version) {
                fresh_robots.push_back(b);
                fresh_msgs.push_back(&r.msg_);
                r.uploaded_version = r.traj_version;
            }
            if (r.at_goal) { // mpc.cpp:184-203: zero CarState, the last predicted input once more, then idle
                cmd[b].state_published = cmd[b].wheel_published = true;
                cmd[b].right_wheel_ome = r.last_input_[kVr];
                cmd[b].left_wheel_ome = r.last_input_[kVl];
                r.receive_traj_ = false;
                r.start_time = -1.0;
                continue;
            }
            solving[b] = 1;
        }
        if (!fresh_robots.empty())
            mpc_wrapper_.setDevicePolynomes(fresh_robots, fresh_msgs, params_.state_seq_res, params_.Integral_appr_resInt);
        if (!params_.if_mpc) { // open-loop replay of the planned flat velocities (mpc.cpp:211-234)
            std::vector<double> flat((size_t)B * 4, 0.0);
            mpc_wrapper_.flatStateAt(now, flat.data());
            for (int b = 0; b < B; ++b) {
                if (!solving[b]) continue;
                RobotNode& r = robots[b];
                if (now - r.start_time > r.traj_duration) r.at_goal = true;
                cmd[b].state_published = true;
                cmd[b].omega = flat[(size_t)b * 4]; cmd[b].v = flat[(size_t)b * 4 + 1];
                cmd[b].alpha = flat[(size_t)b * 4 + 2]; cmd[b].a = flat[(size_t)b * 4 + 3];
            }
            return;
        }
        mpc_wrapper_.sampleDeviceRefs(now, est.data(), icr.data(), goal.data()); // goal: valid after update()
        // mpc.cpp:317-320, per robot: x <- est replicated, u <- 0 on its FIRST solve only.  Idle robots (no odometry / no
        // trajectory / at goal / stopped) sit the launch out (problem mask): their iterate and multipliers stay what their last
        // real solve left, and they warm-start from them when they move again -- as the reference's node does, whose
        // CmdCallback returns early for them.
        for (int b = 0; b < B; ++b) {
            if (solving[b] && robots[b].solve_from_scratch_) {
                mpc_wrapper_.resetIterate(b, &est[(size_t)b * 3]);
                robots[b].solve_from_scratch_ = false;
            }
        }
        mpc_wrapper_.setSolveMask(reinterpret_cast<const unsigned char*>(solving.data()));
        mpc_wrapper_.update(est.data(), false, node); // only column `node` of the inputs crosses the bus with the tick
        mpc_wrapper_.prepare(); // the reference's preparation thread (mpc.cpp:336, 394-403)
        for (int b = 0; b < B; ++b) {
            if (!solving[b]) continue;
            RobotNode& r = robots[b];
            r.at_goal = goal[b] != 0; // getRefPoints sets it; the NEXT tick acts on it
            r.last_input_[kVr] = mpc_wrapper_.getInput(b, node, kVr);
            r.last_input_[kVl] = mpc_wrapper_.getInput(b, node, kVl);
            cmd[b].wheel_published = true; // cmdPub
            cmd[b].right_wheel_ome = r.last_input_[kVr];
            cmd[b].left_wheel_ome = r.last_input_[kVl];
        }
    }
};

} // namespace alore
