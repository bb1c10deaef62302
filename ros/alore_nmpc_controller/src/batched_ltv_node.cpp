// batched_ltv_node.cpp -- the reference's `mpc` node (planning_ddr_opt/mpc_controller, the controller that
// planner_sim.launch:113-120 starts) for a fleet: ONE process, B robots, one GPU launch per control tick.
// NOT built by this repository (no ROS in the image); compiles in a catkin workspace that holds the reference's
// `carstatemsgs`, against include/alore_nmpc.h + include/alore_ltv_mpc.h and libalore_nmpc.so.
//
// Per robot b, under <robot_ns><b>/ (mpc_controller/src/mpc.cpp:32-60):
//   sub  traj (carstatemsgs/Polynome), odom (nav_msgs/Odometry)      pub  cmd (carstatemsgs/CarState: v, omega)
// and /planner/emergency_stop (std_msgs/Bool).  Parameters: the reference's /mpc/* set (dt, predict_steps, delay_num,
// matrix_q, matrix_r, matrix_rd, max_vel, min_vel, max_omega, max_acc, max_domega, cmd_timer_rate) plus ~robots,
// ~robot_ns, ~device, ~relinearisations (the reference's loop is bounded by a 9.7 ms wall clock, mpc.cpp:585).
// The tick is CmdCallback (mpc.cpp:131-216): pending trajectories go to the device store (TrajAnal::setTraj on the
// GPU), getRefPoints + smooth_yaw are sampled there, getCmd runs for all robots, cmd = output(:, delay_num);
// at the goal a zero command is published once and the trajectory is dropped.
#include <carstatemsgs/CarState.h>
#include <carstatemsgs/Polynome.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <std_msgs/Bool.h>
#include <tf/transform_datatypes.h>

#include <string>
#include <vector>

#include "alore_ltv_mpc.h"
#include "alore_nmpc.h"

namespace {

struct Pending { // one Polynome waiting for the next tick
    bool valid = false;
    std::vector<double> inner, t_pts;
    double init[6], tail[6], start[3], icr[3], start_time = 0.0;
};

struct Fleet {
    alore_nmpc_handle store = nullptr; // only its trajectory store is used (alore_nmpc_refs_*)
    alore_ltv_handle ltv = nullptr;
    alore_ltv_config cfg;
    int B = 1, n_relin = 5;
    double state_seq_res = 0.1;
    int integral_res_int = 4;
    std::vector<Pending> pending;
    std::vector<double> est;          // [B][3]
    std::vector<char> has_odom, receive_traj, stopped, fresh;
    std::vector<int> at_goal;
    std::vector<double> output;       // [B][2] the commands of the tick
    std::vector<ros::Subscriber> subs;
    std::vector<ros::Publisher> cmd_pub;

    void odom(int b, const nav_msgs::Odometry::ConstPtr& m)
    {
        est[3 * b] = m->pose.pose.position.x; est[3 * b + 1] = m->pose.pose.position.y;
        est[3 * b + 2] = tf::getYaw(m->pose.pose.orientation);
        has_odom[b] = 1;
    }
    void traj(int b, const carstatemsgs::Polynome::ConstPtr& m)
    {
        Pending& p = pending[b];
        const int n = (int)m->t_pts.size();
        if (n <= 0) return;
        p.t_pts.assign(m->t_pts.begin(), m->t_pts.end());
        p.inner.assign(2 * (n > 1 ? n - 1 : 1), 0.0);
        for (int i = 0; i + 1 < n && i < (int)m->innerpoints.size(); ++i) { p.inner[2 * i] = m->innerpoints[i].x; p.inner[2 * i + 1] = m->innerpoints[i].y; }
        const double init[6] = {m->init_p.x, m->init_p.y, m->init_v.x, m->init_v.y, m->init_a.x, m->init_a.y};
        const double tail[6] = {m->tail_p.x, m->tail_p.y, m->tail_v.x, m->tail_v.y, m->tail_a.x, m->tail_a.y};
        for (int i = 0; i < 6; ++i) { p.init[i] = init[i]; p.tail[i] = tail[i]; }
        p.start[0] = m->start_position.x; p.start[1] = m->start_position.y; p.start[2] = m->start_position.z;
        p.icr[0] = m->ICR.x; p.icr[1] = m->ICR.y; p.icr[2] = m->ICR.z;
        p.start_time = m->traj_start_time.toSec();
        p.valid = true;
    }
    void stop(const std_msgs::Bool::ConstPtr& m)
    {
        if (!m->data) return;
        for (int b = 0; b < B; ++b) { stopped[b] = 1; receive_traj[b] = 0; }
    }
    void publish(int b, const ros::Time& now, double v, double w)
    {
        carstatemsgs::CarState c;
        c.Header.frame_id = "world"; c.Header.stamp = now;
        c.v = v; c.omega = w; c.a = 0.0; c.alpha = 0.0; c.js = 0.0; c.jyaw = 0.0;
        cmd_pub[b].publish(c);
    }
    void tick(const ros::TimerEvent&)
    {
        const ros::Time now = ros::Time::now();
        // new trajectories whose start time has come (mpc.cpp:135-140) -> device store, all robots in one call
        std::vector<int> robots;
        std::vector<alore_polynome> msgs;
        for (int b = 0; b < B; ++b) {
            Pending& p = pending[b];
            if (!p.valid || now.toSec() <= p.start_time) continue;
            alore_polynome m;
            m.n_pieces = (int)p.t_pts.size(); m.innerpoints = p.inner.data(); m.t_pts = p.t_pts.data();
            for (int i = 0; i < 2; ++i) {
                m.init_p[i] = p.init[i]; m.init_v[i] = p.init[2 + i]; m.init_a[i] = p.init[4 + i];
                m.tail_p[i] = p.tail[i]; m.tail_v[i] = p.tail[2 + i]; m.tail_a[i] = p.tail[4 + i];
            }
            for (int i = 0; i < 3; ++i) { m.start_position[i] = p.start[i]; m.ICR[i] = p.icr[i]; }
            m.traj_start_time = p.start_time;
            robots.push_back(b); msgs.push_back(m);
        }
        if (!robots.empty()) {
            if (alore_nmpc_refs_set_polynomes(store, (int)robots.size(), robots.data(), msgs.data(), state_seq_res, integral_res_int, nullptr) != 0)
                ROS_ERROR_THROTTLE(1.0, "trajectory rejected: %s", alore_nmpc_last_error(store));
            for (int b : robots) { pending[b].valid = false; receive_traj[b] = 1; stopped[b] = 0; fresh[b] = 1; }
        }
        if (alore_ltv_refs_from_store(ltv, store, B, now.toSec(), est.data(), at_goal.data(), nullptr) != 0 ||
            alore_ltv_tick(ltv, B, est.data(), n_relin, 0, output.data(), nullptr, nullptr) != 0) {
            ROS_ERROR_THROTTLE(1.0, "alore_ltv tick failed: %s", alore_ltv_last_error(ltv));
            return;
        }
        for (int b = 0; b < B; ++b) {
            if (stopped[b]) { publish(b, now, 0.0, 0.0); continue; }         // emergency stop: zero command
            if (!has_odom[b] || !receive_traj[b]) continue;                   // mpc.cpp:132-133
            if (at_goal[b]) { publish(b, now, 0.0, 0.0); receive_traj[b] = 0; continue; }   // :142-156
            publish(b, now, output[(size_t)b * 2], output[(size_t)b * 2 + 1]);
        }
    }
};

} // namespace

int main(int argc, char** argv)
{
    ros::init(argc, argv, "alore_batched_ltv_mpc");
    ros::NodeHandle nh("~");
    Fleet f;
    alore_ltv_default_config(&f.cfg);
    nh.param("dt", f.cfg.dt, f.cfg.dt);
    nh.param("predict_steps", f.cfg.predict_steps, f.cfg.predict_steps);
    nh.param("delay_num", f.cfg.delay_num, f.cfg.delay_num);
    nh.param("max_vel", f.cfg.max_vel, f.cfg.max_vel);
    nh.param("min_vel", f.cfg.min_vel, f.cfg.min_vel);
    nh.param("max_omega", f.cfg.max_omega, f.cfg.max_omega);
    nh.param("max_acc", f.cfg.max_acc, f.cfg.max_acc);
    nh.param("max_domega", f.cfg.max_domega, f.cfg.max_domega);
    std::vector<double> Q, R, Rd;
    nh.param("matrix_q", Q, std::vector<double>());
    nh.param("matrix_r", R, std::vector<double>());
    nh.param("matrix_rd", Rd, std::vector<double>());
    for (size_t i = 0; i < 4 && i < Q.size(); ++i) f.cfg.matrix_q[i] = Q[i];
    for (size_t i = 0; i < 2 && i < R.size(); ++i) f.cfg.matrix_r[i] = R[i];
    for (size_t i = 0; i < 2 && i < Rd.size(); ++i) f.cfg.matrix_rd[i] = Rd[i];
    double rate = 100.0;
    int device = 0, max_pieces = 64, max_checkpoints = 1024;
    std::string ns = "robot_";
    nh.param("cmd_timer_rate", rate, 100.0);
    nh.param("robots", f.B, 1);
    nh.param("robot_ns", ns, ns);
    nh.param("device", device, 0);
    nh.param("relinearisations", f.n_relin, 5);
    nh.param("state_seq_res", f.state_seq_res, 0.1);
    nh.param("Integral_appr_resInt", f.integral_res_int, 4);
    nh.param("max_pieces", max_pieces, 64);
    nh.param("max_checkpoints", max_checkpoints, 1024);

    alore_nmpc_config sc{20, 0.01f, device, 0, 0, -1};
    if (alore_nmpc_create(&sc, &f.store) != 0 || alore_nmpc_refs_init(f.store, f.B, max_pieces, max_checkpoints) != 0 ||
        alore_ltv_create(&f.cfg, device, f.B, &f.ltv) != 0) {
        ROS_FATAL("GPU engine creation failed: no usable GPU (there is no CPU path)");
        return 1;
    }
    f.pending.resize(f.B); f.est.assign(3 * f.B, 0.0); f.has_odom.assign(f.B, 0); f.receive_traj.assign(f.B, 0);
    f.stopped.assign(f.B, 0); f.fresh.assign(f.B, 0); f.at_goal.assign(f.B, 0);
    f.output.assign((size_t)f.B * 2, 0.0);
    ros::NodeHandle root;
    // robot_ns = "" with one robot: the reference node's own private topic names (~traj, ~odom, ~cmd, ...), so that the
    // remaps of planner_sim.launch apply unchanged (launch/planner_sim_dropin.launch)
    const bool dropin = ns.empty() && f.B == 1;
    if (dropin) root = nh;
    for (int b = 0; b < f.B; ++b) {
        const std::string pre = dropin ? std::string() : ns + std::to_string(b) + "/";
        f.subs.push_back(root.subscribe<nav_msgs::Odometry>(pre + "odom", 1, boost::bind(&Fleet::odom, &f, b, _1)));
        f.subs.push_back(root.subscribe<carstatemsgs::Polynome>(pre + "traj", 1, boost::bind(&Fleet::traj, &f, b, _1)));
        f.cmd_pub.push_back(root.advertise<carstatemsgs::CarState>(pre + "cmd", 1));
    }
    f.subs.push_back(root.subscribe<std_msgs::Bool>("/planner/emergency_stop", 1, &Fleet::stop, &f));
    ros::Timer timer = nh.createTimer(ros::Duration(1.0 / rate), &Fleet::tick, &f);
    ros::spin();
    alore_ltv_destroy(f.ltv);
    alore_nmpc_destroy(f.store);
    return 0;
}
