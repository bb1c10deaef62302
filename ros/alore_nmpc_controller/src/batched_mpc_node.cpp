// batched_mpc_node.cpp -- the reference's `nmpc_controller` node for a fleet: ONE process, B robots, one GPU launch
// per control tick.  NOT built by this repository (the image has no ROS); it compiles in a catkin workspace that
// also holds the reference's `carstatemsgs` package, against include/alore_nmpc_host.h + libalore_nmpc_host.so.
//
// Per robot b the topics are the reference's, under the namespace <robot_ns><b>/ (mpc.cpp:22-53, 94):
//   sub  odom (nav_msgs/Odometry), EKF_ICR (geometry_msgs/PointStamped), traj (carstatemsgs/Polynome)
//   pub  wheel_cmd (carstatemsgs/CarControl), cmd (carstatemsgs/CarState)
// and /planner/emergency_stop (std_msgs/Bool) stops every robot.  Private parameters are the reference's
// (max_omega, max_domega, max_vel, min_vel, max_acc, cmd_timer_rate, max_mpc_time, if_mpc, delay_num, state_seq_res,
// Integral_appr_resInt, matrix_q, matrix_r) plus ~robots, ~robot_ns, ~device, ~horizon, ~max_pieces, ~max_checkpoints.
#include <carstatemsgs/CarControl.h>
#include <carstatemsgs/CarState.h>
#include <carstatemsgs/Polynome.h>
#include <geometry_msgs/PointStamped.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <std_msgs/Bool.h>
#include <tf/transform_datatypes.h>

#include <string>
#include <vector>

#include "alore_nmpc_host.h"

namespace {

struct Fleet {
    void* ctl = nullptr;
    int B = 1;
    std::vector<ros::Subscriber> subs;
    std::vector<ros::Publisher> wheel_pub, cmd_pub;
    std::vector<alore_host_command> cmds;

    void odom(int b, const nav_msgs::Odometry::ConstPtr& m)
    {
        alore_host_controller_odom(ctl, b, m->pose.pose.position.x, m->pose.pose.position.y, tf::getYaw(m->pose.pose.orientation));
    }
    void icr(int b, const geometry_msgs::PointStamped::ConstPtr& m) { alore_host_controller_icr(ctl, b, m->point.x, m->point.y, m->point.z); }
    void traj(int b, const carstatemsgs::Polynome::ConstPtr& m)
    {
        const int n = (int)m->t_pts.size();
        if (n <= 0) return;
        std::vector<double> inner(2 * (n > 1 ? n - 1 : 1), 0.0);
        for (int i = 0; i + 1 < n && i < (int)m->innerpoints.size(); ++i) { inner[2 * i] = m->innerpoints[i].x; inner[2 * i + 1] = m->innerpoints[i].y; }
        const double init[6] = {m->init_p.x, m->init_p.y, m->init_v.x, m->init_v.y, m->init_a.x, m->init_a.y};
        const double tail[6] = {m->tail_p.x, m->tail_p.y, m->tail_v.x, m->tail_v.y, m->tail_a.x, m->tail_a.y};
        const double start[3] = {m->start_position.x, m->start_position.y, m->start_position.z};
        const double icr3[3] = {m->ICR.x, m->ICR.y, m->ICR.z};
        alore_host_controller_traj(ctl, b, m->traj_start_time.toSec(), n, inner.data(), m->t_pts.data(), init, tail, start, icr3);
    }
    void stop(const std_msgs::Bool::ConstPtr& m)
    {
        if (!m->data) return;
        for (int b = 0; b < B; ++b) alore_host_controller_emergency_stop(ctl, b);
    }
    void tick(const ros::TimerEvent&)
    {
        const ros::Time now = ros::Time::now();
        if (alore_host_controller_tick(ctl, now.toSec(), cmds.data()) != 0) {
            ROS_ERROR_THROTTLE(1.0, "alore_host_controller_tick failed");
            return;
        }
        for (int b = 0; b < B; ++b) {
            const alore_host_command& c = cmds[b];
            if (c.wheel_published) {
                carstatemsgs::CarControl w;
                w.Header.frame_id = "world"; w.Header.stamp = now;
                w.right_wheel_ome = c.right_wheel_ome; w.left_wheel_ome = c.left_wheel_ome;
                wheel_pub[b].publish(w);
            }
            if (c.state_published) {
                carstatemsgs::CarState s;
                s.Header.frame_id = "world"; s.Header.stamp = now;
                s.v = c.v; s.omega = c.omega; s.a = c.a; s.alpha = c.alpha; s.js = 0.0; s.jyaw = 0.0;
                cmd_pub[b].publish(s);
            }
        }
    }
};

} // namespace

int main(int argc, char** argv)
{
    ros::init(argc, argv, "alore_batched_mpc");
    ros::NodeHandle nh("~");
    Fleet f;
    alore_host_mpc_params p;
    alore_host_default_params(&p);
    bool if_mpc = true;
    nh.param("max_omega", p.max_omega, p.max_omega);
    nh.param("max_domega", p.max_domega, p.max_domega);
    nh.param("max_vel", p.max_vel, p.max_vel);
    nh.param("min_vel", p.min_vel, p.min_vel);
    nh.param("max_acc", p.max_acc, p.max_acc);
    nh.param("cmd_timer_rate", p.cmd_timer_rate, 100.0);
    nh.param("max_mpc_time", p.max_mpc_time, 10.0);
    nh.param("if_mpc", if_mpc, true);
    p.if_mpc = if_mpc ? 1 : 0;
    nh.param("delay_num", p.delay_num, p.delay_num);
    nh.param("state_seq_res", p.state_seq_res, p.state_seq_res);
    nh.param("Integral_appr_resInt", p.Integral_appr_resInt, p.Integral_appr_resInt);
    std::vector<double> Q, R;
    nh.param("matrix_q", Q, std::vector<double>());
    nh.param("matrix_r", R, std::vector<double>());
    for (size_t i = 0; i < 3 && i < Q.size(); ++i) p.matrix_q[i] = Q[i];
    for (size_t i = 0; i < 2 && i < R.size(); ++i) p.matrix_r[i] = R[i];
    int device = 0, horizon = 20, max_pieces = 64, max_checkpoints = 1024;
    std::string ns = "robot_";
    nh.param("robots", f.B, 1);
    nh.param("robot_ns", ns, ns);
    nh.param("device", device, 0);
    nh.param("horizon", horizon, 20);
    nh.param("max_pieces", max_pieces, 64);
    nh.param("max_checkpoints", max_checkpoints, 1024);

    f.ctl = alore_host_controller_create(f.B, horizon, 0.01, &p, device, max_pieces, max_checkpoints);
    if (!f.ctl) {
        ROS_FATAL("alore_host_controller_create failed: no usable GPU (there is no CPU path)");
        return 1;
    }
    f.cmds.resize(f.B);
    ros::NodeHandle root;
    // robot_ns = "" with one robot: the reference node's own private topic names (~traj, ~odom, ~cmd, ...), so that the
    // remaps of planner_sim.launch apply unchanged (launch/planner_sim_dropin.launch)
    const bool dropin = ns.empty() && f.B == 1;
    if (dropin) root = nh;
    for (int b = 0; b < f.B; ++b) {
        const std::string pre = dropin ? std::string() : ns + std::to_string(b) + "/";
        f.subs.push_back(root.subscribe<nav_msgs::Odometry>(pre + "odom", 1, boost::bind(&Fleet::odom, &f, b, _1)));
        f.subs.push_back(root.subscribe<geometry_msgs::PointStamped>(pre + "EKF_ICR", 1, boost::bind(&Fleet::icr, &f, b, _1)));
        f.subs.push_back(root.subscribe<carstatemsgs::Polynome>(pre + "traj", 1, boost::bind(&Fleet::traj, &f, b, _1)));
        f.wheel_pub.push_back(root.advertise<carstatemsgs::CarControl>(pre + "wheel_cmd", 1));
        f.cmd_pub.push_back(root.advertise<carstatemsgs::CarState>(pre + "cmd", 1));
    }
    f.subs.push_back(root.subscribe<std_msgs::Bool>("/planner/emergency_stop", 1, &Fleet::stop, &f));
    ros::Timer timer = nh.createTimer(ros::Duration(1.0 / p.cmd_timer_rate), &Fleet::tick, &f);
    ros::spin(); // single-threaded spinner: callbacks and the tick never overlap, as in the reference node
    alore_host_controller_destroy(f.ctl);
    return 0;
}
