// polynome.hpp -- the message the planner sends to the controllers (P/utils/carstatemsgs/msg/Polynome.msg,
// filled by PlanManager::MPCPathPub, P/plan_manager/include/plan_manager/plan_manager.hpp:784-831), ROS-free.
#pragma once

#include <array>
#include <vector>

namespace alore {

struct Polynome {
    double traj_start_time = 0.0;
    std::vector<std::array<double, 2>> innerpoints; // (theta, s) = geometry_msgs/Vector3 .x, .y
    std::vector<double> t_pts;                      // piece durations
    double init_p[2] = {0, 0}, init_v[2] = {0, 0}, init_a[2] = {0, 0};
    double tail_p[2] = {0, 0}, tail_v[2] = {0, 0}, tail_a[2] = {0, 0};
    double start_position[3] = {0, 0, 0}; // x, y, theta
    double ICR[3] = {0, 0, 0};            // as sent: (yr, yl, xv)
};

} // namespace alore
