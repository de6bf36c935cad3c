// Stand-in for the generated quadrotor_msgs/PolynomialTrajectory.h -- TEST INFRASTRUCTURE ONLY.  Fields, types and constants of
// /root/reference/src/simulator/utils/quadrotor_msgs/msg/PolynomialTrajectory.msg:1-28 as genmsg would emit them.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <ros/ros.h>
namespace quadrotor_msgs {
struct PolynomialTrajectory {
    struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; } header;
    uint32_t trajectory_id = 0;
    enum { ACTION_ADD = 1u, ACTION_ABORT = 2u, ACTION_WARN_START = 3u, ACTION_WARN_FINAL = 4u, ACTION_WARN_IMPOSSIBLE = 5u };
    uint32_t action = 0;
    uint32_t num_order = 0;
    uint32_t num_segment = 0;
    double start_yaw = 0.0;
    double final_yaw = 0.0;
    std::vector<double> coef_x, coef_y, coef_z, time;
    double mag_coeff = 0.0;
    std::vector<uint32_t> order;
    std::string debug_info;
    typedef std::shared_ptr<const PolynomialTrajectory> ConstPtr;
};
typedef std::shared_ptr<const PolynomialTrajectory> PolynomialTrajectoryConstPtr;
}  // namespace quadrotor_msgs
