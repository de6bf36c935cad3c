// Stand-in for the generated quadrotor_msgs/PositionCommand.h -- TEST INFRASTRUCTURE ONLY: the fields poly_traj_server.cpp:28-52 sets.
#pragma once
#include <string>
#include <ros/ros.h>
namespace quadrotor_msgs {
struct PositionCommand {
    struct Header { ros::Time stamp; std::string frame_id; } header;
    struct V3 { double x = 0.0, y = 0.0, z = 0.0; } position, velocity, acceleration;
    double yaw = 0.0, yaw_dot = 0.0;
};
}  // namespace quadrotor_msgs
