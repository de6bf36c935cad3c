// Stand-in -- TEST INFRASTRUCTURE ONLY: poly_traj_server.cpp:29-30,84-87 reads header.stamp and copies the message.
#pragma once
#include <memory>
#include <string>
#include <ros/ros.h>
namespace nav_msgs {
struct Odometry {
    struct Header { ros::Time stamp; std::string frame_id; } header;
    typedef std::shared_ptr<const Odometry> ConstPtr;
};
}  // namespace nav_msgs
