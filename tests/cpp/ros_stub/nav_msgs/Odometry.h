// Test-only stand-in (see ros/ros.h in this directory).
#pragma once
#include <geometry_msgs/Point.h>
#include <ros/ros.h>
namespace nav_msgs {
struct Odometry {
    typedef std::shared_ptr<const Odometry> ConstPtr;
    struct PoseWithCov { geometry_msgs::Pose pose; } pose;
    struct TwistWithCov { geometry_msgs::Twist twist; } twist;
};
}  // namespace nav_msgs
namespace ros { namespace stub {
template <> inline std::shared_ptr<const nav_msgs::Odometry> synthetic<nav_msgs::Odometry>() {
    auto m = std::make_shared<nav_msgs::Odometry>();
    m->pose.pose.position.x = 1.0; m->pose.pose.position.y = -0.5; m->pose.pose.position.z = 1.0;   // the start
    m->twist.twist.linear.x = 0.3; m->twist.twist.linear.y = -0.2; m->twist.twist.linear.z = 0.1;
    return m;
}
} }
