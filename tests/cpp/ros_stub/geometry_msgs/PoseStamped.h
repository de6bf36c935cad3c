// Test-only stand-in (see ros/ros.h in this directory).
#pragma once
#include <geometry_msgs/Point.h>
#include <ros/ros.h>
namespace geometry_msgs {
struct PoseStamped {
    typedef std::shared_ptr<const PoseStamped> ConstPtr;
    Pose pose;
};
}  // namespace geometry_msgs
namespace ros { namespace stub {
template <> inline std::shared_ptr<const geometry_msgs::PoseStamped> synthetic<geometry_msgs::PoseStamped>() {
    auto m = std::make_shared<geometry_msgs::PoseStamped>();
    m->pose.position.x = 4.0; m->pose.position.y = 2.5; m->pose.position.z = 1.5;    // the goal
    return m;
}
} }
