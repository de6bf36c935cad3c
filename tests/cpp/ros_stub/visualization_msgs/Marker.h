// Test-only stand-in (see ros/ros.h in this directory).
#pragma once
#include <geometry_msgs/Point.h>
#include <ros/ros.h>
#include <string>
#include <vector>
namespace visualization_msgs {
struct Marker {
    enum { LINE_STRIP = 4, ADD = 0 };
    struct Header { std::string frame_id; ros::Time stamp; } header;
    std::string ns;
    int type = 0, action = 0;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    struct Color { float r = 0, g = 0, b = 0, a = 0; } color;
    std::vector<geometry_msgs::Point> points;
};
}  // namespace visualization_msgs
