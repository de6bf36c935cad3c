// Test-only stand-in for <ros/ros.h>: just what the reference's src/planner/test/src/test_qpsolve.cpp:5-6,20 calls, so that
// the reference's own test program can be compiled UNMODIFIED against the drop-in MinimumControl header (ROS is absent
// from this image).  spin() returns instead of blocking.
#pragma once
#include <string>
namespace ros {
inline void init(int&, char**, const std::string&) {}
class NodeHandle {
  public:
    NodeHandle() {}
    explicit NodeHandle(const std::string&) {}
};
inline void spin() {}
}  // namespace ros
