// Test-only stand-in for <ros/ros.h>: just what the reference's two callers of MinimumControl use
// (src/planner/test/src/test_qpsolve.cpp:5-6,20 and test_minimum_jerk.cpp:8-11,177-209), so that the reference's own
// programs can be compiled UNMODIFIED against the drop-in MinimumControl header (ROS is absent from this image).
// spin() does not block: it delivers ONE synthetic odometry message and ONE synthetic goal to the subscribed callbacks
// (test_minimum_jerk.cpp's GoalCallback then runs the reference's real call pattern: one object, solve() for x, y, z,
// getCoef1d() after each, reset() at the end) and returns.
#pragma once
#include <cmath>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace ros {
inline void init(int&, char**, const std::string&) {}
struct Time {
    static Time now() { return Time(); }
};
namespace stub {
inline std::vector<std::function<void()>>& deliveries() {
    static std::vector<std::function<void()>> d;
    return d;
}
template <class M> std::shared_ptr<const M> synthetic();   // specialised by the message stubs
}  // namespace stub
class Subscriber {};
class Publisher {
  public:
    template <class M> void publish(const M& m) const {   // what the caller hands to rviz, summarised on stdout for the test
        std::cout << "published " << m.points.size() << " points";
        if (!m.points.empty())
            std::cout << ", first (" << m.points.front().x << ", " << m.points.front().y << ", " << m.points.front().z << "), last ("
                      << m.points.back().x << ", " << m.points.back().y << ", " << m.points.back().z << ")";
        std::cout << std::endl;
    }
};
class NodeHandle {
  public:
    NodeHandle() {}
    explicit NodeHandle(const std::string&) {}
    template <class M>
    Subscriber subscribe(const std::string&, int, void (*cb)(const std::shared_ptr<const M>&)) {
        stub::deliveries().push_back([cb]() { cb(stub::synthetic<M>()); });
        return Subscriber();
    }
    template <class M> Publisher advertise(const std::string&, int) { return Publisher(); }
};
inline void spin() {
    // odometry first (registered second in test_minimum_jerk.cpp:180-181), then the goal
    auto& d = stub::deliveries();
    for (auto it = d.rbegin(); it != d.rend(); ++it) (*it)();
    d.clear();
}
}  // namespace ros
