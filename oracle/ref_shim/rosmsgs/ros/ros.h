// Stand-in for <ros/ros.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref; ROS is absent from this image).  Just enough for
//   * the member DECLARATIONS of path_searching/kino_astar.h (Publisher, Subscriber, NodeHandle&), and
//   * /root/reference/src/planner/traj_server/src/poly_traj_server.cpp compiled WHOLE and unmodified: Time / Duration arithmetic
//     (:30), Timer / TimerEvent (:19,23), NodeHandle::createTimer / subscribe / advertise (:99-103), Publisher::publish (:54),
//     init / spin (:96,105).  spin() returns at once; publish() keeps the last message of each type where the C wrapper reads it.
#pragma once
#include <memory>
#include <string>

namespace ros {
struct Duration {
    double s = 0.0;
    Duration() {}
    explicit Duration(double v) : s(v) {}
    double toSec() const { return s; }
};
struct Time {
    double s = 0.0;
    Time() {}
    explicit Time(double v) : s(v) {}
    static Time now() { return Time(); }
    Duration operator-(const Time& o) const { return Duration(s - o.s); }
};
struct TimerEvent {};
class Timer {};
class Subscriber {};
namespace stub {
template <class M> M& last_published() { static M m; return m; }
template <class M> int& publish_count() { static int n = 0; return n; }
}  // namespace stub
class Publisher {
  public:
    template <class M> void publish(const M& m) const { stub::last_published<M>() = m; ++stub::publish_count<M>(); }
};
class NodeHandle {
  public:
    NodeHandle() {}
    explicit NodeHandle(const std::string&) {}
    template <class F> Timer createTimer(Duration, F) { return Timer(); }
    template <class M> Subscriber subscribe(const std::string&, int, void (*)(const std::shared_ptr<const M>&)) { return Subscriber(); }
    template <class M, class T> Subscriber subscribe(const std::string&, int, void (T::*)(const std::shared_ptr<const M>&), T*) { return Subscriber(); }
    template <class M> Publisher advertise(const std::string&, int) { return Publisher(); }
    template <class V> void param(const std::string&, V& v, const V& d) { v = d; }
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
}  // namespace ros
