// Drop-in replacement for the reference header
//   src/planner/traj_optimization/include/traj_optimization/minimum_control.h:10-48
// Same include path, namespace, class name, public method signatures, default constructor / destructor and
// Ptr typedef, so src/planner/test/src/test_minimum_jerk.cpp and test_qpsolve.cpp recompile unchanged.
// What differs is private: the OsqpEigen solver and the Eigen sparse P / A / l / u members are gone; the
// object holds an opaque handle of the MI355X back-end's C ABI (include/uavqp.h) instead.
#ifndef UAVQP_DROPIN_MINIMUM_CONTROL_H_
#define UAVQP_DROPIN_MINIMUM_CONTROL_H_

#if defined(__has_include)
#if __has_include(<ros/ros.h>)
#include <ros/ros.h>  // test_qpsolve.cpp:5-6 reaches ros:: only through this header
#endif
#endif
#include <Eigen/Eigen>
#include <memory>

struct uavqp_ctx;

namespace traj_optimization {

class MinimumControl {
  public:
    typedef std::shared_ptr<MinimumControl> Ptr;

    MinimumControl() {}
    ~MinimumControl();
    MinimumControl(const MinimumControl&) = delete;
    MinimumControl& operator=(const MinimumControl&) = delete;

    // One axis per call, exactly as the reference: waypoint coordinates, (start, end) velocity and
    // acceleration, segment durations.  Inputs are not modified.  false = nothing solved, previous
    // coefficients kept.
    bool solve(Eigen::VectorXd& pos_1d, Eigen::Vector2d& bound_vel, Eigen::Vector2d& bound_acc, Eigen::VectorXd& time_vec);
    void reset();                  // zeroes the stored coefficients
    Eigen::VectorXd getCoef1d();   // copy; coef[6*i + k] multiplies t^k of segment i

    // extension, not in the reference: 4 = min-snap (zero boundary jerk); default 3 = min-jerk
    void setOrder(int r) { order_ = r; }

  private:
    uavqp_ctx* ctx_ = nullptr;  // created by the first solve(), like the reference's OSQP workspace
    int order_ = 3;
    Eigen::VectorXd coef_1d_;
};

}  // namespace traj_optimization

#endif  // UAVQP_DROPIN_MINIMUM_CONTROL_H_
