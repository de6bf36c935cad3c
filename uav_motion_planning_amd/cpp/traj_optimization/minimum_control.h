// Drop-in replacement header for the reference's
//   src/planner/traj_optimization/include/traj_optimization/minimum_control.h:10-48
// Same include path, namespace, class name, public signatures, default ctor/dtor and Ptr typedef, so
// src/planner/test/src/test_minimum_jerk.cpp and test_qpsolve.cpp recompile unchanged.  The private
// section differs: the OsqpEigen solver and the Eigen sparse P/A/l/u members are replaced by an
// opaque handle of the MI355X back-end's C ABI (include/uavqp.h) -- callers never touch them.
#ifndef MINIMUM_CONTROL_H_
#define MINIMUM_CONTROL_H_

#if defined(__has_include)
#if __has_include(<ros/ros.h>)
#include <ros/ros.h>  // test_qpsolve.cpp:5-6 gets ros:: only through this header
#endif
#endif
#include <Eigen/Eigen>
#include <memory>

struct uavqp_ctx;

namespace traj_optimization {
class MinimumControl
{
    private:
        uavqp_ctx* ctx_ = nullptr;  // created on the first solve(), as the reference builds its OSQP workspace there
        int order_ = 3;             // 3 = min-jerk: what the reference implements (minimum_control.cpp:9-17)
        Eigen::VectorXd coef_1d_;

    public:
        bool solve(Eigen::VectorXd& pos_1d,
                Eigen::Vector2d& bound_vel,
                Eigen::Vector2d& bound_acc,
                Eigen::VectorXd& time_vec);

        /* helper function */
        void reset();
        Eigen::VectorXd getCoef1d();

        /* extension (not in the reference): 4 = min-snap with zero boundary jerk */
        void setOrder(int r) { order_ = r; }

        MinimumControl() {};
        ~MinimumControl();
        MinimumControl(const MinimumControl&) = delete;
        MinimumControl& operator=(const MinimumControl&) = delete;

        typedef std::shared_ptr<MinimumControl> Ptr;

}; // class MinimumControl

} // namespace traj_optimization

#endif // MINIMUM_CONTROL_H_
