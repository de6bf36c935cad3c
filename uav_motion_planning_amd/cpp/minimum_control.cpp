// MinimumControl over the C ABI (include/uavqp.h).  Mirrors the reference's behaviour at the boundary
// (minimum_control.cpp:127-202): inputs are not modified; returns false and prints the reference's
// message when the solver cannot be initialised or does not solve; on failure coef_1d_ keeps its
// previous contents; one axis per call; result layout coef[6*i + k] (ascending powers, :186).
// Not restated: the std::cout dump of P, q, A, lb, ub (:154-158) and OSQP's verbose banner.
#include <traj_optimization/minimum_control.h>

#include <iostream>
#include <vector>

#include "../../include/uavqp.h"

namespace traj_optimization
{
bool MinimumControl::solve(Eigen::VectorXd& pos_1d,
                        Eigen::Vector2d& bound_vel,
                        Eigen::Vector2d& bound_acc,
                        Eigen::VectorXd& time_vec)
{
    const int seg_num = static_cast<int>(time_vec.size());
    // The reference indexes lb_[3 + 4*(N-2)] out of range for fewer than 2 waypoints (:109-115);
    // here a malformed call is a clean failure.
    if (seg_num < 1 || pos_1d.size() != seg_num + 1)
    {
        std::cout << "solver init failed!" << std::endl;
        return false;
    }
    if (!ctx_)
    {
        if (uavqp_create(&ctx_, 0) != UAVQP_OK)
        {
            std::cout << "solver init failed! (" << uavqp_last_error() << ")" << std::endl;
            ctx_ = nullptr;
            return false;
        }
        // the three settings the reference passes to its solver (minimum_control.cpp:160-162)
        uavqp_settings st;
        uavqp_default_settings(&st);
        st.warm_start = 1;       // solver_.settings()->setWarmStart(true)
        st.eps_prim_inf = 1e-3;  // setPrimalInfeasibilityTollerance(1e-3)
        st.max_iter = 1000;      // setMaxIteration(1000)
        if (uavqp_set_settings(ctx_, &st) != UAVQP_OK)
        {
            std::cout << "solver init failed!" << std::endl;
            uavqp_destroy(ctx_);
            ctx_ = nullptr;
            return false;
        }
    }
    std::vector<double> pos(seg_num + 1), tv(seg_num), coef(2 * order_ * seg_num);
    for (int i = 0; i <= seg_num; i++) pos[i] = pos_1d[i];
    for (int i = 0; i < seg_num; i++) tv[i] = time_vec[i];
    const double bv[2] = {bound_vel[0], bound_vel[1]};
    const double ba[2] = {bound_acc[0], bound_acc[1]};
    int32_t status = 0;
    const int rc = uavqp_solve_axis_host(ctx_, order_, seg_num, pos.data(), bv, ba, nullptr, tv.data(), coef.data(), &status);
    if (rc != UAVQP_OK)
    {
        std::cout << "solver init failed! (" << uavqp_last_error() << ")" << std::endl;
        return false;
    }
    if (status != UAVQP_SOLVED)
    {
        std::cout << "solver solve failed!" << std::endl;
        return false;
    }
    coef_1d_.resize(static_cast<long>(coef.size()));
    for (size_t i = 0; i < coef.size(); i++) coef_1d_[static_cast<long>(i)] = coef[i];
    return true;
}

void MinimumControl::reset()
{
    coef_1d_.setZero();
}

Eigen::VectorXd MinimumControl::getCoef1d()
{
    return coef_1d_;
}

MinimumControl::~MinimumControl()
{
    if (ctx_) uavqp_destroy(ctx_);
}

} // namespace traj_optimization
