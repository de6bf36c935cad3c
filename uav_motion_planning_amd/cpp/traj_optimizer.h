// TrajOptimizer -- batch facade with the method names BASELINE.json's north star asks for
// (setWaypoints / setTimeAllocation / solve / getPolyCoeff).  The reference has no such class
// (its only optimiser is traj_optimization::MinimumControl); this is the batch-oriented
// entry a planner would call once per replanning cycle for all candidate trajectories and all 3 axes.
// Header-only, no Eigen: plain pointers over the C ABI (include/uavqp.h).
#ifndef UAVQP_TRAJ_OPTIMIZER_H_
#define UAVQP_TRAJ_OPTIMIZER_H_

#include <cstdint>
#include <iostream>
#include <vector>

#include "../../include/uavqp.h"

namespace traj_optimization {

class TrajOptimizer {
  public:
    explicit TrajOptimizer(int order = 4, int device = 0) : order_(order), device_(device) {}
    ~TrajOptimizer() { if (ctx_) uavqp_destroy(ctx_); }
    TrajOptimizer(const TrajOptimizer&) = delete;
    TrajOptimizer& operator=(const TrajOptimizer&) = delete;

    // xyz: [sum_b n_waypoints_b][3]; wp_offsets[n_traj+1]: first waypoint row of each trajectory
    // (what a list of A* / kino-A* / RRT* paths flattens to).
    void setWaypoints(const double* xyz, const int32_t* wp_offsets, int n_traj) {
        n_traj_ = n_traj;
        seg_offsets_.resize(n_traj + 1);
        for (int b = 0; b <= n_traj; ++b) seg_offsets_[b] = wp_offsets[b] - b;  // M_b = n_waypoints_b - 1
        wp_.assign(xyz, xyz + 3 * static_cast<size_t>(wp_offsets[n_traj]));
    }
    // T: [sum_b M_b] segment durations in trajectory order.
    void setTimeAllocation(const double* T) { T_.assign(T, T + (n_traj_ > 0 ? seg_offsets_[n_traj_] : 0)); }
    // bc: [n_traj][2][order-1][3] = [start|end][vel, acc(, jerk)][xyz]; default all zero.
    void setBoundary(const double* bc) { bc_.assign(bc, bc + static_cast<size_t>(n_traj_) * 2 * (order_ - 1) * 3); }
    // Optional corridor (north-star extension): boxes lo <= p <= hi, [sum_b n_waypoints_b][3] like xyz, replace the
    // interior-waypoint equalities; first/last rows of a trajectory are ignored.  nullptr restores the equalities.
    void setCorridor(const double* lo, const double* hi) {
        lo_.clear();
        hi_.clear();
        if (lo && hi) {
            lo_.assign(lo, lo + wp_.size());
            hi_.assign(hi, hi + wp_.size());
        }
    }

    bool solve() {
        if (n_traj_ <= 0 || T_.size() != static_cast<size_t>(seg_offsets_[n_traj_])) return false;
        if (!ctx_ && uavqp_create(&ctx_, device_) != UAVQP_OK) {
            std::cout << "solver init failed! (" << uavqp_last_error() << ")" << std::endl;
            ctx_ = nullptr;
            return false;
        }
        if (bc_.empty()) bc_.assign(static_cast<size_t>(n_traj_) * 2 * (order_ - 1) * 3, 0.0);
        coef_.assign(static_cast<size_t>(3) * 2 * order_ * seg_offsets_[n_traj_], 0.0);
        status_.assign(n_traj_, 0);
        const int rc = lo_.empty()
            ? uavqp_solve_batch_host(ctx_, order_, n_traj_, 0, 0, seg_offsets_.data(), wp_.data(), T_.data(), bc_.data(),
                                     coef_.data(), status_.data())
            : uavqp_solve_corridor_batch_host(ctx_, order_, n_traj_, 0, 0, seg_offsets_.data(), wp_.data(), T_.data(), bc_.data(),
                                              lo_.data(), hi_.data(), coef_.data(), status_.data(), nullptr);
        if (rc != UAVQP_OK) {
            std::cout << "solver solve failed! (" << uavqp_last_error() << ")" << std::endl;
            return false;
        }
        for (int32_t s : status_) if (s != UAVQP_SOLVED) return false;
        return true;
    }
    // Flat coefficients: trajectory b starts at 3*2r*segOffset(b), layout [axis][segment][2r], ascending powers.
    const double* getPolyCoeff() const { return coef_.data(); }
    const double* getPolyCoeff(int traj, int axis) const {
        const int M = seg_offsets_[traj + 1] - seg_offsets_[traj];
        return coef_.data() + static_cast<size_t>(3) * 2 * order_ * seg_offsets_[traj] + static_cast<size_t>(axis) * 2 * order_ * M;
    }
    int segOffset(int traj) const { return seg_offsets_[traj]; }
    const std::vector<int32_t>& status() const { return status_; }

  private:
    int order_, device_, n_traj_ = 0;
    uavqp_ctx* ctx_ = nullptr;
    std::vector<int32_t> seg_offsets_, status_;
    std::vector<double> wp_, T_, bc_, coef_, lo_, hi_;
};

}  // namespace traj_optimization
#endif
