// traj_adapters.h -- header-only C++ adapters on either side of the hot path, so that a C++
// planner (the only kind of caller the reference has: src/planner/test/src/test_minimum_jerk.cpp:28-173) needs no Python for
// anything around TrajOptimizer.  Pure host-side data re-packing: no solver arithmetic, nothing here touches the device.
//
// N2  searcher -> optimiser
//     flattenPaths          the waypoint lists the reference's searchers return (RRTStar::getOptimalPath -> std::vector<Eigen::Vector3d>,
//                           rrt_star.cpp:299-302, used at test_minimum_jerk.cpp:41-57; KinoAstar::retrievePath node positions with their
//                           node durations, kino_astar.cpp:473-490,107,124,236) -> the CSR layout of include/uavqp.h.  The reference's
//                           empty-path edge case (RRT* leaves optimal_path_ empty when the first feasible path is never improved,
//                           rrt_star.cpp:348-367 vs :386-394) is dropped and reported through `kept`, not solved.
//     boundaryFromOdometry  start velocity from odometry, every other boundary derivative zero (test_minimum_jerk.cpp:32-37,59-63).
//     downsampleDensePath   thins A*'s one-point-per-cell path (Astar::retrievePath, a_star.cpp:179-189) to corners + bounded spacing.
// N3  optimiser -> executor
//     packPolynomialTrajectory / fillMessage   the fields of quadrotor_msgs/PolynomialTrajectory
//                           (src/simulator/utils/quadrotor_msgs/msg/PolynomialTrajectory.msg:1-28) as trajCallback unpacks them
//                           (traj_server/src/poly_traj_server.cpp:57-81), over the C-ABI packer uavqp_pack_polynomial_trajectory.
//
// Vec3 is any type with operator[](int) returning something convertible to / assignable from double: Eigen::Vector3d,
// std::array<double, 3>, double[3] wrappers.
#ifndef UAVQP_TRAJ_ADAPTERS_H_
#define UAVQP_TRAJ_ADAPTERS_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/uavqp.h"

namespace traj_optimization {
namespace adapters {

// A batch in the layout TrajOptimizer::setWaypoints / setTimeAllocation take.
struct FlatBatch {
    std::vector<int32_t> wp_offsets;   // [n_traj + 1] first waypoint row of each trajectory (setWaypoints)
    std::vector<int32_t> seg_offsets;  // [n_traj + 1] first segment of each trajectory (the C ABI's CSR offsets)
    std::vector<double> xyz;           // [sum_b n_waypoints_b][3]
    std::vector<double> times;         // [sum_b M_b]
    std::vector<int> kept;             // indices of the input paths that made it into the batch, in batch order
    int n_traj() const { return static_cast<int>(kept.size()); }
    int max_segments() const {
        int m = 0;
        for (size_t b = 0; b + 1 < seg_offsets.size(); ++b) m = std::max(m, seg_offsets[b + 1] - seg_offsets[b]);
        return m;
    }
};

// paths[i]: the waypoints of query i; durations (optional): per-segment durations of path i (kino-A* node durations), else
// default_duration per segment -- the reference's constant allocation (test_minimum_jerk.cpp:65-71).  Paths with fewer than two
// waypoints cannot define a segment and are skipped (`kept` tells which survived).  A path whose duration list has the wrong length
// is skipped as well and its index returned in *bad_durations (if given).
// sort_by_segments: order the batch by descending segment count (stable) -- waves of the ragged kernel then hold trajectories of
// similar length; `kept` maps back.
template <class Vec3>
FlatBatch flattenPaths(const std::vector<std::vector<Vec3>>& paths, const std::vector<std::vector<double>>* durations = nullptr,
                       double default_duration = 1.0, bool sort_by_segments = false, std::vector<int>* bad_durations = nullptr) {
    std::vector<int> order(paths.size());
    std::iota(order.begin(), order.end(), 0);
    if (sort_by_segments)
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return paths[a].size() > paths[b].size(); });
    FlatBatch out;
    out.wp_offsets.push_back(0);
    out.seg_offsets.push_back(0);
    for (int i : order) {
        const std::vector<Vec3>& p = paths[i];
        if (p.size() < 2) continue;
        const int m = static_cast<int>(p.size()) - 1;
        if (durations) {
            if (static_cast<size_t>(i) >= durations->size() || (*durations)[i].size() != static_cast<size_t>(m)) {
                if (bad_durations) bad_durations->push_back(i);
                continue;
            }
            out.times.insert(out.times.end(), (*durations)[i].begin(), (*durations)[i].end());
        } else {
            out.times.insert(out.times.end(), static_cast<size_t>(m), default_duration);
        }
        for (const Vec3& w : p)
            for (int c = 0; c < 3; ++c) out.xyz.push_back(static_cast<double>(w[c]));
        out.kept.push_back(i);
        out.wp_offsets.push_back(out.wp_offsets.back() + m + 1);
        out.seg_offsets.push_back(out.seg_offsets.back() + m);
    }
    return out;
}

// bc [n_traj][2][r - 1][3] = [start | end][vel, acc(, jerk)][xyz]: start velocity from odometry (start_velocity [n_traj][3] or
// nullptr = at rest), everything else zero.
inline std::vector<double> boundaryFromOdometry(int n_traj, int r, const double* start_velocity = nullptr) {
    std::vector<double> bc(static_cast<size_t>(n_traj > 0 ? n_traj : 0) * 2 * (r - 1) * 3, 0.0);
    if (start_velocity)
        for (int b = 0; b < n_traj; ++b)
            for (int c = 0; c < 3; ++c) bc[(static_cast<size_t>(b) * 2 * (r - 1)) * 3 + c] = start_velocity[3 * b + c];
    return bc;
}

// The fields of quadrotor_msgs/PolynomialTrajectory (PolynomialTrajectory.msg:1-28) as plain data: what a node assigns to the
// generated message type (fillMessage below does exactly that) -- usable, and testable, without ROS.
struct PolynomialTrajectoryFields {
    enum : uint32_t { ACTION_ADD = 1, ACTION_ABORT = 2, ACTION_WARN_START = 3, ACTION_WARN_FINAL = 4, ACTION_WARN_IMPOSSIBLE = 5 };
    uint32_t trajectory_id = 1;   // "starts from 1" (PolynomialTrajectory.msg:3-4)
    uint32_t action = ACTION_ADD;
    uint32_t num_order = 0, num_segment = 0;
    double start_yaw = 0.0, final_yaw = 0.0;
    std::vector<double> coef_x, coef_y, coef_z, time;
    double mag_coeff = 1.0;
    std::vector<uint32_t> order;
    std::string debug_info;
};

// One trajectory of the solver output (TrajOptimizer::getPolyCoeff() + 3 * 2r * segOffset(b): [axis][segment][2r], ascending powers)
// and its durations -> the message fields.  false = invalid arguments (r, n_seg, a non-positive duration).
inline bool packPolynomialTrajectory(const double* coeff_traj, const double* times, int r, int n_seg, PolynomialTrajectoryFields& out,
                                     uint32_t trajectory_id = 1, double start_yaw = 0.0, double final_yaw = 0.0) {
    if (n_seg < 1 || (r != 3 && r != 4)) return false;
    const size_t n = static_cast<size_t>(n_seg) * 2 * r;
    out.coef_x.assign(n, 0.0);
    out.coef_y.assign(n, 0.0);
    out.coef_z.assign(n, 0.0);
    out.time.assign(static_cast<size_t>(n_seg), 0.0);
    out.order.assign(static_cast<size_t>(n_seg), 0u);
    if (uavqp_pack_polynomial_trajectory(r, n_seg, coeff_traj, times, out.coef_x.data(), out.coef_y.data(), out.coef_z.data(),
                                         out.time.data(), out.order.data(), &out.num_order, &out.num_segment) != UAVQP_OK)
        return false;
    out.trajectory_id = trajectory_id;
    out.action = PolynomialTrajectoryFields::ACTION_ADD;
    out.start_yaw = start_yaw;
    out.final_yaw = final_yaw;
    out.mag_coeff = 1.0;
    out.debug_info.clear();
    return true;
}

// Msg = quadrotor_msgs::PolynomialTrajectory (or anything with its field names); header.stamp stays the caller's (ros::Time::now()).
template <class Msg>
void fillMessage(const PolynomialTrajectoryFields& f, Msg& msg) {
    msg.trajectory_id = f.trajectory_id;
    msg.action = f.action;
    msg.num_order = f.num_order;
    msg.num_segment = f.num_segment;
    msg.start_yaw = f.start_yaw;
    msg.final_yaw = f.final_yaw;
    msg.coef_x.assign(f.coef_x.begin(), f.coef_x.end());
    msg.coef_y.assign(f.coef_y.begin(), f.coef_y.end());
    msg.coef_z.assign(f.coef_z.begin(), f.coef_z.end());
    msg.time.assign(f.time.begin(), f.time.end());
    msg.mag_coeff = f.mag_coeff;
    msg.order.assign(f.order.begin(), f.order.end());
    msg.debug_info = f.debug_info;
}

// A* returns every grid cell of the path (one point per `resolution_` step); a polynomial segment per cell is neither needed nor
// well conditioned.  Keeps the end points and the corners (where the step direction changes), then adds points on straight runs so
// that no segment is longer than max_spacing (cf. RRT* step_length 1.5 m, test_minimum_jerk.launch:45), and finally thins
// uniformly to at most max_segments segments (0 = no cap).  Returns the indices of the kept points (>= 2 for >= 2 input points).
template <class Vec3>
std::vector<int> downsampleDensePathIndices(const std::vector<Vec3>& path, double max_spacing = 2.0, int max_segments = 0,
                                            double collinear_tol = 1e-9) {
    const int n = static_cast<int>(path.size());
    std::vector<int> out;
    if (n <= 2) {
        for (int i = 0; i < n; ++i) out.push_back(i);
        return out;
    }
    std::vector<double> len(n - 1), arc(n, 0.0);
    std::vector<double> u(3 * static_cast<size_t>(n - 1));
    for (int i = 0; i + 1 < n; ++i) {
        double d[3], s = 0.0;
        for (int c = 0; c < 3; ++c) { d[c] = static_cast<double>(path[i + 1][c]) - static_cast<double>(path[i][c]); s += d[c] * d[c]; }
        len[i] = std::sqrt(s);
        const double inv = len[i] > 0.0 ? 1.0 / len[i] : 1.0;
        for (int c = 0; c < 3; ++c) u[3 * i + c] = d[c] * inv;
        arc[i + 1] = arc[i] + len[i];
    }
    std::vector<int> keep;
    keep.push_back(0);
    for (int i = 0; i + 2 < n; ++i) {
        double s = 0.0;
        for (int c = 0; c < 3; ++c) { const double e = u[3 * (i + 1) + c] - u[3 * i + c]; s += e * e; }
        if (std::sqrt(s) > collinear_tol) keep.push_back(i + 1);
    }
    keep.push_back(n - 1);
    out.push_back(keep[0]);
    for (size_t k = 0; k + 1 < keep.size(); ++k) {
        const int a = keep[k], b = keep[k + 1];
        const double length = arc[b] - arc[a];
        const int n_piece = std::max(1, static_cast<int>(std::ceil(length / max_spacing - 1e-12)));
        for (int j = 1; j < n_piece; ++j) {   // the dense-path point closest to the equal split
            const double target = arc[a] + length * j / n_piece;
            int best = a;
            for (int q = a; q <= b; ++q)
                if (std::fabs(arc[q] - target) < std::fabs(arc[best] - target)) best = q;
            if (best > out.back()) out.push_back(best);
        }
        if (b > out.back()) out.push_back(b);
    }
    if (max_segments > 0 && static_cast<int>(out.size()) - 1 > max_segments) {
        std::vector<int> sel;
        for (int k = 0; k <= max_segments; ++k) {
            const int idx = static_cast<int>(std::nearbyint(static_cast<double>(k) * (out.size() - 1) / max_segments));
            if (sel.empty() || out[idx] != sel.back()) sel.push_back(out[idx]);
        }
        out.swap(sel);
    }
    return out;
}

template <class Vec3>
std::vector<Vec3> downsampleDensePath(const std::vector<Vec3>& path, double max_spacing = 2.0, int max_segments = 0, double collinear_tol = 1e-9) {
    std::vector<Vec3> out;
    for (int i : downsampleDensePathIndices(path, max_spacing, max_segments, collinear_tol)) out.push_back(path[i]);
    return out;
}

}  // namespace adapters
}  // namespace traj_optimization
#endif
