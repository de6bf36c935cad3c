"""Host-side adapters on either side of the hot path.  Pure data
re-packing on the CPU -- no solver arithmetic, nothing here touches the device.

N2  front-end -> optimiser:  flatten the waypoint lists the reference's searchers return
    (RRTStar::getOptimalPath, rrt_star.cpp:299-302; KinoAstar::retrievePath node positions + node durations,
    kino_astar.cpp:473-490,107,124,236) into the C ABI's CSR layout, including the reference's empty-path
    edge case (RRT* leaves optimal_path_ empty when the first feasible path is never improved,
    rrt_star.cpp:348-367 vs :386-394 -- SURVEY H8): such paths are dropped and reported, not solved.
N3  optimiser -> executor:   quadrotor_msgs/PolynomialTrajectory fields
    (src/simulator/utils/quadrotor_msgs/msg/PolynomialTrajectory.msg:1-28) exactly as poly_traj_server's
    trajCallback unpacks them (traj_server/src/poly_traj_server.cpp:57-81):
    coef_{x,y,z}[i * (num_order + 1) + j], time[i], num_order, num_segment.
"""
import numpy as np

ACTION_ADD = 1  # PolynomialTrajectory.msg:7


def flatten_paths(paths, durations=None, default_duration=1.0, sort_by_segments=False):
    """paths: list of [n_i, 3] waypoint arrays (a searcher's output per query); durations: optional list of
    per-segment durations (kino-A* node durations), else `default_duration` per segment -- the reference's
    constant allocation (test_minimum_jerk.cpp:65-71).
    Returns dict(seg_offsets, waypoints, times, kept) where `kept` are the indices of the paths that have at
    least two waypoints (the rest cannot define a segment and are skipped), in batch order.
    sort_by_segments: order the batch by descending segment count (stable).  The ragged device kernel runs one
    lane per trajectory, so a wave takes as long as its longest trajectory: a batch grouped by length solves
    2.7x faster (measured, 32768 trajectories with 4..24 segments: 158 -> 58 us); `kept` maps back."""
    order = range(len(paths))
    if sort_by_segments:
        order = sorted(order, key=lambda i: -np.asarray(paths[i]).reshape(-1, 3).shape[0])
    kept, wps, ts, so = [], [], [], [0]
    for i in order:
        p = np.asarray(paths[i], dtype=np.float64).reshape(-1, 3)
        if p.shape[0] < 2:
            continue
        m = p.shape[0] - 1
        if durations is not None:
            t = np.asarray(durations[i], dtype=np.float64).reshape(-1)
            if t.size != m:
                raise ValueError(f"path {i}: {m} segments but {t.size} durations")
        else:
            t = np.full(m, float(default_duration))
        kept.append(i)
        wps.append(p)
        ts.append(t)
        so.append(so[-1] + m)
    return dict(seg_offsets=np.asarray(so, dtype=np.int32),
                waypoints=np.concatenate(wps) if wps else np.zeros((0, 3)),
                times=np.concatenate(ts) if ts else np.zeros(0),
                kept=np.asarray(kept, dtype=np.int64))


def boundary_from_odometry(n_traj, r, start_velocity=None):
    """bc[n_traj][2][r-1][3]: start velocity from odometry, every other boundary derivative zero
    (test_minimum_jerk.cpp:32-37,59-63)."""
    bc = np.zeros((n_traj, 2, r - 1, 3))
    if start_velocity is not None:
        bc[:, 0, 0, :] = np.asarray(start_velocity, dtype=np.float64).reshape(n_traj, 3)
    return bc


def pack_polynomial_trajectory(coeff_traj, times, r, trajectory_id=1, start_yaw=0.0, final_yaw=0.0):
    """One trajectory of the solver output ([axis][segment][2r], ascending powers) -> the fields of
    quadrotor_msgs/PolynomialTrajectory as a plain dict (no ROS needed to build or test it)."""
    times = np.asarray(times, dtype=np.float64).reshape(-1)
    m, nc = times.size, 2 * r
    c = np.asarray(coeff_traj, dtype=np.float64).reshape(3, m, nc)
    return dict(trajectory_id=int(trajectory_id), action=ACTION_ADD, num_order=nc - 1, num_segment=m,
                start_yaw=float(start_yaw), final_yaw=float(final_yaw),
                coef_x=c[0].reshape(-1).copy(), coef_y=c[1].reshape(-1).copy(), coef_z=c[2].reshape(-1).copy(),
                time=times.copy(), mag_coeff=1.0, order=[nc - 1] * m, debug_info="")


def unpack_like_traj_server(msg):
    """What poly_traj_server.cpp:57-81 does with the message: per segment (cx, cy, cz, t)."""
    n = msg["num_order"] + 1
    return [(msg["coef_x"][i * n:(i + 1) * n], msg["coef_y"][i * n:(i + 1) * n], msg["coef_z"][i * n:(i + 1) * n],
             msg["time"][i]) for i in range(msg["num_segment"])]
