"""Host-side adapters on either side of the hot path.  Pure data
re-packing on the CPU -- no solver arithmetic, nothing here touches the device.

N2  front-end -> optimiser:  flatten the waypoint lists the reference's searchers return
    (RRTStar::getOptimalPath, rrt_star.cpp:299-302; KinoAstar::retrievePath node positions + node durations,
    kino_astar.cpp:473-490,107,124,236) into the C ABI's CSR layout, including the reference's empty-path
    edge case (RRT* leaves optimal_path_ empty when the first feasible path is never improved,
    rrt_star.cpp:348-367 vs :386-394 -- SURVEY H8): such paths are dropped and reported, not solved.
    downsample_dense_path thins A*'s one-point-per-cell output (a_star.cpp:179-189) to corners + bounded spacing;
    refine_with_mid_knots adds mid-segment corridor samples by knot insertion (SURVEY 8-a').
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
    quadrotor_msgs/PolynomialTrajectory as a plain dict.  A binding of the C-ABI packer uavqp_pack_polynomial_trajectory
    (include/uavqp.h; host function, no device) -- the same code a C++ node reaches through cpp/traj_adapters.h."""
    import ctypes
    from . import _lib
    times = np.ascontiguousarray(times, dtype=np.float64).reshape(-1)
    m, nc = times.size, 2 * r
    c = np.ascontiguousarray(coeff_traj, dtype=np.float64).reshape(-1)
    if c.size != 3 * m * nc:
        raise ValueError(f"coeff_traj holds {c.size} values, expected 3 x {m} x {nc}")
    cx, cy, cz, t = np.zeros(m * nc), np.zeros(m * nc), np.zeros(m * nc), np.zeros(m)
    order = np.zeros(m, dtype=np.uint32)
    n_ord, n_seg = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = _lib.lib().uavqp_pack_polynomial_trajectory(int(r), int(m), c.ctypes.data, times.ctypes.data, cx.ctypes.data, cy.ctypes.data,
                                                     cz.ctypes.data, t.ctypes.data, order.ctypes.data, ctypes.addressof(n_ord),
                                                     ctypes.addressof(n_seg))
    _lib.check(rc, "uavqp_pack_polynomial_trajectory")
    return dict(trajectory_id=int(trajectory_id), action=ACTION_ADD, num_order=int(n_ord.value), num_segment=int(n_seg.value),
                start_yaw=float(start_yaw), final_yaw=float(final_yaw), coef_x=cx, coef_y=cy, coef_z=cz,
                time=t, mag_coeff=1.0, order=[int(v) for v in order], debug_info="")


def unpack_like_traj_server(msg):
    """What poly_traj_server.cpp:57-81 does with the message: per segment (cx, cy, cz, t)."""
    n = msg["num_order"] + 1
    return [(msg["coef_x"][i * n:(i + 1) * n], msg["coef_y"][i * n:(i + 1) * n], msg["coef_z"][i * n:(i + 1) * n],
             msg["time"][i]) for i in range(msg["num_segment"])]


def refine_with_mid_knots(seg_offsets, waypoints, times, corr_lo, corr_hi, k_mid, mid_half_width):
    """Mid-segment corridor samples by knot insertion (SURVEY.md section 8-a': "or K samples per segment").

    The device corridor solve bounds positions at KNOTS.  To bound the path inside a segment as well, every segment is
    split into k_mid + 1 equal-duration pieces; each inserted knot gets the box  chord point +- mid_half_width  (scalar
    or [n_segments] array), the original knots keep their boxes.  The refined problem is then an ordinary corridor batch
    (k_mid + 1 times the segments).  Note the semantics: the inserted knots are real spline knots (continuity up to
    derivative r-1 there), so the minimiser is taken over a larger spline space than a reference-formulation QP with
    extra inequality rows at those times -- its cost is lower or equal, the position bounds are identical.
    Returns dict(seg_offsets, waypoints, times, corr_lo, corr_hi, parent_segment) with parent_segment[new] = old."""
    so = np.asarray(seg_offsets, dtype=np.int64)
    wp = np.asarray(waypoints, dtype=np.float64).reshape(-1, 3)
    T = np.asarray(times, dtype=np.float64).reshape(-1)
    lo = np.asarray(corr_lo, dtype=np.float64).reshape(-1, 3)
    hi = np.asarray(corr_hi, dtype=np.float64).reshape(-1, 3)
    n = so.size - 1
    k_mid = int(k_mid)
    if k_mid < 0:
        raise ValueError("k_mid must be >= 0")
    hw = np.broadcast_to(np.asarray(mid_half_width, dtype=np.float64), (int(so[-1]),))
    f = k_mid + 1
    new_so = so * f
    n_rows = int(new_so[-1]) + n
    o_wp, o_lo, o_hi = np.zeros((n_rows, 3)), np.zeros((n_rows, 3)), np.zeros((n_rows, 3))
    o_T = np.repeat(T / f, f)
    parent = np.repeat(np.arange(int(so[-1])), f)
    frac = (np.arange(1, f) / f)[:, None]
    for b in range(n):
        s0, M = int(so[b]), int(so[b + 1] - so[b])
        r_old, r_new = s0 + b, int(new_so[b]) + b
        for i in range(M):
            a, c = wp[r_old + i], wp[r_old + i + 1]
            base = r_new + i * f
            o_wp[base], o_lo[base], o_hi[base] = a, lo[r_old + i], hi[r_old + i]
            if k_mid:
                mid = a + frac * (c - a)
                o_wp[base + 1:base + f] = mid
                o_lo[base + 1:base + f] = mid - hw[s0 + i]
                o_hi[base + 1:base + f] = mid + hw[s0 + i]
        o_wp[r_new + M * f], o_lo[r_new + M * f], o_hi[r_new + M * f] = wp[r_old + M], lo[r_old + M], hi[r_old + M]
    return dict(seg_offsets=new_so.astype(np.int32), waypoints=o_wp, times=o_T, corr_lo=o_lo, corr_hi=o_hi,
                parent_segment=parent)


def downsample_dense_path(path, max_spacing=2.0, max_segments=None, collinear_tol=1e-9):
    """A* returns every grid cell of the path (Astar::retrievePath, path_searching/src/a_star.cpp:179-189: one point
    per `resolution_` step); a polynomial segment per cell is neither needed nor well conditioned.  Keeps the end
    points and the corners (where the step direction changes), then adds points on straight runs so that no segment
    is longer than max_spacing (cf. RRT* step_length 1.5 m, test_minimum_jerk.launch:45), and finally thins uniformly
    to at most max_segments segments.  Returns the kept points [m, 3] (m >= 2 for an input of >= 2 points)."""
    p = np.asarray(path, dtype=np.float64).reshape(-1, 3)
    if p.shape[0] <= 2:
        return p.copy()
    d = np.diff(p, axis=0)
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    u = d / np.where(nrm > 0, nrm, 1.0)
    corner = np.linalg.norm(u[1:] - u[:-1], axis=1) > collinear_tol          # direction change at interior point i+1
    keep = [0] + [i + 1 for i in np.nonzero(corner)[0]] + [p.shape[0] - 1]
    out = [keep[0]]
    arc = np.concatenate([[0.0], np.cumsum(nrm[:, 0])])
    for a, b in zip(keep[:-1], keep[1:]):
        length = arc[b] - arc[a]
        n_piece = max(1, int(np.ceil(length / max_spacing - 1e-12)))
        for j in range(1, n_piece):                                          # dense-path points closest to the equal split
            target = arc[a] + length * j / n_piece
            idx = a + int(np.argmin(np.abs(arc[a:b + 1] - target)))
            if idx > out[-1]:
                out.append(idx)
        if b > out[-1]:
            out.append(b)
    if max_segments is not None and len(out) - 1 > max_segments:
        sel = np.unique(np.round(np.linspace(0, len(out) - 1, max_segments + 1)).astype(int))
        out = [out[i] for i in sel]
    return p[out].copy()
