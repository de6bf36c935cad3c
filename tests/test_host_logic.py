"""CPU: host-side logic that needs no device -- workload generators, facade bookkeeping, sharding."""
import numpy as np
import pytest

from uav_motion_planning_amd import TrajOptimizer
from uav_motion_planning_amd import distributed as D
from uav_motion_planning_amd import workloads as W


def test_algorithmic_bytes_match_survey_8d():
    assert W.algorithmic_bytes(4, 8) == 1960      # config 2: 424 in + 1536 out
    assert W.algorithmic_bytes(4, 7) == 392 + 1344  # config 1 (SURVEY quotes 376 in: arithmetic slip, 8*(24+7+18) = 392)
    assert W.algorithmic_bytes(3, 16) == 632 + 2304


def test_uniform_batch_is_seeded_and_inside_the_map():
    a = W.uniform_batch(2, 64, 8, 4, time_mode="distance")
    b = W.uniform_batch(2, 64, 8, 4, time_mode="distance")
    assert np.array_equal(a["waypoints"], b["waypoints"]) and np.array_equal(a["times"], b["times"])
    wp = a["waypoints"]
    assert wp.shape == (64, 9, 3) and a["bc"].shape == (64, 2, 3, 3)
    assert np.all(wp >= W.BOX_LO - 1e-9) and np.all(wp <= W.BOX_HI + 1e-9)
    step = np.linalg.norm(np.diff(wp, axis=1), axis=2)
    assert step.max() <= 2.0 + 1e-9
    assert np.all(a["times"] >= 0.3) and np.all(W.uniform_batch(2, 4, 8, 4)["times"] == 1.0)  # reference: 1.0 s
    assert np.all(a["bc"][:, 1] == 0) and np.all(a["bc"][:, 0, 1:] == 0)


def test_ragged_batch_layout():
    b = W.ragged_batch(4, 50, 4)
    so = b["seg_offsets"]
    Ms = np.diff(so)
    assert Ms.min() >= 4 and Ms.max() <= 24 and so[0] == 0
    assert b["waypoints"].shape == (so[-1] + 50, 3) and b["times"].shape == (so[-1],)
    for k in range(50):
        t = b["times"][so[k]:so[k + 1]]
        assert np.allclose(t[:-1], 0.3) and 0.5 <= t[-1] <= 2.0


def test_traj_optimizer_offset_bookkeeping():
    opt = TrajOptimizer(order=4)
    xyz = np.zeros((9 + 5 + 3, 3))
    opt.setWaypoints(xyz, wp_offsets=[0, 9, 14, 17])
    assert list(opt._so) == [0, 8, 12, 14]
    opt.setWaypoints(np.zeros((18, 3)), n_waypoints=9)
    assert list(opt._so) == [0, 8, 16]
    assert opt.solve() is False            # no time allocation yet: refuses before touching the device
    opt.setTimeAllocation(np.ones(5))
    assert opt.solve() is False            # wrong length


def test_shard_bounds():
    assert D.shard_bounds(4096, 8) == [512 * g for g in range(9)]
    assert D.shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    b = W.ragged_batch(4, 200, 4)
    bounds = D.shard_bounds_ragged(b["seg_offsets"], 8)
    assert bounds[0] == 0 and bounds[-1] == 200 and all(x <= y for x, y in zip(bounds, bounds[1:]))
    work = [b["seg_offsets"][hi] - b["seg_offsets"][lo] for lo, hi in zip(bounds, bounds[1:])]
    assert max(work) - min(work) <= 2 * 24
    sl = D.local_slice(b, bounds[2], bounds[3])
    assert sl["seg_offsets"][0] == 0 and sl["times"].size == sl["seg_offsets"][-1]
    assert sl["waypoints"].shape[0] == sl["seg_offsets"][-1] + (bounds[3] - bounds[2])


def test_adapters_flatten_paths_and_empty_rrt_star_path():
    from uav_motion_planning_amd import adapters as A
    paths = [np.arange(12.0).reshape(4, 3), np.zeros((0, 3)), np.ones((1, 3)), np.arange(6.0).reshape(2, 3)]
    f = A.flatten_paths(paths)                      # H8: empty / single-point paths are skipped, not solved
    assert list(f["kept"]) == [0, 3] and list(f["seg_offsets"]) == [0, 3, 4]
    assert f["waypoints"].shape == (6, 3) and np.all(f["times"] == 1.0)
    s = A.flatten_paths(paths, sort_by_segments=True)
    assert list(s["kept"]) == [0, 3] and list(s["seg_offsets"]) == [0, 3, 4]
    s2 = A.flatten_paths([paths[3], paths[0]], sort_by_segments=True)
    assert list(s2["kept"]) == [1, 0] and list(s2["seg_offsets"]) == [0, 3, 4]
    g = A.flatten_paths(paths[:1], durations=[[0.3, 0.3, 1.2]])
    assert list(g["times"]) == [0.3, 0.3, 1.2]
    bc = A.boundary_from_odometry(2, 4, start_velocity=[[1, 2, 3], [4, 5, 6]])
    assert bc.shape == (2, 2, 3, 3) and np.all(bc[:, 1] == 0) and list(bc[1, 0, 0]) == [4, 5, 6]


def test_polynomial_trajectory_packer_fields():
    """Field values and sizes of the message (PolynomialTrajectory.msg:1-28); what the reference's CONSUMER makes of them is checked
    on the reference's own poly_traj_server.cpp in tests/test_n3_packer_vs_reference_consumer.py."""
    from uav_motion_planning_amd import adapters as A
    import uav_motion_planning_amd as U
    r, m = 3, 4
    c = np.arange(3 * m * 2 * r, dtype=np.float64)
    msg = A.pack_polynomial_trajectory(c, [1.0, 2.0, 0.5, 1.5], r, trajectory_id=7)
    assert (msg["num_order"], msg["num_segment"], msg["action"], msg["trajectory_id"]) == (5, 4, 1, 7)
    assert msg["coef_x"].size == msg["coef_y"].size == msg["coef_z"].size == m * 2 * r and msg["order"] == [5] * m
    assert list(msg["time"]) == [1.0, 2.0, 0.5, 1.5] and msg["mag_coeff"] == 1.0
    # round trip through the consumer's unpacking rule (poly_traj_server.cpp:57-81): segment i of axis x is c[x][i][:]
    segs = A.unpack_like_traj_server(msg)
    cc = c.reshape(3, m, 2 * r)
    for i, (cx, cy, cz, t) in enumerate(segs):
        assert np.array_equal(cx, cc[0, i]) and np.array_equal(cy, cc[1, i]) and np.array_equal(cz, cc[2, i]) and t == msg["time"][i]
    with pytest.raises(U.UavqpError):
        A.pack_polynomial_trajectory(c, [1.0, 0.0, 0.5, 1.5], r)          # a zero duration: the server would walk off the end
    with pytest.raises(ValueError):
        A.pack_polynomial_trajectory(c[:-1], [1.0, 2.0, 0.5, 1.5], r)


def test_refine_with_mid_knots_layout():
    from uav_motion_planning_amd import adapters as A
    b = W.ragged_batch(5, 6, 4, m_lo=1, m_hi=5)
    so = b["seg_offsets"]
    lo, hi = W.corridor_boxes(b, config_index=5)
    ref = A.refine_with_mid_knots(so, b["waypoints"], b["times"], lo, hi, k_mid=2, mid_half_width=0.25)
    assert np.array_equal(ref["seg_offsets"], so * 3)
    assert ref["waypoints"].shape == (int(so[-1]) * 3 + 6, 3) and ref["times"].size == int(so[-1]) * 3
    wp = np.asarray(b["waypoints"]).reshape(-1, 3)
    for k in range(6):
        M = int(so[k + 1] - so[k])
        r_old, r_new = int(so[k]) + k, int(so[k]) * 3 + k
        for i in range(M + 1):                         # original knots keep position and box
            assert np.array_equal(ref["waypoints"][r_new + 3 * i], wp[r_old + i])
            assert np.array_equal(ref["corr_lo"][r_new + 3 * i], lo.reshape(-1, 3)[r_old + i])
        for i in range(M):                             # inserted knots: chord points, +-0.25, a third of the duration each
            third = wp[r_old + i] + (wp[r_old + i + 1] - wp[r_old + i]) / 3
            assert np.allclose(ref["waypoints"][r_new + 3 * i + 1], third)
            assert np.allclose(ref["corr_hi"][r_new + 3 * i + 1] - ref["corr_lo"][r_new + 3 * i + 1], 0.5)
            assert np.allclose(ref["times"][(int(so[k]) + i) * 3:(int(so[k]) + i) * 3 + 3], b["times"][int(so[k]) + i] / 3)
            assert np.all(ref["parent_segment"][(int(so[k]) + i) * 3:(int(so[k]) + i) * 3 + 3] == int(so[k]) + i)
    same = A.refine_with_mid_knots(so, b["waypoints"], b["times"], lo, hi, k_mid=0, mid_half_width=0.1)
    assert np.array_equal(same["waypoints"], wp) and np.array_equal(same["times"], b["times"])


def test_downsample_dense_astar_path():
    from uav_motion_planning_amd import adapters as A
    res = 0.1
    leg1 = np.stack([np.arange(0, 3.0 + 1e-9, res), np.zeros(31), np.ones(31)], axis=1)            # 3 m along x
    leg2 = np.stack([np.full(20, 3.0), np.arange(res, 2.0 + 1e-9, res), np.ones(20)], axis=1)      # 2 m along y
    leg3 = np.stack([3.0 + np.arange(res, 1.0 + 1e-9, res)] * 2 + [np.ones(10)], axis=1)
    leg3[:, 1] += 2.0 - 3.0                                                                         # diagonal from (3,2)
    dense = np.concatenate([leg1, leg2, leg3])
    out = A.downsample_dense_path(dense, max_spacing=2.0)
    assert np.array_equal(out[0], dense[0]) and np.array_equal(out[-1], dense[-1])
    for corner in ([3.0, 0.0, 1.0], [3.0, 2.0, 1.0]):
        assert np.any(np.all(np.isclose(out, corner), axis=1))
    assert np.max(np.linalg.norm(np.diff(out, axis=0), axis=1)) <= 2.0 + 1e-9
    assert 4 <= out.shape[0] <= 7
    assert A.downsample_dense_path(dense, max_spacing=2.0, max_segments=2).shape[0] == 3
    assert A.downsample_dense_path(dense[:2]).shape[0] == 2 and A.downsample_dense_path(dense[:1]).shape[0] == 1
    flat = A.flatten_paths([out, A.downsample_dense_path(dense[:15])])
    assert flat["seg_offsets"][-1] == out.shape[0] - 1 + 1
