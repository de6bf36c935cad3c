"""CPU: the corridor-from-cloud restatement (oracle/ellipsoid.c oracle_corridor_box) against the reference's own
collision test restated beside it (oracle_is_collision_free, kino_astar.cpp:721-758), and the pillar-cloud generator."""
import numpy as np

from uav_motion_planning_amd import workloads as W

ROBOT_R, ROBOT_H = 0.4, 0.1


def test_pillar_cloud_is_deterministic_and_respects_keep_clear():
    a = W.pillar_cloud(5, n_pillars=20, resolution=0.25)
    b = W.pillar_cloud(5, n_pillars=20, resolution=0.25)
    assert a.shape[1] == 3 and a.shape[0] > 1000 and np.array_equal(a, b)
    assert a[:, 2].min() > 0.0 and a[:, 2].max() < 3.1
    start = np.array([[0.0, 0.0, 1.0]])
    c = W.pillar_cloud(5, n_pillars=40, resolution=0.25, keep_clear=start, clear_radius=2.0)
    assert np.min(np.linalg.norm(c[:, :2] - start[:, :2], axis=1)) >= 2.0 - 1e-9


def test_corridor_box_hover_widths_and_caps(oracle):
    # one obstacle straight ahead at 2 m, hover: E = diag(r, r, h), metric distance 2 / 0.4 = 5, margin 4
    g, lo, hi = oracle.corridor_box([0, 0, 1], [0, 0, 0], [[2.0, 0, 1]], ROBOT_R, ROBOT_H, 10.0)
    assert abs(g - 5.0) < 1e-14
    assert np.allclose(hi - [0, 0, 1], [4 * 0.4 / 3, 4 * 0.4 / 3, 4 * 0.1 / 3], rtol=1e-14)
    assert np.allclose(hi - [0, 0, 1], np.array([0, 0, 1]) - lo, rtol=1e-14)
    g, lo, hi = oracle.corridor_box([0, 0, 1], [0, 0, 0], [[2.0, 0, 1]], ROBOT_R, ROBOT_H, 0.2)
    assert np.allclose(hi - [0, 0, 1], [0.2, 0.2, 4 * 0.1 / 3])
    g, lo, hi = oracle.corridor_box([0, 0, 1], [0, 0, 0], np.zeros((0, 3)), ROBOT_R, ROBOT_H, 0.3)
    assert np.isinf(g) and np.allclose(hi - lo, 0.6)
    g, lo, hi = oracle.corridor_box([0, 0, 1], [0, 0, 0], [[0.1, 0, 1]], ROBOT_R, ROBOT_H, 0.3)   # inside the ellipsoid
    assert g < 1.0 and np.array_equal(lo, hi)


def test_corridor_box_guarantee_holds_for_the_reference_collision_test(oracle):
    rng = np.random.default_rng(11)
    obs = W.pillar_cloud(5, n_pillars=40, resolution=0.25)
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    n_free = 0
    for _ in range(150):
        p = rng.uniform(W.BOX_LO, W.BOX_HI)
        acc = rng.uniform(-10, 10, size=3)
        g, lo, hi = oracle.corridor_box(p, acc, obs, ROBOT_R, ROBOT_H, 0.8)
        free = oracle.is_collision_free(p, acc, obs, ROBOT_R, ROBOT_H)
        assert free == (g > 1.0)
        if not free:
            assert np.array_equal(lo, hi)
            continue
        n_free += 1
        h = hi - p
        for d in np.concatenate([corners, rng.uniform(-1, 1, size=(6, 3))]) * h * (1 - 1e-9):
            assert oracle.is_collision_free(p + d, acc, obs, ROBOT_R, ROBOT_H)
    assert n_free > 50
