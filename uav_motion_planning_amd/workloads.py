"""Synthetic waypoint / time-allocation batches for BASELINE.json's configs.

Seed rule: numpy default_rng(20260925 + config index).  Map box, step lengths, velocities follow the
reference launch files: map 40 x 20 x 3 m and virtual ceiling 2.5 m (test_minimum_jerk.launch:6-8,37),
RRT* step_length 1.5 m (:45), start velocity from odometry, all other boundary derivatives zero
(test_minimum_jerk.cpp:32-37,59-63), constant 1.0 s per segment (test_minimum_jerk.cpp:65-71).
"""
import numpy as np

SEED0 = 20260925
BOX_LO = np.array([-20.0, -10.0, 0.5])
BOX_HI = np.array([20.0, 10.0, 2.5])


def _random_rotation_within(rng, d, max_angle):
    """Rotate unit vectors d [n,3] by a random angle <= max_angle about a random perpendicular axis."""
    n = d.shape[0]
    rnd = rng.normal(size=(n, 3))
    perp = rnd - np.sum(rnd * d, axis=1, keepdims=True) * d
    perp /= np.linalg.norm(perp, axis=1, keepdims=True) + 1e-300
    ang = rng.uniform(0.0, max_angle, size=(n, 1))
    return np.cos(ang) * d + np.sin(ang) * perp


def search_like_paths(rng, n_traj, n_seg, step=(1.0, 2.0), max_turn_deg=60.0):
    """'A*/RRT*-like' waypoint lists: persistent heading, bounded turn, reflected at the map box."""
    p = rng.uniform(BOX_LO, BOX_HI, size=(n_traj, 3))
    d = rng.normal(size=(n_traj, 3))
    d[:, 2] *= 0.2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    wps = [p.copy()]
    for _ in range(n_seg):
        d = _random_rotation_within(rng, d, np.deg2rad(max_turn_deg))
        q = p + d * rng.uniform(step[0], step[1], size=(n_traj, 1))
        for ax in range(3):  # reflect at the walls
            lo, hi = BOX_LO[ax], BOX_HI[ax]
            over, under = q[:, ax] > hi, q[:, ax] < lo
            q[over, ax] = 2 * hi - q[over, ax]
            q[under, ax] = 2 * lo - q[under, ax]
            d[over | under, ax] *= -1.0
        p = q
        wps.append(p.copy())
    return np.stack(wps, axis=1)  # [n_traj, n_seg+1, 3]


def uniform_batch(config_index, n_traj, n_seg, r, time_mode="reference", seed=None):
    """Uniform-M batch.  Returns dict(seg_offsets, waypoints[n,M+1,3], times[n,M], bc[n,2,r-1,3], r, M)."""
    rng = np.random.default_rng(SEED0 + config_index if seed is None else seed)
    wp = search_like_paths(rng, n_traj, n_seg)
    if time_mode == "reference":  # test_minimum_jerk.cpp:70: time_vec(i) = 1.0
        T = np.ones((n_traj, n_seg))
    elif time_mode == "distance":  # T_i = max(0.3, |dp| / 2 m/s)
        T = np.maximum(0.3, np.linalg.norm(np.diff(wp, axis=1), axis=2) / 2.0)
    elif time_mode == "wide":  # stress: T in [0.2, 5] s
        T = rng.uniform(0.2, 5.0, size=(n_traj, n_seg))
    else:
        raise ValueError(time_mode)
    bc = np.zeros((n_traj, 2, r - 1, 3))
    bc[:, 0, 0, :] = rng.uniform(-1.0, 1.0, size=(n_traj, 3))  # start velocity from odometry
    so = (np.arange(n_traj + 1) * n_seg).astype(np.int32)
    return dict(r=r, M=n_seg, seg_offsets=so, waypoints=wp, times=T, bc=bc)


def ragged_batch(config_index, n_traj, r, m_lo=4, m_hi=24, seed=None):
    """Config-4 style 'kino-A*-like' ragged batch: M ~ U{m_lo..m_hi}; node duration 0.3 s
    (sample_tau, test_kino_astar_searching.launch:52) except a final one-shot segment U(0.5, 2.0)
    (kino_astar.cpp:124); waypoints from double-integrator roll-outs with a in {-10,-5,0,5,10}^3,
    |v| <= 7 (launch :49-51, kino_astar.cpp:651-670)."""
    rng = np.random.default_rng(SEED0 + config_index if seed is None else seed)
    Ms = rng.integers(m_lo, m_hi + 1, size=n_traj)
    so = np.zeros(n_traj + 1, dtype=np.int32)
    so[1:] = np.cumsum(Ms)
    total = int(so[-1])
    wp = np.zeros((total + n_traj, 3))
    T = np.zeros(total)
    bc = np.zeros((n_traj, 2, r - 1, 3))
    acc_set = np.array([-10.0, -5.0, 0.0, 5.0, 10.0])
    for b in range(n_traj):
        M = int(Ms[b])
        p = rng.uniform(BOX_LO, BOX_HI)
        v = rng.uniform(-2.0, 2.0, size=3)
        bc[b, 0, 0] = v
        row = int(so[b]) + b
        wp[row] = p
        for i in range(M):
            tau = 0.3 if i < M - 1 else rng.uniform(0.5, 2.0)
            a = rng.choice(acc_set, size=3)
            p = p + v * tau + 0.5 * a * tau * tau
            v = np.clip(v + a * tau, -7.0, 7.0)
            wp[row + i + 1] = p
            T[so[b] + i] = tau
    return dict(r=r, M=0, seg_offsets=so, waypoints=wp, times=T, bc=bc)


def algorithmic_bytes(r, n_seg):
    """float64, 3 axes, equality-only: in = 8[3(M+1)+M+3*2*(r-1)], out = 8*3*2r*M."""
    return 8 * (3 * (n_seg + 1) + n_seg + 3 * 2 * (r - 1)) + 8 * 3 * 2 * r * n_seg


def corridor_boxes(batch, config_index=3, h_lo=0.3, h_hi=0.8, seed=None):
    """Config-3 style corridors: axis-aligned box of half-width h ~ U(0.3, 0.8) m around
    each interior waypoint (cf. pillar radii 0.5-0.7 and inflation 0.099, simulator.xml:26-27), replacing the
    waypoint equality.  Returns (lo, hi) in the waypoint layout; first/last waypoint rows are lo = hi = waypoint."""
    rng = np.random.default_rng(SEED0 + 100 + config_index if seed is None else seed)
    wp = np.asarray(batch["waypoints"], dtype=np.float64)
    flat = wp.reshape(-1, 3)
    h = rng.uniform(h_lo, h_hi, size=(flat.shape[0], 1))
    lo, hi = flat - h, flat + h
    so = np.asarray(batch["seg_offsets"], dtype=np.int64)
    first = so[:-1] + np.arange(so.size - 1)
    last = so[1:] + np.arange(so.size - 1)
    for idx in (first, last):
        lo[idx] = flat[idx]
        hi[idx] = flat[idx]
    return lo.reshape(wp.shape), hi.reshape(wp.shape)


def config3_rows(batch, rows_per_segment=2, h_pos=0.25, v_lim=3.5):
    """Config 3's "K = 2 mid-segment samples"  as general rows lo <= p_i^(d)(tau T_i) <= hi: slot 0 a position
    sample at mid-segment inside the chord midpoint +- h_pos, slot 1 a per-axis velocity limit there.  The reference hands any
    l <= A x <= u to OSQP (minimum_control.cpp:146-147) but builds equality rows only (:98-125): no reference data for these.
    Uniform batches.  Returns (tau [n M][K], deriv [n M][K] int32, lo [n M][K][3], hi [n M][K][3]) -- what bench.py --config 3 --rows 2
    times and tests/test_gpu_baseline_sizes.py checks at full size."""
    wp = np.asarray(batch["waypoints"], dtype=np.float64)
    n, M = wp.shape[0], wp.shape[1] - 1
    K = rows_per_segment
    tau = np.full((n * M, K), 0.5)
    drv = np.tile(np.array([0, 1], dtype=np.int32)[:K], (n * M, 1))
    mid = 0.5 * (wp[:, :-1] + wp[:, 1:]).reshape(n * M, 3)
    lo, hi = np.zeros((n * M, K, 3)), np.zeros((n * M, K, 3))
    lo[:, 0], hi[:, 0] = mid - h_pos, mid + h_pos
    if K > 1:
        lo[:, 1], hi[:, 1] = -v_lim, v_lim
    return tau, drv, lo, hi


def pillar_cloud(config_index=5, n_pillars=60, resolution=0.2, radius=(0.5, 0.7), height=3.0, box=None, seed=None,
                 keep_clear=None, clear_radius=1.0):
    """Obstacle point cloud in the style of the reference's random map (simulator map_generator,
    random_forest.cpp:203-228: vertical pillars rasterised at `resolution_`, points at voxel centres + 1e-2;
    pillar radii 0.5-0.7, simulator.xml:26-27) over the reference map box.  Only the shell of each pillar is kept
    (interior voxels can never be the nearest obstacle of a free waypoint).  keep_clear: optional [k,3] points
    (e.g. start positions) no pillar axis may come within `clear_radius` + radius of.  Returns [n_obs, 3] float64."""
    rng = np.random.default_rng(SEED0 + 200 + config_index if seed is None else seed)
    lo_b, hi_b = (BOX_LO, BOX_HI) if box is None else box
    pts = []
    kc = None if keep_clear is None else np.asarray(keep_clear, dtype=np.float64).reshape(-1, 3)[:, :2]
    placed, tries = 0, 0
    while placed < n_pillars and tries < 100 * n_pillars:
        tries += 1
        c = rng.uniform(lo_b[:2], hi_b[:2])
        rad = rng.uniform(*radius)
        if kc is not None and kc.size and np.min(np.linalg.norm(kc - c, axis=1)) < clear_radius + rad:
            continue
        c = np.floor(c / resolution) * resolution + resolution / 2.0
        nw = int(np.ceil(rad / resolution)) + 1
        g = (np.arange(-nw, nw + 1) + 0.5) * resolution
        xx, yy = np.meshgrid(g, g, indexing="ij")
        d = np.hypot(xx, yy)
        shell = (d <= rad) & (d > rad - 1.5 * resolution)
        zz = (np.arange(int(np.ceil(height / resolution))) + 0.5) * resolution
        xy = np.stack([xx[shell] + c[0], yy[shell] + c[1]], axis=1) + 1e-2
        p = np.concatenate([np.repeat(xy, zz.size, axis=0), np.tile(zz + 1e-2, xy.shape[0])[:, None]], axis=1)
        pts.append(p)
        placed += 1
    return np.concatenate(pts) if pts else np.zeros((0, 3))
