"""-m gpu: BASELINE.json's full sizes through size-independent properties (the binary128 oracle is too
slow for 4096+ trajectories): equality residual of the reference's own A x = b on a sample, C^(r-1)
continuity at every knot of every trajectory, linearity in the waypoints, translation and time-scaling
invariance, agreement between the two kernel variants, bitwise run-to-run determinism."""
import numpy as np
import pytest

import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W

pytestmark = pytest.mark.gpu


def poly_derivs(c, t, nd):
    """c[..., K] ascending powers; returns derivatives 0..nd-1 at time t[...] -> [..., nd]."""
    K = c.shape[-1]
    out = []
    for d in range(nd):
        acc = np.zeros(c.shape[:-1])
        for k in range(K - 1, d - 1, -1):
            f = 1.0
            for j in range(d):
                f *= (k - j)
            acc = acc * t + f * c[..., k]
        out.append(acc)
    return np.stack(out, axis=-1)


def check_continuity_and_interpolation(coef, batch, r, M, tol):
    n = batch["waypoints"].shape[0]
    c = coef.reshape(n, 3, M, 2 * r)
    T = batch["times"][:, None, :]                                  # [n,1,M]
    end = poly_derivs(c, np.broadcast_to(T, c.shape[:-1]), r)       # [n,3,M,r] derivatives at segment ends
    start = poly_derivs(c, np.zeros(c.shape[:-1]), r)
    wp = np.transpose(batch["waypoints"], (0, 2, 1))                # [n,3,M+1]
    scale = max(1.0, np.max(np.abs(coef)))
    assert np.max(np.abs(start[..., 0] - wp[..., :-1])) < tol * scale          # p_i(0) = w_i
    assert np.max(np.abs(end[..., 0] - wp[..., 1:])) < tol * scale             # p_i(T_i) = w_{i+1}
    assert np.max(np.abs(end[:, :, :-1, :] - start[:, :, 1:, :])) < tol * scale  # continuity of derivatives 0..r-1
    bc = batch["bc"]                                                # [n,2,r-1,3]
    assert np.max(np.abs(start[:, :, 0, 1:] - np.transpose(bc[:, 0], (0, 2, 1)))) < tol * scale
    assert np.max(np.abs(end[:, :, -1, 1:] - np.transpose(bc[:, 1], (0, 2, 1)))) < tol * scale


@pytest.mark.parametrize("r,M,n,mode", [(4, 8, 4096, "distance"), (4, 8, 4096, "reference"), (3, 16, 65536, "distance"),
                                        (4, 7, 1, "reference")])
def test_full_size_configs_properties(gpu_ctx, oracle, r, M, n, mode):
    b = W.uniform_batch(2, n, M, r, time_mode=mode)
    coef, st = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    assert np.all(st == U.UAVQP_SOLVED)
    check_continuity_and_interpolation(coef, b, r, M, 1e-9)
    # reference's own equality rows on a sample (oracle = checker)
    c = coef.reshape(n, 3, 2 * r * M)
    for k in np.linspace(0, n - 1, min(n, 16)).astype(int):
        for ax in range(3):
            res = oracle.residual(r, b["waypoints"][k, :, ax], b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax], b["times"][k], c[k, ax])
            assert res < 1e-8 * max(1.0, np.max(np.abs(c[k, ax])))
    # both kernels agree everywhere (generic lane-per-trajectory vs register-resident twisted)
    gpu_ctx.set_variant(1)
    coef1, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    gpu_ctx.set_variant(0)
    assert np.max(np.abs(coef1 - coef)) < 1e-9 * np.max(np.abs(coef))
    # bitwise determinism
    coef2, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    assert np.array_equal(coef, coef2)


def test_linearity_translation_time_scaling_4096(gpu_ctx):
    r, M, n = 4, 8, 4096
    a = W.uniform_batch(2, n, M, r, time_mode="distance")
    b = W.uniform_batch(2, n, M, r, time_mode="distance", seed=99)

    def solve(wp, T, bc):
        c, st = gpu_ctx.solve_batch_host(r, None, wp, T, bc, uniform_segments=M)
        assert np.all(st == U.UAVQP_SOLVED)
        return c.reshape(n, 3, M, 2 * r)
    ca = solve(a["waypoints"], a["times"], a["bc"])
    cb = solve(b["waypoints"], a["times"], b["bc"])
    # the minimiser is linear in (waypoints, boundary values) for a fixed time allocation
    cab = solve(2.0 * a["waypoints"] - 0.5 * b["waypoints"], a["times"], 2.0 * a["bc"] - 0.5 * b["bc"])
    assert np.max(np.abs(cab - (2.0 * ca - 0.5 * cb))) < 1e-9 * np.max(np.abs(ca))
    # translation only moves c0
    ct = solve(a["waypoints"] + np.array([1.0, -2.0, 0.5]), a["times"], a["bc"])
    d = ct - ca
    assert np.max(np.abs(d[..., 0] - np.array([1.0, -2.0, 0.5])[None, :, None])) < 1e-9
    assert np.max(np.abs(d[..., 1:])) < 1e-8 * np.max(np.abs(ca))
    # T -> sT with BC derivatives scaled s^-d: c_k -> c_k / s^k
    s = 1.5
    bcs = a["bc"] * np.array([s ** -(d + 1) for d in range(r - 1)])[None, None, :, None]
    cs = solve(a["waypoints"], a["times"] * s, bcs)
    assert np.max(np.abs(cs * s ** np.arange(2 * r) - ca)) < 1e-9 * np.max(np.abs(ca))


def test_ragged_config4_like_batch(gpu_ctx, oracle):
    """Config 4 shape: ragged M in [4, 24], kino-A*-like roll-outs; oracle on a sample, continuity on all."""
    r, n = 4, 2048
    b = W.ragged_batch(4, n, r)
    coef, st = gpu_ctx.solve_batch_host(r, b["seg_offsets"], b["waypoints"], b["times"], b["bc"])
    assert np.all(st == U.UAVQP_SOLVED)
    so = b["seg_offsets"]
    for k in np.linspace(0, n - 1, 24).astype(int):
        M = so[k + 1] - so[k]
        sub = dict(seg_offsets=np.array([0, M], dtype=np.int32), waypoints=b["waypoints"][so[k] + k:so[k + 1] + k + 1],
                   times=b["times"][so[k]:so[k + 1]], bc=b["bc"][k:k + 1])
        ref, _ = oracle.solve_exact_batch(r, sub["seg_offsets"], sub["waypoints"], sub["times"], sub["bc"])
        got = coef[24 * so[k]:24 * so[k + 1]]
        assert np.max(np.abs(got - ref)) < 1e-8 * np.max(np.abs(ref))


def test_empty_batch_and_single_segment(gpu_ctx, oracle):
    c, st = gpu_ctx.solve_batch_host(4, np.zeros(1, dtype=np.int32), np.zeros((0, 3)), np.zeros(0), np.zeros((0, 2, 3, 3)))
    assert c.size == 0 and st.size == 0
    b = W.uniform_batch(3, 33, 1, 4, time_mode="wide")
    got, st = gpu_ctx.solve_batch_host(4, None, b["waypoints"], b["times"], b["bc"], uniform_segments=1)
    ref, _ = oracle.solve_exact_batch(4, b["seg_offsets"], b["waypoints"], b["times"], b["bc"])
    assert np.all(st == U.UAVQP_SOLVED) and np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


def test_quarter_million_batch_throughput_shape_agrees_with_generic(gpu_ctx):
    """262 144 trajectories (64 x the benchmark batch) through the throughput shape (32 trajectories / wave,
    persistent tiles, LDS-DMA prefetch across tiles): continuity / interpolation everywhere and agreement with
    the generic lane-per-trajectory kernel on the whole batch."""
    r, M, n = 4, 8, 262144
    b = W.uniform_batch(2, n, M, r, time_mode="distance", seed=4242)
    gpu_ctx.set_variant(32)
    coef, st = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    gpu_ctx.set_variant(1)
    coef1, st1 = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    gpu_ctx.set_variant(0)
    assert np.all(st == U.UAVQP_SOLVED) and np.all(st1 == U.UAVQP_SOLVED)
    check_continuity_and_interpolation(coef, b, r, M, 1e-9)
    c = coef.reshape(n, -1)
    c1 = coef1.reshape(n, -1)
    rel = np.max(np.abs(c - c1), axis=1) / np.max(np.abs(c1), axis=1)
    assert rel.max() < 1e-10


@pytest.mark.parametrize("variant", [0, 1, 4, 8, 32])
def test_non_finite_waypoints_and_extreme_allocations(gpu_ctx, variant):
    """NaN / inf waypoints give UAVQP_NON_FINITE for that trajectory only; time ratios of 1:200 inside one
    trajectory still solve (they arise in re-allocation loops)."""
    r, M, n = 4, 8, 40
    b = W.uniform_batch(9, n, M, r, time_mode="distance")
    wp = b["waypoints"].copy()
    wp[3, 4, 1] = np.nan
    wp[17, 0, 2] = np.inf
    T = b["times"].copy()
    T[5] = [0.3, 0.3, 0.3, 60.0, 0.3, 0.3, 60.0, 0.3]
    gpu_ctx.set_variant(variant)
    coef, st = gpu_ctx.solve_batch_host(r, None, wp, T, b["bc"], uniform_segments=M)
    gpu_ctx.set_variant(0)
    assert st[3] == U.UAVQP_NON_FINITE and st[17] == U.UAVQP_NON_FINITE
    good = np.ones(n, dtype=bool)
    good[[3, 17]] = False
    assert np.all(st[good] == U.UAVQP_SOLVED)
    c = coef.reshape(n, -1)
    assert np.all(np.isfinite(c[good]))
    sub = dict(waypoints=wp[good], times=T[good], bc=b["bc"][good])
    check_continuity_and_interpolation(c[good].ravel(), sub, r, M, 1e-8)


def test_long_trajectories_generic_path(gpu_ctx, oracle):
    """M = 40 and 63 segments (beyond every specialised instantiation; SURVEY section 5 'long dimension')."""
    for M in (40, 63):
        b = W.uniform_batch(11, 6, M, 3, time_mode="distance")
        got, st = gpu_ctx.solve_batch_host(3, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
        ref, _ = oracle.solve_exact_batch(3, b["seg_offsets"], b["waypoints"], b["times"], b["bc"])
        assert np.all(st == U.UAVQP_SOLVED)
        assert np.max(np.abs(got - ref)) < 1e-8 * np.max(np.abs(ref))


@pytest.mark.parametrize("r,mx,n", [(4, 48, 5003), (3, 48, 2048), (4, 7, 40000)])
def test_waves_that_rank_their_own_window_deal_like_the_sort_kernel(gpu_ctx, r, mx, n):
    """Round 6: no window_sort_kernel launch in front of the pair kernel -- each of the 16 waves of a window ranks the window's 512 segment
    counts itself (descending count, ascending index: the same order in every wave, no atomics) and takes its 32 ranks.  A rank computed
    differently by two waves would solve a trajectory twice or never: statuses pre-filled with 0, outputs with NaN, everything compared bit for
    bit with the separate sort launch (setting 2) and the plain lane order (0).  Up to the longest trajectories the pair kernel takes (48
    segments), with zero-segment and over-long (invalid) trajectories, a last window of 395 / exactly 4 windows / many equal counts."""
    import torch
    rng = np.random.default_rng(7 * mx + r)
    Ms = rng.integers(0, mx + 6, size=n)                    # 0 and mx + 1 .. mx + 5: invalid input
    so = np.zeros(n + 1, dtype=np.int32)
    so[1:] = np.cumsum(Ms)
    tot = int(so[-1])
    wp = np.cumsum(rng.uniform(-1.0, 1.0, size=(tot + n, 3)), axis=0)
    T = rng.uniform(0.4, 2.0, size=tot)
    bc = rng.uniform(-1.0, 1.0, size=(n, 2, r - 1, 3))
    dev = torch.device("cuda", 0)
    d_so, d_wp, d_T, d_bc = (torch.from_numpy(x).to(dev) for x in (so, wp, T, bc))
    res = {}
    try:
        for mode in (1, 2, 0):
            gpu_ctx.set_settings(ragged_window_sort=mode)
            out = torch.full((tot * 6 * r,), np.nan, dtype=torch.float64, device=dev)
            st = torch.zeros(n, dtype=torch.int32, device=dev)
            gpu_ctx.solve_batch_device(r, n, 0, mx, d_so, d_wp, d_T, d_bc, out, st)
            gpu_ctx.synchronize()
            res[mode] = (out.cpu().numpy(), st.cpu().numpy())
    finally:
        gpu_ctx.set_settings(ragged_window_sort=1)
    ok = (Ms >= 1) & (Ms <= mx)
    assert np.array_equal(res[1][1] == U.UAVQP_SOLVED, ok) and np.all(res[1][1][~ok] == U.UAVQP_INVALID_INPUT)
    for mode in (2, 0):
        assert np.array_equal(res[mode][1], res[1][1]) and np.array_equal(res[mode][0], res[1][0], equal_nan=True), mode
    valid = np.repeat(ok, Ms * 6 * r)
    assert np.all(np.isfinite(res[1][0][valid])) and np.all(np.isnan(res[1][0][~valid]))


def test_ragged_dealing_by_segment_count_is_invisible_in_the_results(gpu_ctx):
    """Large ragged batches are dealt to the lanes in windows of 16 waves' trajectories by descending segment count
    (solve_generic2_kernel<R, LSORT>: a lane pair per trajectory, the default -- since round 6 every wave ranks its window itself, setting 2
    = window_sort_kernel in front as before; solve_generic_kernel<R, LSORT, NAX>: one lane per trajectory or per (trajectory, axis)).  Which lane solves a trajectory must not change a single bit:
    compare with the plain lane order (uavqp_settings.ragged_window_sort = 0) on a batch that needs two grid rounds, is
    not a multiple of the window, and contains single-segment and over-long (flagged invalid) trajectories.  The status
    buffer is pre-filled with 0 (no valid code): a trajectory the dealing dropped would keep it."""
    import torch
    r, n = 4, 140001
    rng = np.random.default_rng(44)
    Ms = rng.integers(1, 27, size=n)                       # max_segments = 24 below: 25 and 26 are invalid input
    so = np.zeros(n + 1, dtype=np.int32)
    so[1:] = np.cumsum(Ms)
    tot = int(so[-1])
    wp = np.cumsum(rng.uniform(-1.0, 1.0, size=(tot + n, 3)), axis=0)
    T = rng.uniform(0.4, 2.0, size=tot)
    bc = rng.uniform(-1.0, 1.0, size=(n, 2, r - 1, 3))
    dev = torch.device("cuda", 0)
    d_so, d_wp, d_T, d_bc = (torch.from_numpy(x).to(dev) for x in (so, wp, T, bc))

    def run():
        out = torch.full((tot * 24,), np.nan, dtype=torch.float64, device=dev)
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        gpu_ctx.solve_batch_device(r, n, 0, 24, d_so, d_wp, d_T, d_bc, out, st)
        gpu_ctx.synchronize()
        return out.cpu().numpy(), st.cpu().numpy()

    assert gpu_ctx.get_settings().ragged_window_sort == 1
    c_deal, st_deal = run()                                  # (round 6: the waves of the pair kernel rank their window themselves)
    gpu_ctx.set_settings(ragged_window_sort=2)               # the same dealing from window_sort_kernel, its own launch (rounds 2-5)
    try:
        c_sep, st_sep = run()
    finally:
        gpu_ctx.set_settings(ragged_window_sort=1)
    assert np.array_equal(st_sep, st_deal) and np.array_equal(c_sep, c_deal, equal_nan=True)
    gpu_ctx.set_settings(ragged_window_sort=0)
    try:
        c_plain, st_plain = run()
        gpu_ctx.set_settings(generic_lanes_per_traj=3)      # one lane per (trajectory, axis), windows of 336
        c_plain3, st_plain3 = run()
        gpu_ctx.set_settings(ragged_window_sort=1)
        c_deal3, st_deal3 = run()
        gpu_ctx.set_settings(generic_lanes_per_traj=1)      # one lane per trajectory, windows of 1024
        c_deal1, st_deal1 = run()
        gpu_ctx.set_settings(ragged_window_sort=0)
        c_plain1, st_plain1 = run()
    finally:
        gpu_ctx.set_settings(ragged_window_sort=1, generic_lanes_per_traj=0)
    assert np.array_equal(st_deal3, st_plain3) and np.array_equal(st_deal3, st_deal)
    assert np.array_equal(st_deal, st_plain)
    assert np.array_equal(st_deal == U.UAVQP_SOLVED, Ms <= 24)
    assert np.all(st_deal[Ms > 24] == U.UAVQP_INVALID_INPUT)
    valid = np.repeat(Ms <= 24, Ms * 24)
    assert np.array_equal(c_deal[valid], c_plain[valid])
    assert np.array_equal(c_deal3[valid], c_plain3[valid])
    assert np.array_equal(c_deal1[valid], c_plain1[valid]) and np.array_equal(st_deal1, st_deal) and np.array_equal(st_plain1, st_deal)
    # the three lane layouts order the eliminations differently: same solution to rounding
    scale = 1.0 + np.abs(c_deal1[valid])
    assert np.max(np.abs(c_deal[valid] - c_deal1[valid]) / scale) < 1e-9 and np.max(np.abs(c_deal3[valid] - c_deal1[valid]) / scale) < 1e-9
    assert np.all(np.isfinite(c_deal[valid])) and np.all(np.isnan(c_deal[~valid]))   # invalid ones are left untouched
    # spot check: waypoint interpolation of a few trajectories from both ends of the batch
    for k in (0, 1, n // 2, n - 2, n - 1):
        if Ms[k] > 24:
            continue
        c = c_deal[24 * so[k]:24 * so[k + 1]].reshape(3, Ms[k], 8)
        assert np.allclose(c[:, :, 0].T, wp[so[k] + k:so[k + 1] + k], rtol=0, atol=1e-9 * max(1.0, np.abs(wp).max()))


@pytest.mark.parametrize("r", [3, 4])
def test_pair_kernel_edges_long_ragged_unaligned_and_empty_trajectories(gpu_ctx, oracle, r):
    """The lane-pair kernel of ragged batches (qp_generic2.h) off its tuned shape: segment counts 1..58 in one batch (more than
    64 sixteen-byte units per trajectory: the extra staging loop), zero-segment trajectories in the CSR (flagged, nothing written),
    views that start at an odd double (only 8-byte aligned: the one-lane kernel must take them), and every lane layout against
    the exact oracle."""
    import torch
    rng = np.random.default_rng(58 + r)
    n = 150
    Ms = rng.integers(1, 59, size=n)
    Ms[[3, 77, 149]] = 0                                      # empty trajectories: invalid input, not a crash
    so = np.zeros(n + 1, dtype=np.int32)
    so[1:] = np.cumsum(Ms)
    tot = int(so[-1])
    wp = np.cumsum(rng.uniform(-1.0, 1.0, size=(tot + n, 3)), axis=0)
    T = rng.uniform(0.4, 2.0, size=tot)
    bc = rng.uniform(-1.0, 1.0, size=(n, 2, r - 1, 3))
    valid = Ms > 0
    dev = torch.device("cuda", 0)
    d_so = torch.from_numpy(so).to(dev)
    nc = 3 * 2 * r

    def run(offset):
        # buffers shifted by `offset` doubles: offset 1 = 8-byte-aligned-only views
        def shifted(x):
            buf = torch.zeros(x.size + 2, dtype=torch.float64, device=dev)
            v = buf[offset:offset + x.size]
            v.copy_(torch.from_numpy(np.ascontiguousarray(x).reshape(-1)).to(dev))
            return v
        d_wp, d_T, d_bc = shifted(wp), shifted(T), shifted(bc)
        obuf = torch.full((tot * nc + 2,), np.nan, dtype=torch.float64, device=dev)
        out = obuf[offset:offset + tot * nc]
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        gpu_ctx.solve_batch_device(r, n, 0, 58, d_so, d_wp, d_T, d_bc, out, st)
        gpu_ctx.synchronize()
        return out.cpu().numpy(), st.cpu().numpy()

    results = {}
    try:
        for mode in (0, 1, 2, 3):
            gpu_ctx.set_settings(generic_lanes_per_traj=mode)
            for off in (0, 1):
                results[(mode, off)] = run(off)
    finally:
        gpu_ctx.set_settings(generic_lanes_per_traj=0)
    for key, (c, st) in results.items():
        assert np.all(st[valid] == U.UAVQP_SOLVED) and np.all(st[~valid] == U.UAVQP_INVALID_INPUT), key
        assert np.all(np.isfinite(c)), key                      # every coefficient of every valid trajectory was written
    base = results[(1, 0)][0]
    for key, (c, _) in results.items():
        assert np.max(np.abs(c - base) / (1.0 + np.abs(base))) < 1e-9, key
    # against the exact oracle, trajectory by trajectory (the oracle wants M >= 1)
    for k in np.nonzero(valid)[0][::7]:
        s0, s1 = int(so[k]), int(so[k + 1])
        ref, _ = oracle.solve_exact_batch(r, np.array([0, s1 - s0], dtype=np.int32), wp[s0 + k:s1 + k + 1], T[s0:s1], bc[k:k + 1])
        g = base[nc * s0:nc * s1]
        assert np.max(np.abs(g - ref)) < 1e-8 * np.max(np.abs(ref)), k
