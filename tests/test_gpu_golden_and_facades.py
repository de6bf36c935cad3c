"""-m gpu: committed exact-rational fixtures through the C ABI, the host-side mirrors of the reference
interface (Python MinimumControl / TrajOptimizer, C++ MinimumControl / TrajOptimizer), device-pointer
entry point with torch buffers and an explicit stream."""
import json
import os
import subprocess

import numpy as np
import pytest

import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "kkt_exact.json")))["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("variant", [0, 1, 4, 8])
def test_golden_fixture_through_c_abi(gpu_ctx, case, variant):
    """Tolerance: 1e-10 relative to max|coef| per axis (float64 direct solve vs exact rationals);
    the north star's budget is 1e-5."""
    r, M = case["r"], case["M"]
    if variant >= 4 and (M < 2 or (r == 4 and M > 12) or M == 11):
        pytest.skip("no specialised instantiation for this M")
    gpu_ctx.set_variant(variant)
    # replicate into a small batch so that partial tiles and both lanes of a pair are exercised
    n = 5
    wp = np.tile(np.array(case["waypoints"])[None], (n, 1, 1))
    T = np.tile(np.array(case["times"])[None], (n, 1))
    bc = np.tile(np.array(case["bc"])[None], (n, 1, 1, 1))
    got, st = gpu_ctx.solve_batch_host(r, None, wp, T, bc, uniform_segments=M)
    gpu_ctx.set_variant(0)
    assert np.all(st == U.UAVQP_SOLVED)
    got = got.reshape(n, 3, 2 * r * M)
    ref = np.array(case["coef"])
    for b in range(n):
        for ax in range(3):
            assert np.max(np.abs(got[b, ax] - ref[ax])) <= 1e-10 * max(1.0, np.max(np.abs(ref[ax])))


def test_python_minimum_control_mirror(oracle):
    """Same call sequence as test_minimum_jerk.cpp:75-78,100-103,125-128,171-172: three axis solves on one object."""
    opt = U.MinimumControl()
    b = W.uniform_batch(2, 1, 6, 3, time_mode="distance")
    wp, T, bc = b["waypoints"][0], b["times"][0], b["bc"][0]
    for ax in range(3):
        assert opt.solve(wp[:, ax], [bc[0, 0, ax], bc[1, 0, ax]], [bc[0, 1, ax], bc[1, 1, ax]], T) is True
        c = opt.getCoef1d()
        ref = oracle.solve_exact(3, wp[:, ax], bc[0, :, ax], bc[1, :, ax], T)
        assert c.shape == (36,) and np.max(np.abs(c - ref)) < 1e-9 * np.max(np.abs(ref))
    prev = opt.getCoef1d()
    assert opt.solve(wp[:, 0], [0, 0], [0, 0], T[:-1]) is False          # size mismatch -> false, result kept
    assert np.array_equal(opt.getCoef1d(), prev)
    assert opt.solve(wp[:, 0], [0, 0], [0, 0], -T) is False               # invalid time allocation -> "solve failed"
    opt.reset()
    assert np.all(opt.getCoef1d() == 0)


@pytest.mark.parametrize("order", [3, 4])
def test_single_axis_latency_path_from_one_to_many_segments(oracle, order):
    """uavqp_solve_axis_host runs over a pinned page mapped into the device (no copy calls): the page grows with the problem, every
    kernel family reads its inputs from host memory there -- 1 segment (no interior knot), the specialised shapes, the lane-pair
    kernel (13, 40 segments) and the one-lane kernel (100 segments) -- and a failed call leaves the previous result in place."""
    opt = U.MinimumControl(order=order)
    rng = np.random.default_rng(order)
    for M in (1, 2, 7, 8, 13, 40, 100, 3):
        pos = np.cumsum(rng.uniform(-1.0, 1.0, size=M + 1))
        T = rng.uniform(0.5, 2.0, size=M)
        bv, ba = rng.uniform(-1, 1, size=2), rng.uniform(-1, 1, size=2)
        assert opt.solve(pos, bv, ba, T) is True
        c = opt.getCoef1d()
        bcs = np.array([bv[0], ba[0]] + ([0.0] if order == 4 else []))
        bce = np.array([bv[1], ba[1]] + ([0.0] if order == 4 else []))
        ref = oracle.solve_exact(order, pos, bcs, bce, T)
        assert c.shape == (2 * order * M,) and np.max(np.abs(c - ref)) < 1e-8 * np.max(np.abs(ref)), M
    prev = opt.getCoef1d()
    assert opt.solve(pos, bv, ba, np.full(3, np.nan)) is False and np.array_equal(opt.getCoef1d(), prev)


def test_python_traj_optimizer_batch_facade(oracle):
    b = W.ragged_batch(4, 40, 4, m_lo=2, m_hi=12)
    so = b["seg_offsets"]
    opt = U.TrajOptimizer(order=4)
    opt.setWaypoints(b["waypoints"], wp_offsets=so + np.arange(so.size))
    opt.setTimeAllocation(b["times"])
    opt.setBoundary(b["bc"])
    assert opt.solve() is True
    ref, _ = oracle.solve_exact_batch(4, so, b["waypoints"], b["times"], b["bc"])
    got = opt.getPolyCoeff()
    assert np.max(np.abs(got - ref)) < 1e-8 * np.max(np.abs(ref))
    k = 7
    per = opt.getPolyCoeff(k)
    assert per.shape == (3, so[k + 1] - so[k], 8)
    assert np.array_equal(per.ravel(), got[24 * so[k]:24 * so[k + 1]])
    # corridor extension through the facade: boxes from the config-3 generator; setCorridor(None, None) restores equalities
    lo, hi = W.corridor_boxes(b, config_index=4)
    opt.setCorridor(lo, hi)
    assert opt.solve() is True and opt.iterations.max() >= 1
    cor = opt.getPolyCoeff()
    wp = np.asarray(b["waypoints"]).reshape(-1, 3)
    moved = 0
    for t in range(40):
        c = cor[24 * so[t]:24 * so[t + 1]].reshape(3, -1, 8)
        for j in range(1, so[t + 1] - so[t]):
            row = so[t] + t + j
            assert np.all(c[:, j, 0] >= lo[row] - 1e-9) and np.all(c[:, j, 0] <= hi[row] + 1e-9)
            moved += int(np.any(np.abs(c[:, j, 0] - wp[row]) > 1e-6))
    assert moved > 20
    opt.setCorridor(None, None)
    assert opt.solve() is True
    assert np.array_equal(opt.getPolyCoeff(), got)


def test_cpp_facades_mirror_of_test_qpsolve():
    """Compiles tests/cpp/test_qpsolve_mirror.cpp (the reference's test_qpsolve.cpp body + expected values)
    against the drop-in header and runs it on the GPU."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_qpsolve_mirror")
    pkg = os.path.join(ROOT, "uav_motion_planning_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++14", f"-I{pkg}/cpp", f"-I{pkg}/cpp/eigen_shim",
                           os.path.join(ROOT, "tests", "cpp", "test_qpsolve_mirror.cpp"), f"{pkg}/cpp/minimum_control.cpp",
                           f"-L{pkg}", "-luavqp", f"-Wl,-rpath,{pkg}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "MinimumControl KAT" in out.stdout and "TrajOptimizer KAT" in out.stdout


def test_device_pointer_entry_with_torch_stream(oracle):
    import torch
    b = W.uniform_batch(2, 300, 8, 4, time_mode="distance")
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    with U.Context(0) as ctx, torch.cuda.stream(s):
        ctx.set_stream(s.cuda_stream)
        d_wp = torch.from_numpy(b["waypoints"]).to(dev)
        d_T = torch.from_numpy(b["times"]).to(dev)
        d_bc = torch.from_numpy(b["bc"]).to(dev)
        d_out = torch.zeros(300 * 192, dtype=torch.float64, device=dev)
        d_st = torch.zeros(300, dtype=torch.int32, device=dev)
        s.synchronize()
        ctx.solve_batch_device(4, 300, 8, 8, None, d_wp, d_T, d_bc, d_out, d_st)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        assert bool((d_st == U.UAVQP_SOLVED).all())
    ref, _ = oracle.solve_exact_batch(4, b["seg_offsets"], b["waypoints"], b["times"], b["bc"])
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


@pytest.mark.parametrize("r,ragged", [(3, False), (4, False), (4, True)])
def test_batched_evaluation_matches_polytraj_restatement(oracle, r, ragged):
    """N1: uavqp_eval_batch_device vs the restated PolyTraj::evaluatePos/Vel/Acc (poly_traj.hpp:74-168) on the
    100 Hz-style grid of poly_traj_server.cpp:33-37, including samples past the end (clamped) and the 1e-4
    segment-switch slack.  Tolerance 1e-12 relative: same float64 polynomial, Horner vs power-vector order."""
    import torch
    n, ns, dt = 37, 173, 0.05
    b = W.ragged_batch(4, n, r, m_lo=1, m_hi=9) if ragged else W.uniform_batch(5, n, 6, r, time_mode="distance")
    so = b["seg_offsets"]
    dev = torch.device("cuda", 0)
    with U.Context(0) as ctx:
        coef, st = ctx.solve_batch_host(r, so, np.asarray(b["waypoints"]).reshape(-1, 3), np.asarray(b["times"]).reshape(-1), b["bc"])
        assert np.all(st == U.UAVQP_SOLVED)
        d_coef = torch.from_numpy(coef).to(dev)
        d_T = torch.from_numpy(np.asarray(b["times"]).reshape(-1).copy()).to(dev)
        d_so = torch.from_numpy(so).to(dev)
        for what in (7, 1, 5):
            K = bin(what).count("1")
            d_out = torch.full((n, ns, K, 3), float("nan"), dtype=torch.float64, device=dev)
            ctx.eval_batch_device(r, n, 0 if ragged else 6, d_so if ragged else None, d_T, d_coef, ns, 0.0, dt, what, d_out)
            ctx.synchronize()
            got = d_out.cpu().numpy()
            T = np.asarray(b["times"]).reshape(-1)
            for k in range(n):
                tk = T[so[k]:so[k + 1]]
                ck = coef[3 * 2 * r * so[k]:3 * 2 * r * so[k + 1]]
                for s in range(0, ns, 7):
                    ref = oracle.poly_eval(2 * r, tk, ck, s * dt, what)
                    assert np.max(np.abs(got[k, s] - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def test_batched_ellipsoid_collision_check_matches_kino_astar_restatement(oracle):
    """N4: uavqp_ellipsoid_check_device vs the restated KinoAstar::isCollisionFree (kino_astar.cpp:721-758) with the
    reference's robot ellipsoid r = 0.4, h = 0.1 (test_kino_astar_searching.launch:56-57).  Flags must agree
    bit-exactly except for samples whose decisive point sits within 1e-9 of the ellipsoid surface."""
    import torch
    r, M, n, ns, dt = 4, 6, 24, 64, 0.08
    robot_r, robot_h = 0.4, 0.1
    b = W.uniform_batch(6, n, M, r, time_mode="distance")
    rng = np.random.default_rng(77)
    # obstacle cloud: random pillars' points near the paths so that a good fraction of samples collide
    wp = b["waypoints"].reshape(-1, 3)
    centres = wp[rng.integers(0, wp.shape[0], size=400)] + rng.normal(scale=0.5, size=(400, 3))
    obs = (centres[:, None, :] + rng.normal(scale=0.08, size=(400, 6, 3))).reshape(-1, 3)
    dev = torch.device("cuda", 0)
    with U.Context(0) as ctx:
        coef, st = ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
        assert np.all(st == U.UAVQP_SOLVED)
        d_coef = torch.from_numpy(coef).to(dev)
        d_T = torch.from_numpy(b["times"].reshape(-1).copy()).to(dev)
        d_obs = torch.from_numpy(obs.copy()).to(dev)
        d_first = torch.zeros(n, dtype=torch.int32, device=dev)
        d_flags = torch.zeros(n * ns, dtype=torch.uint8, device=dev)
        d_ev = torch.zeros(n * ns * 6, dtype=torch.float64, device=dev)
        ctx.ellipsoid_check_device(r, n, M, None, d_T, d_coef, ns, 0.0, dt, d_obs, obs.shape[0], robot_r, robot_h, d_first, d_flags)
        ctx.eval_batch_device(r, n, M, None, d_T, d_coef, ns, 0.0, dt, 5, d_ev)   # pos + acc at the same samples
        ctx.synchronize()
        flags = d_flags.cpu().numpy().reshape(n, ns)
        first = d_first.cpu().numpy()
        ev = d_ev.cpu().numpy().reshape(n, ns, 2, 3)
    n_hit = 0
    for k in range(n):
        exp_first = ns
        for s in range(ns):
            free = oracle.is_collision_free(ev[k, s, 0], ev[k, s, 1], obs, robot_r, robot_h)
            if bool(flags[k, s]) == free:   # disagreement: only tolerated on the surface of the ellipsoid
                free_in = oracle.is_collision_free(ev[k, s, 0], ev[k, s, 1], obs, robot_r * (1 - 1e-9), robot_h * (1 - 1e-9))
                free_out = oracle.is_collision_free(ev[k, s, 0], ev[k, s, 1], obs, robot_r * (1 + 1e-9), robot_h * (1 + 1e-9))
                assert free_in != free_out, (k, s)
            if flags[k, s] and exp_first == ns:
                exp_first = s
            n_hit += int(flags[k, s])
        assert first[k] == exp_first
    assert 0.02 * n * ns < n_hit < 0.9 * n * ns     # the cloud really cuts through some of the paths


def test_misaligned_device_views_take_the_generic_kernel(oracle):
    """Arrays that start at an odd double (8-byte but not 16-byte aligned views) must still solve correctly: the
    specialised kernels need 16-byte alignment for LDS-DMA / dwordx4 traffic, the library has to notice."""
    import torch
    r, M, n = 4, 8, 100
    b = W.uniform_batch(2, n, M, r, time_mode="distance")
    dev = torch.device("cuda", 0)

    def odd(x):
        buf = torch.zeros(x.size + 1, dtype=torch.float64, device=dev)
        buf[1:] = torch.from_numpy(np.ascontiguousarray(x).reshape(-1)).to(dev)
        return buf, buf[1:]
    with U.Context(0) as ctx:
        keep, d_wp = odd(b["waypoints"])
        keep2, d_T = odd(b["times"])
        keep3, d_bc = odd(b["bc"])
        keep4, d_out = odd(np.zeros(n * 192))
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        assert d_wp.data_ptr() % 16 == 8
        ctx.solve_batch_device(r, n, M, M, None, d_wp, d_T, d_bc, d_out, d_st)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        assert bool((d_st == U.UAVQP_SOLVED).all())
    ref, _ = oracle.solve_exact_batch(r, b["seg_offsets"], b["waypoints"], b["times"], b["bc"])
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


@pytest.mark.parametrize("r,mode,ragged", [(3, "reference", False), (4, "distance", False), (4, "reference", True)])
def test_batched_traj_length_and_mean_velocity_vs_oracle(gpu_ctx, oracle, r, mode, ragged):
    """uavqp_traj_length_device = PolyTraj::getTraj + getLength + getMeanVel (poly_traj.hpp:175-207) for a batch, against the
    restated reference loop (oracle/poly_eval.c, itself pinned on the reference header): the sample COUNT exactly -- constant
    1.0 s segments make the total time a multiple of the 0.01 s step, where the accumulated t decides -- length and mean
    velocity to 1e-10 relative (parallel chord sum, t_s = s dt instead of the accumulated t)."""
    import torch
    dev = torch.device("cuda", 0)
    n = 40
    if ragged:
        b = W.ragged_batch(4, n, r, m_lo=1, m_hi=9)
        b["times"] = np.round(np.asarray(b["times"]) * 4) / 4 + 0.25          # multiples of 0.25 s: totals on the 0.01 grid
        uni = 0
    else:
        b = W.uniform_batch(2, n, 5, r, time_mode=mode)
        uni = 5
    so = np.asarray(b["seg_offsets"])
    T = np.asarray(b["times"]).reshape(-1)
    coef, st = gpu_ctx.solve_batch_host(r, so, np.asarray(b["waypoints"]).reshape(-1, 3), T, b["bc"])
    assert np.all(st == U.UAVQP_SOLVED)
    d_so, d_T, d_c = torch.from_numpy(so).to(dev), torch.from_numpy(T).to(dev), torch.from_numpy(coef).to(dev)
    length = torch.zeros(n, dtype=torch.float64, device=dev)
    mean_v = torch.zeros(n, dtype=torch.float64, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    gpu_ctx.traj_length_device(r, n, uni, None if uni else d_so, d_T, d_c, 0.01, length, mean_v, cnt)
    gpu_ctx.synchronize()
    length, mean_v, cnt = length.cpu().numpy(), mean_v.cpu().numpy(), cnt.cpu().numpy()
    for k in range(n):
        ck = coef[3 * 2 * r * so[k]:3 * 2 * r * so[k + 1]]
        le, ve, ne = oracle.traj_length(2 * r, T[so[k]:so[k + 1]], ck)
        assert cnt[k] == ne, (k, cnt[k], ne)
        assert abs(length[k] - le) <= 1e-10 * le and abs(mean_v[k] - ve) <= 1e-10 * ve
