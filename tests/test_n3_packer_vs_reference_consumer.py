"""N3 pin: the quadrotor_msgs/PolynomialTrajectory packer (C ABI uavqp_pack_polynomial_trajectory, its Python binding
adapters.pack_polynomial_trajectory and the C++ header cpp/traj_adapters.h) against the REFERENCE's OWN consumer:
/root/reference/src/planner/traj_server/src/poly_traj_server.cpp compiled whole and unmodified (oracle/_ref/libref_traj_server.so,
recipe in oracle/Makefile; ROS / message / Eigen headers are stand-ins).  The message goes through the reference's trajCallback
(:57-81), the reference's cmdPubCallback (:23-55) publishes position / velocity / acceleration at odometry times, and those must
be the polynomials the solver produced -- evaluated independently by the reference's own PolyTraj on the ORIGINAL
[axis][segment][2r] layout (oracle.ref_polytraj_eval) and by oracle/poly_eval.c.

The libraries are built in this container (where /root/reference is mounted) and travel to the GPU box prebuilt."""
import os
import subprocess

import numpy as np
import pytest

from uav_motion_planning_amd import adapters as A
from uav_motion_planning_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uav_motion_planning_amd")
REF = "/root/reference/src/planner"
EXE = os.path.join(ROOT, "tests", "cpp", "test_traj_adapters")


@pytest.fixture(scope="module")
def refts(oracle):
    oracle.build_ref()
    if not oracle.ref_traj_server_available():
        pytest.skip("oracle/_ref/libref_traj_server.so not built (needs /root/reference)")
    return oracle


def feed_and_compare(oracle, r, coef_traj, times, tid, stamp, tol=1e-12):
    nc, M = 2 * r, len(times)
    msg = A.pack_polynomial_trajectory(coef_traj, times, r, trajectory_id=tid)
    oracle.ref_traj_server_feed(msg["trajectory_id"], msg["num_order"], msg["num_segment"], msg["coef_x"], msg["coef_y"], msg["coef_z"],
                                msg["time"], stamp=stamp)
    total = float(np.sum(times))
    assert abs(oracle.ref_traj_server_total_time() - total) < 1e-12 * max(1.0, total)
    ts = np.concatenate([np.linspace(0.0, total, 37), np.cumsum(times)[:-1] + 1e-5, [total + 0.5]])   # incl. just past knots, past the end
    for t in ts:
        # dyadic stamp: (stamp + t) - stamp == t exactly only if stamp + t is exact; use the server's own subtraction for the reference value
        got, ids = oracle.ref_traj_server_tick(stamp + t)
        assert ids == (tid, nc - 1, M)
        t_srv = max(0.0, (stamp + t) - stamp)
        want = oracle.ref_polytraj_eval(nc, times, coef_traj, t_srv)           # the reference's PolyTraj on the ORIGINAL layout
        assert np.max(np.abs(got - want)) <= tol * max(1.0, np.max(np.abs(want))), (t, got, want)
        want2 = oracle.poly_eval(nc, times, coef_traj, t_srv)
        assert np.max(np.abs(got - want2)) <= 1e-9 * max(1.0, np.max(np.abs(want2)))
    # before the trajectory starts the server clamps t to 0: the start state
    got, _ = oracle.ref_traj_server_tick(stamp - 3.0)
    c = np.asarray(coef_traj).reshape(3, M, nc)
    assert np.allclose(got[0], c[:, 0, 0], rtol=0, atol=1e-15) and np.allclose(got[1], c[:, 0, 1], rtol=0, atol=1e-15)


def test_packed_message_through_the_reference_traj_server(refts):
    oracle = refts
    rng = np.random.default_rng(31)
    assert oracle.ref_traj_server_tick(5.0) is None or True      # (a fresh process has no trajectory; other tests may have fed one)
    for r in (3, 4):
        for M in (1, 2, 5, 8):
            times = rng.uniform(0.3, 2.0, size=M)
            coef = rng.normal(size=3 * M * 2 * r)
            feed_and_compare(oracle, r, coef, times, tid=int(10 * r + M), stamp=64.0)


def test_exact_minimiser_of_the_reference_qp_arrives_intact_at_the_server(refts):
    """The reference's own fixed input (test_qpsolve.cpp:10-17) solved exactly by the oracle, packed, consumed by the reference's server:
    waypoints are hit at the knot times and the published start / end states are the boundary conditions."""
    oracle = refts
    pos, T = np.array([1.0, 2.0, 3.0, 4.0]), np.array([1.0, 1.0, 1.0])
    coef1 = oracle.solve_exact(3, pos, [0.0, 0.0], [0.0, 0.0], T)
    coef = np.concatenate([coef1, 2.0 * coef1, -coef1])        # three axes: scaled copies (the QP is linear in the waypoints)
    feed_and_compare(oracle, 3, coef, T, tid=1, stamp=128.0)
    for k, tk in enumerate([0.0, 1.0, 2.0, 3.0]):
        got, _ = oracle.ref_traj_server_tick(128.0 + tk)
        assert np.allclose(got[0], [pos[k], 2 * pos[k], -pos[k]], atol=1e-12)
    got, _ = oracle.ref_traj_server_tick(128.0 + 3.0)
    assert np.max(np.abs(got[1:])) < 1e-11                      # end velocity and acceleration = 0


def test_cpp_adapters_with_the_reference_server_as_consumer():
    """cpp/traj_adapters.h in a C++ program that has the reference's poly_traj_server.cpp compiled in (tests/cpp/test_traj_adapters.cpp):
    flattenPaths (+ the RRT* empty-path case), boundaryFromOdometry, packPolynomialTrajectory -> fillMessage -> trajCallback ->
    cmdPubCallback, downsampleDensePath."""
    if not os.path.exists(os.path.join(REF, "traj_server", "src", "poly_traj_server.cpp")):
        if not os.path.exists(EXE):
            pytest.skip("/root/reference not mounted and no prebuilt program")
    else:
        import uav_motion_planning_amd as U
        U.build()
        subprocess.check_call(["g++", "-O2", "-std=c++14", f"-I{PKG}/cpp", f"-I{ROOT}/oracle/ref_shim/rosmsgs", f"-I{ROOT}/oracle/ref_shim",
                               f"-I{REF}", f"-I{REF}/traj_utils/include", os.path.join(ROOT, "tests", "cpp", "test_traj_adapters.cpp"),
                               f"-L{PKG}", "-luavqp", f"-Wl,-rpath,{PKG}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "traj_adapters ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_device_solution_packed_and_consumed_by_the_reference_server(gpu_ctx, refts):
    """Config-1 shape (7-segment snap) and a ragged batch solved on the device; every trajectory of a sample goes solver output ->
    packer -> reference trajCallback -> reference cmdPubCallback, and the published positions hit the searcher's waypoints at the
    knot times."""
    oracle = refts
    r = 4
    b = W.ragged_batch(4, 64, r)
    so = b["seg_offsets"]
    coef, st = gpu_ctx.solve_batch_host(r, so, b["waypoints"], b["times"], b["bc"])
    wp = np.asarray(b["waypoints"]).reshape(-1, 3)
    for k in range(0, 64, 5):
        M = int(so[k + 1] - so[k])
        c = coef[24 * so[k]:24 * so[k + 1]]
        T = np.asarray(b["times"][so[k]:so[k + 1]])
        feed_and_compare(oracle, r, c, T, tid=k + 1, stamp=256.0, tol=1e-11)
        knots = np.concatenate([[0.0], np.cumsum(T)])
        for i, tk in enumerate(knots):
            got, _ = oracle.ref_traj_server_tick(256.0 + tk)
            assert np.max(np.abs(got[0] - wp[so[k] + k + i])) < 1e-7 * max(1.0, np.max(np.abs(c)))
