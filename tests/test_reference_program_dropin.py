"""The reference's OWN programs that call the optimiser -- src/planner/test/src/test_qpsolve.cpp and
src/planner/test/src/test_minimum_jerk.cpp (the real call pattern: one MinimumControl object, solve() for x, y, z with
getCoef1d() after each, reset() at the end: lines 75-172) -- compiled UNMODIFIED (from where they lie) against the drop-in
header uav_motion_planning_amd/cpp/traj_optimization/minimum_control.h and linked with libuavqp.so: the source-level drop-in
claim of INTEGRATION.md, executed for both callers the reference has.  Stand-ins used: the package's Eigen shim and the
test-only stubs of tests/cpp/ros_stub (ros, the message types, and the two out-of-scope packages test_minimum_jerk.cpp
includes: rrt_star.h "finds" a fixed path, grid_map.h is empty; ros::spin() delivers one odometry message and one goal).

CPU part (here, /root/reference mounted, no GPU): they compile, run, and fail LOUDLY through the reference's own messages
-- the product has no CPU fallback.  GPU part (the prebuilt binaries travel): they run to completion without them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/planner/test/src"
PKG = os.path.join(ROOT, "uav_motion_planning_amd")
PROGRAMS = {"test_qpsolve": os.path.join(ROOT, "tests", "cpp", "ref_test_qpsolve"),
            "test_minimum_jerk": os.path.join(ROOT, "tests", "cpp", "ref_test_minimum_jerk")}


def _build(name):
    subprocess.check_call(["g++", "-O2", "-std=c++14", f"-I{PKG}/cpp", f"-I{PKG}/cpp/eigen_shim", f"-I{ROOT}/tests/cpp/ros_stub",
                           os.path.join(REF, name + ".cpp"), f"{PKG}/cpp/minimum_control.cpp", f"-L{PKG}", "-luavqp", f"-Wl,-rpath,{PKG}",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", PROGRAMS[name]])


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_reference_program_compiles_unmodified_and_fails_loudly_without_a_gpu(name):
    if not os.path.exists(os.path.join(REF, name + ".cpp")):
        pytest.skip("/root/reference not mounted")
    import uav_motion_planning_amd as U
    U.build()
    _build(name)
    out = subprocess.run([PROGRAMS[name]], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    if not _has_gpu():
        # the reference's own messages (minimum_control.cpp:175; test_minimum_jerk.cpp:97,122,147), no fallback
        assert "solver init failed!" in out.stdout
        if name == "test_minimum_jerk":
            assert out.stdout.count("solver init failed!") == 3 and out.stdout.count("optimize faiure!") == 3


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_reference_program_runs_on_the_gpu_backend(name):
    exe = PROGRAMS[name]
    if not os.path.exists(exe):
        pytest.skip(f"prebuilt {exe} did not travel (built where /root/reference is mounted)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "failed" not in out.stdout and "faiure" not in out.stdout
    if name == "test_minimum_jerk":
        # 4 segments of 1.0 s sampled every 0.1 s (floating-point loop of the caller: 10 or 11 samples each) + the end point, x/y/z alike;
        # the sampled curve starts at the odometry position and ends at the goal of the stubs
        m = re.search(r"published (\d+) points, first \(([-\d.e]+), ([-\d.e]+), ([-\d.e]+)\), last \(([-\d.e]+), ([-\d.e]+), ([-\d.e]+)\)", out.stdout)
        assert m, out.stdout
        assert 41 <= int(m.group(1)) <= 45
        first = [float(m.group(k)) for k in (2, 3, 4)]
        last = [float(m.group(k)) for k in (5, 6, 7)]
        assert max(abs(a - b) for a, b in zip(first, [1.0, -0.5, 1.0])) < 1e-9
        assert last == [4.0, 2.5, 1.5]
