"""The reference's OWN test program, src/planner/test/src/test_qpsolve.cpp, compiled unmodified (from where it lies) against
the drop-in header uav_motion_planning_amd/cpp/traj_optimization/minimum_control.h and linked with libuavqp.so -- the
source-level drop-in claim of INTEGRATION.md, executed.  Stand-ins used: the package's Eigen shim (VectorXd / Vector2d
with the comma initialiser) and a three-function ros stub (tests/cpp/ros_stub).

CPU part (here, /root/reference mounted, no GPU): it compiles, runs, and fails LOUDLY through the reference's own message
-- the product has no CPU fallback.  GPU part (the prebuilt binary travels): it runs to completion without that message."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/planner/test/src/test_qpsolve.cpp"
EXE = os.path.join(ROOT, "tests", "cpp", "ref_test_qpsolve")
PKG = os.path.join(ROOT, "uav_motion_planning_amd")


def _build():
    subprocess.check_call(["g++", "-O2", "-std=c++14", f"-I{PKG}/cpp", f"-I{PKG}/cpp/eigen_shim", f"-I{ROOT}/tests/cpp/ros_stub",
                           SRC, f"{PKG}/cpp/minimum_control.cpp", f"-L{PKG}", "-luavqp", f"-Wl,-rpath,{PKG}",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])


def test_reference_test_program_compiles_unmodified_and_fails_loudly_without_a_gpu():
    if not os.path.exists(SRC):
        pytest.skip("/root/reference not mounted")
    import uav_motion_planning_amd as U
    U.build()
    _build()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    if not has_gpu:
        assert "solver init failed!" in out.stdout            # the reference's own message (minimum_control.cpp:175), no fallback


@pytest.mark.gpu
def test_reference_test_program_runs_on_the_gpu_backend():
    if not os.path.exists(EXE):
        pytest.skip("prebuilt tests/cpp/ref_test_qpsolve did not travel (built where /root/reference is mounted)")
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "failed" not in out.stdout
