"""-m gpu: world > 1 through the C ABI on ONE GPU.  Real RCCL refuses two ranks on one device and the builder has one GPU at a
time, so the rank > 1 branches of the multi-GPU entry points (the grouped ncclSend / ncclRecv all-gather-v of allgather_shards in
csrc/uavqp.hip, zero-sized shards, in-place aliasing, TrajOptimizer::solveSharded with real shard bounds) had never executed
anywhere.  Here they do: tests/cpp/test_multirank_fake_rccl.cpp runs the ranks as THREADS of one process (one uavqp_ctx each on
device 0) against a TEST-ONLY librccl.so.1 (tests/cpp/fake_rccl/fake_rccl.cpp: the eleven symbols libuavqp.so binds with dlopen,
rendezvous through process memory, device-to-device copies) put in front of the real one with LD_LIBRARY_PATH.  The 8-GPU run over
real RCCL / xGMI is the driver's; this test is about rank bookkeeping, not about bandwidth."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_three_and_eight_ranks_through_the_c_abi_with_a_fake_rccl():
    rocm = "/opt/rocm"
    if shutil.which("g++") is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no g++ / HIP headers on this box")
    libdir = os.path.join(ROOT, "uav_motion_planning_amd")
    fake_dir = os.path.join(ROOT, "tests", "cpp", "fake_rccl")
    exe = os.path.join(ROOT, "tests", "cpp", "test_multirank_fake_rccl")
    cp = subprocess.run(["g++", "-std=c++14", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                         "-o", os.path.join(fake_dir, "librccl.so.1"), os.path.join(fake_dir, "fake_rccl.cpp"),
                         "-L", os.path.join(rocm, "lib"), "-lamdhip64", f"-Wl,-rpath,{rocm}/lib", "-lpthread"], capture_output=True, text=True)
    assert cp.returncode == 0, cp.stderr
    cp = subprocess.run(["g++", "-std=c++14", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I", os.path.join(libdir, "cpp"),
                         os.path.join(ROOT, "tests", "cpp", "test_multirank_fake_rccl.cpp"), "-o", exe, "-L", libdir, "-luavqp",
                         "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lpthread", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib"],
                        capture_output=True, text=True)
    assert cp.returncode == 0, cp.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=fake_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout[-3000:] + run.stderr[-2000:]
    assert "mode 2 rank 1" in run.stdout and "mode 0 rank 7" in run.stdout       # eight ranks = the driver's node
