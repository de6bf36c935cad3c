"""-m gpu: the multi-GPU entry points of the C ABI on ONE GPU (world size 1 -- RCCL wants one GPU per rank, the driver runs
the real N = 2, 4, 8 through bench.py): uavqp_comm_unique_id / uavqp_comm_create on the ctx, uavqp_allgather_coeffs and
_status in place and out of place behind a solve on the ctx stream, uavqp_comm_destroy; and distributed.solve_sharded with
that communicator on device tensors (views in, results written straight into the full output)."""
import os

import numpy as np
import pytest
import torch

import uav_motion_planning_amd as U
from uav_motion_planning_amd import distributed as D
from uav_motion_planning_amd import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_communicator_and_sharded_solve(oracle):
    r, n = 4, 96
    b = W.ragged_batch(4, n, r, m_lo=2, m_hi=12)
    so = np.asarray(b["seg_offsets"])
    dev = torch.device("cuda", 0)
    tb = dict(r=r, seg_offsets=so, waypoints=torch.from_numpy(np.asarray(b["waypoints"]).reshape(-1, 3)).to(dev),
              times=torch.from_numpy(np.asarray(b["times"])).to(dev), bc=torch.from_numpy(np.asarray(b["bc"])).to(dev))
    with U.Context(0) as ctx:
        uid = U.Context.comm_unique_id()
        assert len(uid) == 128
        ctx.comm_create(0, 1, uid)
        with pytest.raises(U.UavqpError):
            ctx.comm_create(0, 1, uid)            # one communicator per ctx

        def solve_local(sh, c_out, st_out):
            d_so = torch.from_numpy(sh["seg_offsets"]).to(dev)
            ctx.solve_batch_device(r, len(sh["seg_offsets"]) - 1, 0, 12, d_so, sh["waypoints"], sh["times"], sh["bc"], c_out, st_out)
        coeff, status, bounds = D.solve_sharded(tb, solve_local, 0, 1, ctx=ctx)
        ctx.synchronize()
        assert bounds == [0, n]
        ref, _ = oracle.solve_exact_batch(r, so, np.asarray(b["waypoints"]).reshape(-1, 3), np.asarray(b["times"]), b["bc"])
        got = coeff.cpu().numpy()
        assert bool((status == U.UAVQP_SOLVED).all())
        assert np.max(np.abs(got - ref)) < 1e-8 * np.max(np.abs(ref))
        # out-of-place gather of a separate local buffer
        full = torch.zeros_like(coeff)
        ctx.allgather_coeffs(coeff.clone(), [coeff.numel()], full)
        st_full = torch.zeros_like(status)
        ctx.allgather_status(status.clone(), [status.numel()], st_full)
        ctx.synchronize()
        assert torch.equal(full, coeff) and torch.equal(st_full, status)
        ctx.comm_destroy()
        ctx.comm_destroy()                        # idempotent
        with pytest.raises(U.UavqpError):
            ctx.allgather_coeffs(coeff, [coeff.numel()], full)   # no communicator any more


def test_cpp_traj_optimizer_sharded_path_from_cpp():
    """The same entry points from C++: traj_optimization::TrajOptimizer::initDistributed / solveSharded (uavqp_comm_create,
    uavqp_shard_bounds_ragged, device solves on views, uavqp_allgather_coeffs / _status) against the facade's own host-pointer
    solve, with and without a corridor -- tests/cpp/test_traj_optimizer_sharded.cpp, compiled here with the HIP runtime API on
    the include path (the facade's sharded entry keeps its buffers on the device)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = "/opt/rocm"
    if shutil.which("g++") is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no g++ / HIP headers on this box")
    exe = os.path.join(root, "tests", "cpp", "test_traj_optimizer_sharded")
    libdir = os.path.join(root, "uav_motion_planning_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-I", os.path.join(rocm, "include"), "-I", os.path.join(libdir, "cpp"),
           os.path.join(root, "tests", "cpp", "test_traj_optimizer_sharded.cpp"), "-o", exe, "-L", libdir, "-luavqp",
           "-L", os.path.join(rocm, "lib"), "-lamdhip64", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib"]
    cp = subprocess.run(cmd, capture_output=True, text=True)
    assert cp.returncode == 0, cp.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "OK" in run.stdout, run.stdout + run.stderr


@pytest.mark.parametrize("config", [2, 4, 5])
def test_bench_exchange_leg_goes_through_the_ctx_communicator(config):
    """The situation on the driver's 8-GPU node, as far as one GPU can rehearse it: bench.py under
    torch.distributed with the nccl backend (= RCCL) AND the library's own communicator (uavqp_comm_create) in the same process --
    two RCCL communicators side by side.  The exchange leg must go through uavqp_allgather_coeffs, not fall back to torch's
    all-gather, and must leave this rank's shard intact.  (World size 1: RCCL refuses two ranks on one device.)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + config), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("UAVQP_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--gpus", "1", "--config", str(config), "--steps", "3", "--warmup", "1",
           "--cpu-sample", "0", "--no-traffic", "--no-fp64", "--pipelined-streams", "0", "--no-time-modes"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    g = d["allgather"]
    assert g["through"].startswith("uavqp_allgather_coeffs (RCCL, ctx communicator)"), g["through"]
    assert g["own_shard_intact"] is True and g["ms"] > 0 and d["n_gpus"] == 1 and d["value"] > 0
    # round 6: RCCL's own view of the communicator, and the overlapped leg (gather of step i beside the solve of step i + 1) on every config
    assert g["rccl_world"] == 1 and g["rccl_rank"] == 0
    assert g["value_with_gather_overlapped"] and g["overlapped"]["own_shard_intact"], g.get("overlapped")
