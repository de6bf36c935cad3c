"""-m gpu: bench.py with TWO ranks sharing the one GPU of the box (UAVQP_BENCH_BACKEND=gloo: RCCL refuses two ranks per device, so
torch.distributed runs over gloo and the exchange leg uses its stand-in), launched exactly as the driver launches the multi-GPU
bench (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...).  A rehearsal of the rank
bookkeeping of all three bench configurations -- weak scaling of configs[1], the segment-balanced shards of configs[3] and the
sharded corridor pipeline of configs[4] -- before the driver's 8-GPU node runs them over RCCL; no scaling number is read off it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,port", [(2, 29611), (4, 29612), (5, 29613)])
def test_bench_two_ranks_on_one_gpu(config, port):
    env = dict(os.environ, UAVQP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--config", str(config),
           "--cpu-sample", "0", "--no-traffic", "--no-fp64", "--repeats", "2"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["unit"] == "trajectories/s"
    assert d["scaling"] == ("weak" if config == 2 else "strong")
    b = d["config"]["shard_bounds"]
    assert len(b) == 3 and b[0] == 0 and b[1] > 0 and b[2] > b[1]
    if config == 2:
        assert b == [0, 4096, 8192] and abs(d["value"] - 2 * 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    else:
        n = 32768 if config == 4 else 16384
        assert b[2] == n and abs(b[1] - n / 2) < 0.05 * n          # balanced by segment count, not by trajectory count
        assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if "allgather" in d:
        assert d["allgather"]["own_shard_intact"]


def test_bench_gpus_2_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's N = 1 command with another N): the
    script re-executes itself under torch.distributed.run instead of dying on an assertion (bench.py:343)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(UAVQP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--cpu-sample", "0", "--no-traffic",
           "--no-fp64", "--repeats", "2"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["shard_bounds"] == [0, 4096, 8192] and "other_configs" not in d


@pytest.mark.parametrize("config", [2, 4, 5])
def test_bench_eight_ranks_rehearsal_in_the_drivers_shape(config):
    """the command the driver will run on its 8-GPU node -- `python3 bench.py --gpus 8 --steps 20 --warmup 5`, no
    launcher: the script spawns its eight ranks -- rehearsed with the eight ranks sharing THIS box's one GPU (UAVQP_BENCH_BACKEND=gloo: RCCL
    refuses two ranks per device).  Exactly one JSON line, n_gpus = 8, `value` = the sum over the ranks, the shard bounds those of
    uavqp_shard_bounds_ragged, `allgather` with `value_with_gather` (and the keys of the overlapped leg, null on the stand-in).  Nothing about
    speed is read off it."""
    import numpy as np

    from uav_motion_planning_amd import distributed as D
    from uav_motion_planning_amd import workloads as W
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(UAVQP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"] + (["--config", str(config)] if config != 2 else [])
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"{len(lines)} JSON lines on stdout"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "trajectories/s" and "other_configs" not in d
    b = d["config"]["shard_bounds"]
    assert len(b) == 9 and b[0] == 0 and all(b[i + 1] > b[i] for i in range(8))
    if config == 2:
        assert d["scaling"] == "weak" and b == [4096 * g for g in range(9)]
        assert abs(d["value"] - 8 * 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]           # whole-job aggregate: every rank's 4096 per step
    else:
        n = 32768 if config == 4 else 16384
        full = W.ragged_batch(config, n, 4)
        assert d["scaling"] == "strong" and b == D.shard_bounds_ragged(np.asarray(full["seg_offsets"], dtype=np.int64), 8)
        assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    g = d["allgather"]
    assert g["own_shard_intact"] and g["ms"] > 0 and 0 < g["value_with_gather"] < d["value"]
    assert "value_with_gather_overlapped" in g and "rccl_world" in g       # (null here: the stand-in has no RCCL communicator)
    assert g["bytes_total"] >= 4 * g["bytes_per_rank_out"]       # every rank receives the whole batch


def test_force_dist_world_one_costs_nothing_and_reports_the_rccl_world():
    """VERDICT r5 item 5(b, d): with torch.distributed + RCCL initialised at world size 1 (`--force-dist`) the headline figure stays within
    3 % of the plain N = 1 run on the same box, the exchange goes through the ctx communicator, RCCL itself reports rank 0 of 1
    (allgather.rccl_world, from ncclCommCount), and the overlapped leg (gather of step i while step i + 1 solves) produces a figure."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "20", "--cpu-sample", "0", "--no-traffic", "--no-fp64",
            "--pipelined-streams", "0", "--no-time-modes", "--no-other-configs", "--parity-sample", "0"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("UAVQP_BENCH_BACKEND", None)
    vals = {}
    for tag, extra in (("plain", []), ("dist", ["--force-dist"]), ("plain2", [])):
        p = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        vals[tag] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    plain = max(vals["plain"]["value"], vals["plain2"]["value"])
    assert vals["dist"]["value"] > 0.97 * min(vals["plain"]["value"], vals["plain2"]["value"]) and vals["dist"]["value"] < 1.03 * plain * 1.03
    g = vals["dist"]["allgather"]
    assert g["through"].startswith("uavqp_allgather_coeffs (RCCL, ctx communicator)") and g["rccl_world"] == 1 and g["rccl_rank"] == 0 and g["rccl_world_matches_n_gpus"]
    assert g["value_with_gather_overlapped"] and g["value_with_gather_overlapped"] > 0 and g["overlapped"]["own_shard_intact"]
    # (no speed claim at world 1: the "exchange" of one rank is a 6 us copy, and the overlapped leg pays two events and eager launches per step where the
    #  serial figure adds the gather to a graph-launched step; it pays off when the gather is the longer of the two, i.e. on N > 1 ranks)
    assert g["overlapped"]["steps"] >= 20 and g["overlapped"]["ms_per_step"] > 0
