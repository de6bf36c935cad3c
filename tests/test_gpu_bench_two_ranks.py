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
