#!/usr/bin/env python3
"""bench.py -- trajectories/s of the batched min-snap QP hot path on N MI355X GPUs (one process per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (uavqp_solve_batch_device: assembly + factorisation + solve +
coefficient write-back for all 3 axes) over one batch of synthetic waypoints already resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- 4096 independent 8-segment 7th-order (min-snap) 3-axis
trajectories, synthetic A*-like waypoints (uav_motion_planning_amd/workloads.py, seed 20260925+2).
N>1: every rank solves its own 4096-trajectory shard (weak scaling, no data-path collective in the
timed loop); the RCCL all-gather of the solved coefficient shards is run and timed separately and
reported under "allgather" (DESIGN.md section 7).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU per step")
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--order", type=int, default=4, help="4 = min-snap (7th-order), 3 = min-jerk")
    ap.add_argument("--time-mode", default="distance", choices=["reference", "distance", "wide"])
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 auto)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="trajectories in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent HIP streams the steps are issued on round-robin (0 = auto = 1; 2 pipelines consecutive steps: "
                         "+40 % trajectories/s at the 4096 batch, but each kernel then shares the GPU and its own duration grows)")
    ap.add_argument("--graph", type=int, default=0, metavar="G",
                    help="replay the steps as hipGraphs of G launches per stream (uavqp_capture_*): takes the per-launch host "
                         "cost and part of the inter-kernel gap out (measured with G = 50: 6.11 -> 5.85 us per step on one stream, "
                         "1.06e9 trajectories/s on four); 0 = plain launches (default)")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) and run the barrier / all-reduce / all-gather code even at world size 1 (self-test)")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)  # child run under rocprofv3: kernels only
    ap.add_argument("--data", default="astar", choices=["astar", "uniform"],
                    help="astar: configs[1] generator; uniform: iid waypoints/times (tuning aid)")
    return ap.parse_args()


CPU_BASELINE_SECONDS = 10.0


def cpu_baseline(batch, r, n_sample):
    """OSQP-faithful CPU restatement (oracle/osqp_port.c) timed on this host's cores: one full
    setup+solve+cleanup per axis, exactly the reference's call pattern (test_minimum_jerk.cpp:75,100,125;
    minimum_control.cpp:164-190), reference settings, stdout dumps excluded.  The reference itself cannot
    be built (OSQP / osqp-eigen / Eigen / ROS absent), hence kind = "port"."""
    from oracle import oracle
    oracle.build()
    n = min(n_sample, batch["waypoints"].shape[0])
    M = batch["M"]
    so = batch["seg_offsets"][: n + 1]
    args = (r, so, batch["waypoints"][:n], batch["times"][:n], batch["bc"][:n])
    # bounded sample of about 10 s of single-core work: the batch is passed over repeatedly (same inputs)
    passes, dt1 = 0, 0.0
    while passes == 0 or (dt1 < CPU_BASELINE_SECONDS and passes < 16):
        t0 = time.perf_counter()
        _, st, iters = oracle.osqp_solve_batch(*args, threads=1)
        dt1 += time.perf_counter() - t0
        passes += 1
    dt1 /= passes
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    oracle.osqp_solve_batch(*args, threads=cores)
    dtn = time.perf_counter() - t0
    return {"value": n / dt1, "unit": "trajectories/s", "cores": 1, "kind": "port",
            "sample": f"first {n} trajectories of the same batch (M={M}, r={r}); OSQP-port, reference settings "
                      f"(eps 1e-3, max_iter 1000), 3 x (setup+solve+cleanup) per trajectory; {passes} passes of {dt1:.2f} s on 1 core; "
                      f"median {int(np.median(iters))} ADMM iterations, {int((st == 1).sum())}/{n} reported solved",
            "all_cores": {"value": n / dtn, "cores": cores}}


def measure_traffic(args):
    """HBM bytes per launch of the solve kernel from the PMC counters, as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (TCC has 4 slots: 3 + 2 do not fit), kernel-trace
    only; both report KiB; on gfx950 FETCH_SIZE counts exactly half of a 16-B-per-lane coalesced read stream
    (128-B requests tallied as 64 B) -> doubled; WRITE_SIZE calibrated 1.000 on a known 1.61 GB write
    (tools/ubench/write_patterns).  Returns None when rocprofv3 is unavailable or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    out = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="uavqp_pmc_", dir="/tmp")
            cmd = [prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--inner", "--steps", "10", "--warmup", "2",
                   "--batch", str(args.batch), "--segments", str(args.segments), "--order", str(args.order),
                   "--time-mode", args.time_mode, "--variant", str(args.variant), "--streams", str(args.streams)]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "uavqp::solve" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if not vals:
                return None
            out[ctr] = float(np.mean(vals)) * 1024.0
        return {"bytes": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"], "fetch_bytes_corrected": 2.0 * out["FETCH_SIZE"],
                "write_bytes": out["WRITE_SIZE"]}
    except Exception:
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import uav_motion_planning_amd as U
    from uav_motion_planning_amd import workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # UAVQP_BENCH_BACKEND=gloo is a self-test aid: it lets several ranks share ONE GPU (RCCL refuses that) so that the
        # rank bookkeeping of this script can be exercised on a single-GPU box; collectives then run on CPU tensors.
        backend = os.environ.get("UAVQP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        backend = None
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the uavqp product path has no CPU fallback")
    if backend not in (None, "nccl"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend in (None, "nccl") else torch.device("cpu")

    r, M, B = args.order, args.segments, args.batch
    batch = W.uniform_batch(2, B, M, r, time_mode=args.time_mode, seed=W.SEED0 + 2 + 1000 * rank)
    if args.data == "uniform":
        rng = np.random.default_rng(1)
        batch["waypoints"] = rng.uniform(-2, 2, size=batch["waypoints"].shape)
        batch["times"] = rng.uniform(0.5, 2.0, size=batch["times"].shape)
        batch["bc"] = np.zeros_like(batch["bc"])
    d_wp = torch.from_numpy(batch["waypoints"]).to(dev)
    d_T = torch.from_numpy(batch["times"]).to(dev)
    d_bc = torch.from_numpy(batch["bc"]).to(dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    S = args.streams if args.streams > 0 else 1
    # one ctx + stream + output buffer per pipeline slot: step i runs on slot i % S.  Steps are independent
    # batches, so consecutive steps may overlap on the GPU (kernel boundary of one hides under the next).
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ctxs, d_outs = [], []
    for st_ in streams:
        c = U.Context(local_rank)
        c.set_stream(st_.cuda_stream)
        c.set_variant(args.variant)
        ctxs.append(c)
        d_outs.append(torch.zeros(B * 3 * M * 2 * r, dtype=torch.float64, device=dev))
    d_out = d_outs[0]
    stream = streams[0]

    def launch(k):
        ctxs[k].solve_batch_device(r, B, M, M, None, d_wp, d_T, d_bc, d_outs[k], d_st)

    graphs = None
    if args.graph > 0:
        # one executable hipGraph of G launches per pipeline slot (uavqp_capture_*): a "step" stays one kernel launch over
        # one batch, steps are replayed G at a time; whatever does not fill a whole group is launched eagerly.
        for k in range(S):
            launch(k)          # eager once: sizes the workspaces before capture
        torch.cuda.synchronize()
        graphs = []
        for k in range(S):
            ctxs[k].capture_begin()
            for _ in range(args.graph):
                launch(k)
            graphs.append(ctxs[k].capture_end())
        for k in range(S):
            ctxs[k].graph_launch(graphs[k])   # first replay uploads the graph: keep that out of the timed region
        torch.cuda.synchronize()

    def run_steps(n):
        """Exactly n launches, round-robin over the pipeline slots."""
        done, per_slot = 0, [0] * S
        if graphs is not None:
            k = 0
            while n - done >= args.graph:
                ctxs[k].graph_launch(graphs[k])
                done += args.graph
                per_slot[k] += args.graph
                k = (k + 1) % S
        for i in range(n - done):
            launch(i % S)
            per_slot[i % S] += 1
        return per_slot

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    t0 = time.perf_counter()
    for k in range(S):
        ev0[k].record(streams[k])
    per_slot = run_steps(args.steps)
    for k in range(S):
        ev1[k].record(streams[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0   # this rank's K steps, complete on the device; MAX over ranks below
    fence()                         # closing barrier + synchronize (its own cost is not part of the K steps)
    # average time one launch occupies its stream (kernel + boundary), from the events of the timed region
    region_ms = float(np.mean([ev0[k].elapsed_time(ev1[k]) / per_slot[k] for k in range(S) if per_slot[k] > 0]))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert int((d_st == U.UAVQP_SOLVED).sum().item()) == B, "some trajectories were not solved"

    # per-launch kernel duration: HIP events bracketing each launch on its launch stream (post-pass, same
    # round-robin issue pattern as the timed region)
    n_ev = min(args.steps, 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for i, (a, b) in enumerate(evs):
        a.record(streams[i % S])
        launch(i % S)
        b.record(streams[i % S])
    torch.cuda.synchronize()
    per_launch_ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    kernel_ms = min(per_launch_ms, region_ms)

    gather = None
    if use_dist and not args.no_allgather:
        # RCCL all-gather of the solved coefficient shards over xGMI (equal shards)
        g_src = d_out if coll_dev == dev else d_out.cpu()   # (gloo self-test: CPU tensors)
        full = torch.empty(world * d_out.numel(), dtype=torch.float64, device=coll_dev)
        for _ in range(3):
            dist.all_gather_into_tensor(full, g_src)
        fence()
        g0 = time.perf_counter()
        n_g = 10
        for _ in range(n_g):
            dist.all_gather_into_tensor(full, g_src)
        fence()
        g_ms = (time.perf_counter() - g0) / n_g * 1e3
        ok = bool(torch.equal(full[rank * d_out.numel():(rank + 1) * d_out.numel()], g_src))
        gather = {"ms": g_ms, "bytes_per_rank_out": d_out.numel() * 8, "bytes_total": full.numel() * 8,
                  "own_shard_intact": ok,
                  "value_with_gather": world * B / (dt / args.steps + g_ms * 1e-3)}

    if args.inner:
        return
    out = None
    if rank == 0:
        bytes_per_traj = W.algorithmic_bytes(r, M)
        achieved = B * bytes_per_traj / (kernel_ms * 1e-3) / 1e9
        n_cpu = args.cpu_sample if args.cpu_sample >= 0 else 4096
        cpu = cpu_baseline(batch, r, n_cpu) if (n_cpu > 0 and world == 1) else None
        traffic = measure_traffic(args) if (world == 1 and not args.no_traffic) else None
        out = {
            "metric": "trajectories/sec (8-seg 7th-order min-snap, 3-axis)",
            "value": world * B * args.steps / dt,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: batch of {B} independent {M}-segment order-{2 * r - 1} "
                                   f"(r={r}) 3-axis trajectories per GPU, synthetic A*-like waypoints, "
                                   f"time allocation '{args.time_mode}'",
                       "batch_per_gpu": B, "segments": M, "r": r, "variant": args.variant, "streams": S, "graph": args.graph,
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic["bytes"] if traffic else None,
                         "traffic_detail": traffic, "algorithmic_bytes_per_launch": B * bytes_per_traj,
                         "kernel_ms": kernel_ms, "per_launch_event_ms": per_launch_ms,
                         "stream_ms_per_launch": region_ms,
                         "algorithmic_bytes_per_trajectory": bytes_per_traj},
            "cpu_baseline": cpu,
        }
        if gather:
            out["allgather"] = gather
    if use_dist:
        dist.destroy_process_group()  # RCCL prints its version banner here: keep the JSON line last
    if rank == 0:
        # RCCL (NCCL_DEBUG=VERSION on the GPU boxes) leaves its banner in the C stdio buffer; flush it so that
        # the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
