#!/usr/bin/env python3
"""bench.py -- trajectories/s of the batched min-snap QP hot path on N MI355X GPUs (one process per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (uavqp_solve_batch_device: assembly + factorisation + solve + coefficient write-back for
all 3 axes) over one batch of synthetic waypoints already resident in HBM.

--config 2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): 4096 independent 8-segment
  7th-order (min-snap) 3-axis trajectories per GPU, synthetic A*-like waypoints (workloads.py, seed 20260925+2).  N > 1: every
  rank solves its own 4096-trajectory shard (weak scaling, no data-path collective in the timed loop).
--config 4 (BASELINE.json configs[3]): ONE batch of 32768 ragged (4-24 segment) min-snap QPs, kino-A*-like inputs, sharded
  over the N ranks by segment count (uavqp_shard_bounds_ragged); strong scaling.
--config 5 (BASELINE.json configs[4]): ONE batch of 16384 ragged min-snap trajectories through the whole corridor pipeline
  (pipeline.py: boxes from a pillar cloud with the SE(3) robot ellipsoid, <= 5 x (corridor solve + time re-allocation), collision
  check), sharded like config 4; a step = one pass of the pipeline, host-sequenced (no graph).
In all modes the RCCL all-gather of the solved coefficient shards (uavqp_allgather_coeffs: the ctx-owned communicator of the
C ABI, device buffers, in place) is run and timed separately and reported under "allgather" (DESIGN.md section 7).

What is timed: the K steps are replayed as ONE hipGraph of K launches (uavqp_capture_*; --graph 0 = K eager launches), step i
reads and writes buffer set i mod S where the S sets together exceed the 256 MiB Infinity Cache (no step finds its inputs
or its output lines in a cache: the roofline figure is an HBM figure at every batch size).  The K-step block is bracketed by
barrier + synchronize on both sides and by HIP events on the launch stream; it is repeated R >= 10 times and the MEDIAN block
is reported (`ms_per_step` = block / K), MAX over ranks.  `roofline.achieved` uses the same clock as `value` (the HIP-event
time of that block), never a different estimator.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20  # buffer sets rotate over more than this


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="2: 4096 x 8-segment snap per GPU (headline); 3: 65536 x 16-segment jerk + corridor boxes per GPU (--rows 2: + K = 2 general rows per segment); 4: 32768 ragged, sharded; 5: 16384 ragged + cloud corridors + time re-allocation pipeline, sharded")
    ap.add_argument("--rows", type=int, default=0, choices=[0, 2], help="config 3: general inequality rows per segment (2 = mid-segment position sample + velocity limit, SURVEY 8-d 'K = 2 mid-segment samples')")
    ap.add_argument("--no-time-modes", action="store_true", help="config 2: skip the `time_modes` sub-record (the same block on the other time allocation of SURVEY 8-d)")
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU per step (config 2; default 4096) / in total (config 4; default 32768)")
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--order", type=int, default=4, help="4 = min-snap (7th-order), 3 = min-jerk")
    ap.add_argument("--time-mode", default="distance", choices=["reference", "distance", "wide"])
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 auto)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="trajectories in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--parity-sample", type=int, default=-1, help="trajectories of the LAST timed step's output checked against the oracle after the timed region -> `parity` (-1 = auto: all of config 2, 256 of the other configs; 0 = skip)")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the K-step block (0 = auto: >= 10, about 0.25 s in total)")
    ap.add_argument("--graph", type=int, default=1, help="1: the K steps are one hipGraph of K launches (default); 0: K eager launches")
    ap.add_argument("--sets", type=int, default=0, help="distinct in/out buffer sets the steps rotate over (0 = auto: > 256 MiB in total)")
    ap.add_argument("--pipelined-streams", type=int, default=4, help="streams of the `pipelined` sub-record (0 = skip it)")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--no-fp64", action="store_true", help="skip the rocprofv3 passes that fill roofline.fp64 and the per-kernel table")
    ap.add_argument("--kernels-only", action="store_true", help="with --no-fp64: still run the rocprofv3 --kernel-trace --stats pass that fills the per-kernel table (durations, no FP64 counters)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the barrier / communicator / all-gather code even at world size 1 (self-test)")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)  # child run under rocprofv3: kernels only
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default invocation (config 2, one GPU): skip the `other_configs` sub-record (short runs of configs 3, 3 + rows, 4, 5 and the config-1 latency)")
    ap.add_argument("--master-port", type=int, default=0, help="--gpus N > 1 without a launcher: rendezvous port of the ranks this script spawns (0 = a free one)")
    ap.add_argument("--data", default="astar", choices=["astar", "uniform"],
                    help="astar: configs[1] generator; uniform: iid waypoints/times (tuning aid)")
    return ap.parse_args()


CPU_BASELINE_SECONDS = 8.0


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def usable_cores():
    """(threads to use, note): the physical cores of the host, capped by the CPU-time quota of this container's cgroup (cpu.max).  The
    GPU boxes of this pool show 256 logical CPUs to a container that may burn 16 CPUs' worth of time per period: 128 pinned threads
    then deliver what 16 cores deliver (measured, tools/cpu_scaling_probe.py: linear up to 16 threads, flat beyond) -- the all-cores
    leg of the CPU baseline runs one thread per core it can actually have."""
    phys = physical_cores()
    quota = None
    try:      # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    if quota is None:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = max(1, int(q / per))
        except Exception:
            pass
    try:      # cpuset / taskset restriction
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = None
    use = phys
    if quota:
        use = min(use, quota)
    if aff:
        use = min(use, aff)
    return use, (f"host: {phys} physical cores, {os.cpu_count()} logical; cgroup cpu quota (v2 cpu.max / v1 cfs_quota_us): {quota if quota else 'none'} CPUs; "
                 f"affinity mask: {aff if aff else 'n/a'} CPUs; threads used: {use}")


def cpu_baseline(batch, r, n_sample):
    """OSQP-faithful CPU restatement (oracle/osqp_port.c) timed on this host's cores: one full setup+solve+cleanup per
    axis, exactly the reference's call pattern (test_minimum_jerk.cpp:75,100,125; minimum_control.cpp:164-190), reference
    settings, stdout dumps excluded.  The reference itself cannot be built (OSQP / osqp-eigen / Eigen / ROS absent), hence
    kind = "port".  value = 1 core (the reference is single-threaded); all_cores = one trajectory per thread on the
    physical cores: warm-up pass, then the median of 7 passes."""
    from oracle import oracle
    oracle.build()
    n = min(n_sample, batch["waypoints"].shape[0])
    M = batch["M"]
    so = batch["seg_offsets"][: n + 1]
    args = (r, so, batch["waypoints"][:n], batch["times"][:n], batch["bc"][:n])
    # bounded sample: the batch is passed over repeatedly (same inputs) for about CPU_BASELINE_SECONDS of single-core work
    passes, dts = 0, []
    while passes == 0 or (sum(dts) < CPU_BASELINE_SECONDS and passes < 16):
        t0 = time.perf_counter()
        port_coef, st, iters = oracle.osqp_solve_batch(*args, threads=1)
        dts.append(time.perf_counter() - t0)
        passes += 1
    dt1 = float(np.median(dts))
    cores, cores_note = usable_cores()
    # all cores: the sample is tiled so that every thread gets >= 512 trajectories (about 0.15 s of work) -- thread creation and pinning
    # (once per pass, oracle/osqp_port.c) must not be what is measured (32 trajectories per thread gave 10 % efficiency)
    rep = max(1, -(-cores * 512 // n))
    n_all = n * rep
    so_all = (np.arange(n_all + 1, dtype=np.int64) * M).astype(np.int32)
    args_all = (r, so_all, np.tile(batch["waypoints"][:n], (rep, 1, 1)), np.tile(batch["times"][:n], (rep, 1)), np.tile(batch["bc"][:n], (rep, 1, 1, 1)))
    oracle.osqp_solve_batch(*args_all, threads=cores)   # warm-up: page faults, allocator arenas
    dtn = []
    for _ in range(5):
        t0 = time.perf_counter()
        oracle.osqp_solve_batch(*args_all, threads=cores)
        dtn.append(time.perf_counter() - t0)
    # second, stronger CPU baseline: the exact KKT solve of the same QPs (oracle/qp_oracle.c, binary128 LU:
    # the checker of the parity tests), one core, a smaller sample
    ne = min(n, 256)
    t0 = time.perf_counter()
    oracle.solve_exact_batch(r, so[: ne + 1], batch["waypoints"][:ne], batch["times"][:ne], batch["bc"][:ne])
    dte = time.perf_counter() - t0
    exact = {"value": ne / dte, "cores": 1, "sample": f"first {ne} trajectories, exact KKT solve in binary128 (the parity oracle), one pass of {dte:.2f} s"}
    return {"value": n / dt1, "unit": "trajectories/s", "cores": 1, "kind": "port", "exact_kkt": exact,
            "_port_coef": (np.asarray(port_coef).reshape(n, -1), np.asarray(st)),   # popped by run(): feeds parity.vs_osqp_port_at_reference_eps
            "sample": f"first {n} trajectories of the same batch (M={M}, r={r}); OSQP-port, reference settings "
                      f"(eps 1e-3, max_iter 1000), 3 x (setup+solve+cleanup) per trajectory; median of {passes} passes of {dt1:.2f} s on 1 core; "
                      f"median {int(np.median(iters))} ADMM iterations, {int((st == 1).sum())}/{n} reported solved",
            "all_cores": {"value": n_all / float(np.median(dtn)), "cores": cores, "passes_s": [round(x, 4) for x in dtn],
                          "sample": f"the same {n} trajectories tiled {rep} x = {n_all} per pass ({n_all // cores} per thread)",
                          "parallel_efficiency": (n_all / float(np.median(dtn))) / (cores * n / dt1),
                          "host": cores_note,
                          "note": "one pinned thread per usable core, contiguous shares of the sample; warm-up pass excluded, median of 5"}}


def cpu_baseline_corridor(batch, r, lo, hi, rows_np, n_sample):
    """Config 3's CPU leg: the OSQP port with the corridor rows (and the general rows) appended, reference settings, on one core -- the
    formulation the reference would hand to OSQP (l < u on the interior-waypoint rows of minimum_control.cpp:118-124).  ADMM at
    eps 1e-3 does not reach the minimiser the device computes; it is the reference's own accuracy."""
    from oracle import oracle
    oracle.build()
    n = min(n_sample, batch["waypoints"].shape[0])
    M = batch["M"]
    so = batch["seg_offsets"][: n + 1]
    kw = dict(corr_lo=lo[:n], corr_hi=hi[:n])
    if rows_np:
        tau, drv, rlo, rhi = rows_np
        kw.update(rows_per_segment=tau.shape[1], row_tau=tau[: n * M], row_deriv=drv[: n * M], row_lo=rlo[: n * M], row_hi=rhi[: n * M])
    d1 = []
    for _ in range(3):      # (median of three passes: the first also warms the allocator)
        t0 = time.perf_counter()
        _, st, iters = oracle.osqp_solve_batch(r, so, batch["waypoints"][:n], batch["times"][:n], batch["bc"][:n], threads=1, **kw)
        d1.append(time.perf_counter() - t0)
        if sum(d1) > 20.0:
            break
    dt1 = float(np.median(d1))
    cores, cores_note = usable_cores()
    rep = max(1, -(-cores * 64 // n))
    n_all = n * rep
    tile = lambda x, k: np.tile(np.asarray(x), (rep,) + (1,) * (np.asarray(x).ndim - 1))
    so_all = (np.arange(n_all + 1, dtype=np.int64) * M).astype(np.int32)
    kw_all = {k: (tile(v, 0) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    a_all = (r, so_all, tile(batch["waypoints"][:n], 0), tile(batch["times"][:n], 0), tile(batch["bc"][:n], 0))
    oracle.osqp_solve_batch(*a_all, threads=cores, **kw_all)
    dtn = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.osqp_solve_batch(*a_all, threads=cores, **kw_all)
        dtn.append(time.perf_counter() - t0)
    return {"value": n / dt1, "unit": "trajectories/s", "cores": 1, "kind": "port",
            "sample": f"first {n} trajectories of the same batch (M={M}, r={r}) with their corridor rows" + (" and general rows" if rows_np else "")
                      + f"; OSQP-port, reference settings (eps 1e-3, max_iter 1000), 3 x (setup+solve+cleanup) per trajectory; median of {len(d1)} passes of {dt1:.2f} s on 1 core; "
                      f"median {int(np.median(iters))} ADMM iterations, {int((st == 1).sum())}/{n} reported solved",
            "all_cores": {"value": n_all / float(np.median(dtn)), "cores": cores, "passes_s": [round(x, 4) for x in dtn],
                          "sample": f"the same {n} trajectories tiled {rep} x", "parallel_efficiency": (n_all / float(np.median(dtn))) / (cores * n / dt1),
                          "host": cores_note}}


PARITY_TOL = 1e-9      # relative to max|coef| per trajectory, four orders inside the north star's 1e-5


def parity_exact(r, so, wp, T, bc, coef, st, sample, tol=PARITY_TOL, port=None):
    """The error clause of BASELINE.json's metric ("max |coeff| err vs OSQP") for the buffers the LAST timed step wrote: the trajectories
    `sample` of the device's coefficient buffer against the exact minimiser of the reference's own QP -- P, A, l, u of
    minimum_control.cpp:5-125 restated in oracle/qp_oracle.c and solved as a KKT system in binary128; what a caller would diff is
    coef_1d_, minimum_control.cpp:186.  OSQP itself is absent from the image: `vs_osqp_port_at_reference_eps` says how
    far the OSQP port at the reference's own settings (eps 1e-3, minimum_control.cpp:160-162) lands from the device's answer.  Runs after
    the timed region, on the host cores, never inside it."""
    from oracle import certificates as C
    t0 = time.perf_counter()
    so = np.asarray(so, dtype=np.int64)
    sample = np.asarray(sample, dtype=np.int64)
    Ms = np.diff(so)[sample]
    sub_so = np.zeros(sample.size + 1, dtype=np.int64)
    sub_so[1:] = np.cumsum(Ms)
    wp, T = np.asarray(wp).reshape(-1, 3), np.asarray(T).reshape(-1)
    if sample.size == so.size - 1:
        sub_wp, sub_T, sub_bc = wp, T, bc
    else:
        sub_wp = np.concatenate([wp[so[k] + k:so[k + 1] + k + 1] for k in sample])
        sub_T = np.concatenate([T[so[k]:so[k + 1]] for k in sample])
        sub_bc = np.asarray(bc)[sample]
    cores, _ = usable_cores()
    ref, st_ref = C.solve_exact_batch_mt(r, sub_so, sub_wp, sub_T, sub_bc, threads=cores)
    rel, ab = np.zeros(sample.size), np.zeros(sample.size)
    w = 3 * 2 * r
    for i, k in enumerate(sample):
        got, want = coef[w * so[k]:w * so[k + 1]], ref[w * sub_so[i]:w * sub_so[i + 1]]
        ab[i] = np.max(np.abs(got - want))
        rel[i] = ab[i] / np.max(np.abs(want))
    rec = {"max_rel_err_vs_exact_kkt": float(rel.max()), "max_abs_coeff_err_vs_exact_kkt": float(ab.max()), "median_rel_err_vs_exact_kkt": float(np.median(rel)),
           "n_checked": int(sample.size), "n_total": int(so.size - 1), "tolerance": tol, "within_tolerance": bool(rel.max() <= tol),
           "all_solved": bool(np.all(np.asarray(st)[sample] == 1)), "oracle_failures": int((st_ref != 0).sum()),
           "relative_to": "max|coef| of the trajectory (three axes)",
           "oracle": "oracle/qp_oracle.c (binary128 KKT of the reference's own P, A, l, u: minimum_control.cpp:5-125)"}
    if port is not None:
        pc, pst = port
        m = min(pc.shape[0], so.size - 1)
        d = np.array([np.max(np.abs(coef[w * so[k]:w * so[k + 1]] - pc[k])) / np.max(np.abs(pc[k])) for k in range(m)])
        rec["vs_osqp_port_at_reference_eps"] = {"median": float(np.median(d)), "worst": float(d.max()), "n": int(m), "port_reported_solved": int((pst[:m] == 1).sum()),
                                                "note": "device coefficients vs oracle/osqp_port.c at the reference's settings (eps_abs = eps_rel = 1e-3, max_iter 1000: "
                                                        "minimum_control.cpp:160-162), relative to max|coef| per trajectory -- the distance ADMM at that eps stops from the "
                                                        "minimiser (SURVEY H1), not a device error"}
    rec["seconds"] = time.perf_counter() - t0
    return rec


def parity_certificate(r, so, wp, T, bc, coef, st, sample, lo, hi, rows_np=None, K_rows=0, tol=(1e-9, 1e-7, 1e-6)):
    """Configs 3 / 3 + rows / 5 (no reference code and no closed-form oracle for inequality rows: SURVEY 8-a'): optimality certificate of the
    LAST timed step's coefficients from the reference-formulation matrices (oracle/certificates.py) on the solved trajectories of
    `sample`, every axis: (primal violation, stationarity residual, complementarity violation), worst over the sample."""
    from oracle import certificates as C
    t0 = time.perf_counter()
    so = np.asarray(so, dtype=np.int64)
    wp, T = np.asarray(wp).reshape(-1, 3), np.asarray(T).reshape(-1)
    lo, hi = np.asarray(lo).reshape(-1, 3), np.asarray(hi).reshape(-1, 3)
    worst, n_ok = np.zeros(3), 0
    for k in sample:
        if st[k] != 1:
            continue
        n_ok += 1
        s0, M = int(so[k]), int(so[k + 1] - so[k])
        rows = slice(s0 + k, s0 + k + M + 1)
        c = coef[3 * 2 * r * s0:3 * 2 * r * (s0 + M)].reshape(3, 2 * r * M)
        for ax in range(3):
            if rows_np is not None:
                tau, drv, rlo, rhi = rows_np
                rr = [(s, tau[s0 + s, j], int(drv[s0 + s, j]), rlo[s0 + s, j, ax], rhi[s0 + s, j, ax]) for s in range(M) for j in range(K_rows)]
                v = C.kkt_certificate_rows(r, M, T[s0:s0 + M], c[ax], wp[rows, ax], bc[k, 0, :, ax], bc[k, 1, :, ax], lo[rows, ax][1:M], hi[rows, ax][1:M], rr)
            else:
                v = C.kkt_certificate(r, M, T[s0:s0 + M], c[ax], wp[rows, ax], bc[k, 0, :, ax], bc[k, 1, :, ax], lo[rows, ax][1:M], hi[rows, ax][1:M])
            worst = np.maximum(worst, v)
    return {"kkt_certificate": {"primal_violation": float(worst[0]), "stationarity_residual": float(worst[1]), "complementarity_violation": float(worst[2])},
            "tolerance": {"primal_violation": tol[0], "stationarity_residual": tol[1], "complementarity_violation": tol[2]},
            "within_tolerance": bool(worst[0] <= tol[0] and worst[1] <= tol[1] and worst[2] <= tol[2]),
            "n_checked": int(n_ok), "n_sampled": int(len(sample)), "n_total": int(so.size - 1), "solved_in_batch": int((np.asarray(st) == 1).sum()),
            "oracle": "oracle/certificates.py: primal feasibility, stationarity P x + A' nu = 0 and multiplier signs on the reference-formulation P, A, l, u "
                      "(minimum_control.cpp:5-125) with the relaxed / added rows appended",
            "seconds": time.perf_counter() - t0}


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
            pm_steps, pm_warm = (6, 1) if args.config == 3 else (40, 2)
            cmd = [prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + _child_cmd(args, ["--steps", str(pm_steps), "--warmup", str(pm_warm)])
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    # one solve kernel per step (configs 2 / 4: the window sort moves 0.5 MB) or every kernel of the step (config 3: prep,
                    # dual prelude, solve, emission)
                    if ("uavqp::" if args.config == 3 else "uavqp::solve") in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if not vals:
                return None
            out[ctr] = (float(np.sum(vals)) / (pm_steps + pm_warm) if args.config == 3 else float(np.mean(vals))) * 1024.0
        return {"bytes": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"], "fetch_bytes_corrected": 2.0 * out["FETCH_SIZE"],
                "write_bytes": out["WRITE_SIZE"]}
    except Exception:
        return None


def _child_cmd(args, extra):
    return [sys.executable, os.path.abspath(__file__), "--inner", "--repeats", "1", "--graph", "0", "--config", str(args.config),
            "--batch", str(args.batch), "--segments", str(args.segments), "--order", str(args.order), "--time-mode", args.time_mode,
            "--variant", str(args.variant), "--rows", str(args.rows), "--pipelined-streams", "0", "--no-allgather"] + extra


def _short(name):
    m = re.search(r"uavqp::([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: FP64 vector (= FP64 matrix on this part)


def measure_fp64(args, steps):
    """FP64 VALU work per launch of every uavqp kernel from the SQ instruction counters (one rocprofv3 --pmc pass, kernel-trace only):
    SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 count wave-instructions; FLOPs = 64 lanes x (ADD + MUL + 2 FMA + TRANS) -- issued lane-slots,
    whatever the exec mask.  SURVEY.md section 8-d / H3: the solvers are bound by dependent FP64 issue, not by HBM; this is the figure
    that says how far from THAT roof they are (78.6 TFLOP/s vector FP64).  Returns {kernel: {launches, wave_insts, flops}} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    ctrs = ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]
    d = tempfile.mkdtemp(prefix="uavqp_f64_", dir="/tmp")
    try:
        cmd = [prof, "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "f64", "--"] + _child_cmd(args, ["--steps", str(steps), "--warmup", "1"])
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "uavqp::" not in k or row["Counter_Name"] not in ctrs:
                    continue
                e = acc.setdefault(_short(k), {"ids": set(), "insts": {c: 0.0 for c in ctrs}})
                e["ids"].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
                e["insts"][row["Counter_Name"]] += float(row["Counter_Value"])
        out = {}
        for k, e in acc.items():
            n = max(1, len(e["ids"]))
            wi = {c.replace("SQ_INSTS_VALU_", ""): v / n for c, v in e["insts"].items()}
            out[k] = {"launches": n, "wave_insts_per_launch": wi,
                      "flops_per_launch": 64.0 * (wi["ADD_F64"] + wi["MUL_F64"] + 2.0 * wi["FMA_F64"] + wi["TRANS_F64"])}
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_kernel_times(args, steps):
    """Per-kernel launch counts and average durations of one bench pass: rocprofv3 --kernel-trace --stats around a child run
    (the summary the judge reads; committed under profiles/ by tools/collect_profiles.sh).  Returns {kernel: {calls, avg_us, total_us}}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    d = tempfile.mkdtemp(prefix="uavqp_kt_", dir="/tmp")
    try:
        cmd = [prof, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "kt", "--"] + _child_cmd(args, ["--steps", str(steps), "--warmup", "1"])
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
        out = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "uavqp::" in row["Name"]:
                    out[_short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "total_us": float(row["TotalDurationNs"]) / 1e3}
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run(args):
    """One bench record (a dict on rank 0, None elsewhere and for --inner child runs)."""
    import torch
    import torch.distributed as dist

    import uav_motion_planning_amd as U
    from uav_motion_planning_amd import distributed as D
    from uav_motion_planning_amd import workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    backend = None
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
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, or without a launcher: the script spawns its ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the uavqp product path has no CPU fallback")
    if backend not in (None, "nccl"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ok = backend in (None, "nccl")

    r = args.order
    K = args.steps
    # ------------------------------------------------------------------ workload: this rank's shard, S distinct buffer sets
    if args.config == 3:
        # BASELINE configs[2]: 65 536 x 16-segment minimum-jerk, corridor boxes of half-width U(0.3, 0.8) m around every interior waypoint
        #  replacing the waypoint equalities of minimum_control.cpp:34-42,118-124; --rows 2 adds the "K = 2 mid-segment
        # samples" as general rows (position sample inside the chord +- 0.25 m, velocity limit 3.5 m/s at mid-segment)
        r = args.order = 3
        args.segments = 16
        args.pipelined_streams = 0
        args.graph = 0     # (the corridor entry sizes its workspace per call; eager launches, as a planner would call it)
    if args.config in (2, 3):
        M = args.segments
        B = args.batch if args.batch > 0 else (4096 if args.config == 2 else 65536)
        batch = W.uniform_batch(args.config, B, M, r, time_mode=args.time_mode, seed=W.SEED0 + args.config + 1000 * rank)
        if args.data == "uniform":
            rng = np.random.default_rng(1)
            batch["waypoints"] = rng.uniform(-2, 2, size=batch["waypoints"].shape)
            batch["times"] = rng.uniform(0.5, 2.0, size=batch["times"].shape)
            batch["bc"] = np.zeros_like(batch["bc"])
        n_local, n_total, uni, mx = B, world * B, M, M
        seg_local = B * M
        bytes_local = B * W.algorithmic_bytes(r, M)
        h_so = None
        shard = batch
        bounds = [g * B for g in range(world + 1)]
        c_counts = [3 * 2 * r * seg_local] * world
        workload = (f"configs[1]: batch of {B} independent {M}-segment order-{2 * r - 1} (r={r}) 3-axis trajectories per GPU, "
                    f"synthetic A*-like waypoints, time allocation '{args.time_mode}'")
        scaling = "weak"
        if args.config == 3:
            c_lo, c_hi = W.corridor_boxes(batch, config_index=3)
            bytes_local += B * 8 * 2 * 3 * (M - 1)                      # corridor rows (3656 B per trajectory in all)
            rows_np = None
            if args.rows:
                K_ = args.rows
                rows_np = W.config3_rows(batch, K_)
                bytes_local += B * M * K_ * (8 + 4 + 2 * 3 * 8)
            workload = (f"configs[2]: batch of {B} independent {M}-segment minimum-jerk (r=3) 3-axis trajectories per GPU, interior waypoints relaxed to "
                        f"corridor boxes (half-width U(0.3, 0.8) m)" + (f" + {args.rows} general rows per segment (mid-segment position sample, velocity limit)" if args.rows else "")
                        + f", time allocation '{args.time_mode}'")
    else:
        n_total = args.batch if args.batch > 0 else (32768 if args.config == 4 else 16384)
        full = W.ragged_batch(args.config, n_total, r)            # identical on every rank (seeded)
        so = np.asarray(full["seg_offsets"], dtype=np.int64)
        bounds = D.shard_bounds_ragged(so, world)                  # uavqp_shard_bounds_ragged: balanced by segment count
        shard = D.local_slice(full, bounds[rank], bounds[rank + 1])
        h_so = shard["seg_offsets"]
        n_local, uni, mx = bounds[rank + 1] - bounds[rank], 0, 24
        seg_local = int(h_so[-1])
        Ms = np.diff(h_so)
        bytes_local = int(sum(W.algorithmic_bytes(r, int(m)) for m in Ms)) + 4 * n_local
        c_counts = [3 * 2 * r * int(so[bounds[g + 1]] - so[bounds[g]]) for g in range(world)]
        workload = (f"configs[3]: ONE batch of {n_total} ragged (4-24 segment) order-{2 * r - 1} 3-axis trajectories, kino-A*-like roll-outs, "
                    f"sharded over {world} GPU(s) by segment count (this rank: {n_local} trajectories, {seg_local} segments)")
        scaling = "strong"
        if args.config == 5:
            # the whole device pipeline per step (uav_motion_planning_amd/pipeline.py): plain solve -> corridor boxes from the pillar
            # cloud (SE(3) robot ellipsoid) -> <= 5 x (warm-started corridor solve + time re-allocation) -> grid collision check
            # (+ repair of what it flags).  Host-sequenced in C++ behind the C ABI (a round counter read back behind an event, no stream stop per round): no graph, no pipelined sub-record.
            obstacles = W.pillar_cloud(5, n_pillars=60, resolution=0.2)       # the same map on every rank (seeded)
            bytes_local += int(sum(8 * 2 * 3 * (int(m) - 1) for m in Ms))     # corridor rows
            args.graph = 0
            args.pipelined_streams = 0
            args.no_traffic = True
            workload = (f"configs[4]: ONE batch of {n_total} ragged (4-24 segment) order-{2 * r - 1} trajectories through the corridor pipeline "
                        f"(boxes from a {obstacles.shape[0]}-point pillar cloud, <= 5 outer rounds of corridor solve + time re-allocation, "
                        f"collision check), sharded over {world} GPU(s) by segment count (this rank: {n_local} trajectories)")
    set_bytes = 8 * (np.asarray(shard["waypoints"]).size + np.asarray(shard["times"]).size + np.asarray(shard["bc"]).size + 3 * 2 * r * seg_local)
    S = args.sets if args.sets > 0 else int(INFINITY_CACHE_BYTES // set_bytes) + 2
    S = max(1, min(S, 4096))
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_so = up(h_so) if h_so is not None else None
    if args.config == 3:
        set_bytes += 8 * 2 * np.asarray(c_lo).size + (sum(x.nbytes for x in rows_np) if rows_np else 0)
        S = args.sets if args.sets > 0 else int(INFINITY_CACHE_BYTES // set_bytes) + 2
    sets = []
    for _ in range(S):
        sets.append(dict(wp=up(np.asarray(shard["waypoints"]).reshape(-1, 3)), T=up(np.asarray(shard["times"]).reshape(-1)),
                         bc=up(shard["bc"]), out=torch.zeros(3 * 2 * r * seg_local, dtype=torch.float64, device=dev)))
        if args.config == 3:
            sets[-1].update(lo=up(np.asarray(c_lo).reshape(-1, 3)), hi=up(np.asarray(c_hi).reshape(-1, 3)))
            if rows_np:
                sets[-1].update(tau=up(rows_np[0]), drv=up(rows_np[1]), rlo=up(rows_np[2]), rhi=up(rows_np[3]))
    d_st = torch.zeros(max(n_local, 1), dtype=torch.int32, device=dev)
    d_it = torch.zeros(max(n_local, 1), dtype=torch.int32, device=dev)

    def make_slot():
        st_ = torch.cuda.Stream(device=dev)
        c = U.Context(local_rank)
        c.set_stream(st_.cuda_stream)
        c.set_variant(args.variant)
        return st_, c

    pipe_state = {}
    if args.config == 5:
        from uav_motion_planning_amd.pipeline import corridor_pipeline_device
        d_obs = up(obstacles)
        T0 = sets[0]["T"].clone()

    def launch(c, i):
        s = sets[i % S]
        if args.config == 5:
            s["T"].copy_(T0)                       # the re-allocation stretches the durations in place
            if "grid" not in pipe_state:           # one grid per map, built once (as a planner would)
                pipe_state["grid"] = c.obstacle_grid_build(d_obs, d_obs.shape[0], 0.4 + 0.1)
            # BASELINE config 5 = boxes + <= 5 outer rounds (+ the check); the repair rounds are the pipeline's own extra.  ONE C-ABI call
            # (uavqp_corridor_pipeline_device: the sequencing is host C++), output buffers reused from step to step
            res = corridor_pipeline_device(c, r, d_so, s["wp"], s["T"], s["bc"], d_obs, mx, grid=pipe_state["grid"], repair_rounds=0,
                                           out=pipe_state.get("out"))
            pipe_state["out"] = {k: res[k] for k in ("coeff", "status", "corr_lo", "corr_hi", "first_hit")}
            s["out"], pipe_state["status"], pipe_state["res"] = res["coeff"], res["status"], res
            return
        if args.config == 3:
            if args.rows:
                c.solve_rows_device(r, n_local, uni, mx, None, s["wp"], s["T"], s["bc"], s["lo"], s["hi"], args.rows, s["tau"], s["drv"], s["rlo"], s["rhi"],
                                    s["out"], d_st, d_it)
            else:
                c.solve_corridor_device(r, n_local, uni, mx, None, s["wp"], s["T"], s["bc"], s["lo"], s["hi"], s["out"], d_st, d_it)
            return
        c.solve_batch_device(r, n_local, uni, mx, d_so, s["wp"], s["T"], s["bc"], s["out"], d_st)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    stream, ctx = make_slot()
    for i in range(max(args.warmup, 1)):       # also sizes the workspaces before any capture
        launch(ctx, i)
    fence()
    graph = None
    if args.graph and K > 0:
        ctx.capture_begin()
        for i in range(K):
            launch(ctx, i)
        graph = ctx.capture_end()
        ctx.graph_launch(graph)                # first replay uploads the graph: keep that out of the timed region
        fence()

    def block():
        if graph is not None:
            ctx.graph_launch(graph)
        else:
            for i in range(K):
                launch(ctx, i)

    # ------------------------------------------------------------------ the timed K-step block, repeated, median
    if args.repeats > 0:
        R = args.repeats
    else:
        t0 = time.perf_counter()
        block()
        torch.cuda.synchronize()
        est = max(time.perf_counter() - t0, 1e-6)
        R = int(min(200, max(10, 0.25 / est)))
        if use_dist:
            tR = torch.tensor([R], dtype=torch.int64, device=dev if rccl_ok else torch.device("cpu"))
            dist.all_reduce(tR, op=dist.ReduceOp.MIN)
            R = int(tR.item())
    wall, evt = [], []
    for _ in range(R):
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        block()
        e1.record(stream)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)   # this rank's K steps, complete on the device
        fence()                                 # closing barrier + synchronize (its own cost is not part of the K steps)
        evt.append(e0.elapsed_time(e1) * 1e-3)
    dt = float(np.median(wall))
    dt_evt = float(np.median(evt))
    if use_dist:
        t = torch.tensor([dt, dt_evt], dtype=torch.float64, device=dev if rccl_ok else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_evt = float(t[0].item()), float(t[1].item())
    if n_local > 0 and args.config == 5:
        assert float((pipe_state["status"] == U.UAVQP_SOLVED).double().mean().item()) > 0.999, "corridor pipeline left trajectories unsolved"
    elif n_local > 0 and args.config == 3 and args.rows:
        # (random rows: a few draws have no feasible point and end as UAVQP_PRIMAL_INFEASIBLE -- or UAVQP_MAX_ITER_REACHED where the certificate stays below the margin)
        assert float((d_st[:n_local] == U.UAVQP_SOLVED).double().mean().item()) > 0.99, "rows solve left trajectories unsolved"
    elif n_local > 0:
        assert int((d_st[:n_local] == U.UAVQP_SOLVED).sum().item()) == n_local, "some trajectories were not solved"
    if args.inner:
        return None

    # ------------------------------------------------------------------ parity capture: the buffers the LAST timed step wrote, copied back now
    # (after the timed region); rank 0 checks them against the oracle when the line is assembled (`parity`)
    cap = None
    if rank == 0 and args.parity_sample != 0 and n_local > 0 and K > 0:
        last = sets[(K - 1) % S]
        used = sorted({i % S for i in range(K)})
        cap = {"coef": last["out"].cpu().numpy(), "st": (pipe_state["status"] if args.config == 5 else d_st[:n_local]).cpu().numpy(),
               "set": (K - 1) % S,
               # every buffer set holds the same inputs: every timed step must have written the same bytes
               "sets_equal": bool(all(torch.equal(sets[j]["out"], last["out"]) for j in used)) if args.config != 5 else None}
        if args.config == 5:
            pr_ = pipe_state["res"]
            cap.update(T=last["T"].cpu().numpy(), lo=pr_["corr_lo"].cpu().numpy(), hi=pr_["corr_hi"].cpu().numpy())

    # ------------------------------------------------------------------ config 2: the same block on the OTHER time allocation of SURVEY 8-d
    # (reference: T_i = 1.0, test_minimum_jerk.cpp:65-71; distance: T_i = max(0.3, |dp| / 2 m/s)) -- same kernel, same bytes, other numbers
    time_modes = None
    if args.config == 2 and not args.no_time_modes and args.time_mode in ("reference", "distance") and K > 0:
        other = "reference" if args.time_mode == "distance" else "distance"
        ob = W.uniform_batch(2, B, M, r, time_mode=other, seed=W.SEED0 + 2 + 1000 * rank)
        saved = [(s_["wp"].clone(), s_["T"].clone(), s_["bc"].clone()) for s_ in sets]
        o_wp, o_T, o_bc = up(np.asarray(ob["waypoints"]).reshape(-1, 3)), up(np.asarray(ob["times"]).reshape(-1)), up(ob["bc"])
        for s_ in sets:
            s_["wp"].copy_(o_wp); s_["T"].copy_(o_T); s_["bc"].copy_(o_bc)   # in place: the captured graph reads the same addresses
        tm_w, tm_e = [], []
        for _ in range(R):
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            block()
            e1.record(stream)
            torch.cuda.synchronize()
            tm_w.append(time.perf_counter() - t0)
            fence()
            tm_e.append(e0.elapsed_time(e1) * 1e-3)
        assert int((d_st[:n_local] == U.UAVQP_SOLVED).sum().item()) == n_local, "some trajectories were not solved (other time allocation)"
        if cap is not None:
            cap["other_mode"] = (other, ob, sets[(K - 1) % S]["out"].cpu().numpy())
        for s_, (a_, b_, c_) in zip(sets, saved):
            s_["wp"].copy_(a_); s_["T"].copy_(b_); s_["bc"].copy_(c_)
        torch.cuda.synchronize()
        tw, te = float(np.median(tm_w)), float(np.median(tm_e))
        if use_dist:
            t = torch.tensor([tw, te], dtype=torch.float64, device=dev if rccl_ok else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tw, te = float(t[0].item()), float(t[1].item())
        mk = lambda w_, e_: {"value": world * n_local * K / w_, "ms_per_step": w_ / K * 1e3, "roofline_frac": bytes_local / (e_ / K) / 1e9 / HBM_PEAK_GBS}
        time_modes = {args.time_mode: mk(dt, dt_evt), other: mk(tw, te),
                      "note": "the same K-step block, same buffers and graph, on both time allocations of SURVEY.md 8-d (reference: T_i = 1.0 s, "
                              "test_minimum_jerk.cpp:65-71; distance: T_i = max(0.3, |dp| / 2 m/s)); `value` of the line is the first"}

    # ------------------------------------------------------------------ pipelined sub-record: the same K steps dealt to P streams
    pipelined = None
    if args.pipelined_streams > 1 and K > 0:
        P = args.pipelined_streams
        KP = max(K, 200)   # its own step count: at the driver's 20 steps the overlap of four streams does not reach steady state
        slots = [(stream, ctx)] + [make_slot() for _ in range(P - 1)]
        for k, (_, c) in enumerate(slots):
            launch(c, k)
        torch.cuda.synchronize()
        graphs = []
        for k, (_, c) in enumerate(slots):
            c.capture_begin()
            for i in range(k, KP, P):
                launch(c, i)
            graphs.append(c.capture_end())
            c.graph_launch(graphs[-1])
        torch.cuda.synchronize()
        pw = []
        for _ in range(R):
            fence()
            t0 = time.perf_counter()
            for (_, c), g_ in zip(slots, graphs):
                c.graph_launch(g_)
            torch.cuda.synchronize()
            pw.append(time.perf_counter() - t0)
            fence()
        pdt = float(np.median(pw))
        if use_dist:
            t = torch.tensor([pdt], dtype=torch.float64, device=dev if rccl_ok else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pdt = float(t.item())
        pipelined = {"streams": P, "steps": KP, "value": n_total * KP / pdt if args.config == 4 else world * n_local * KP / pdt, "ms_per_step": pdt / KP * 1e3,
                     "note": f"{KP} steps (max(--steps, 200): independent of the timed block) dealt round-robin to {P} streams, one hipGraph each: consecutive steps overlap on the GPU "
                             "(each kernel then shares the machine; per-kernel duration is not comparable with the single-stream figure)"}
        for (_, c), g_ in zip(slots, graphs):
            c.graph_destroy(g_)

    # ------------------------------------------------------------------ the exchange step: all-gather of the coefficient shards
    gather = None
    if use_dist and not args.no_allgather:
        tot = sum(c_counts)
        full_out = torch.zeros(tot, dtype=torch.float64, device=dev)
        off = sum(c_counts[:rank])
        mine = full_out[off:off + c_counts[rank]]
        mine.copy_(sets[0]["out"])
        gctx = None
        comm_err = None
        if rccl_ok:
            # ctx-owned RCCL communicator (uavqp_comm_create); torch.distributed ships the unique id.  The exchange leg must never
            # take the headline line down with it: a rank that cannot build the communicator says so, the ranks agree (MIN), and
            # the leg falls back to torch.distributed's own all-gather on the same device buffers.
            try:
                D.create_comm(ctx)
                flag = 1
            except Exception as e:  # noqa: BLE001
                comm_err, flag = repr(e), 0
            tf = torch.tensor([flag], dtype=torch.int32, device=dev)
            dist.all_reduce(tf, op=dist.ReduceOp.MIN)
            if int(tf.item()) == 0:
                if flag:
                    ctx.comm_destroy()
                comm_err = comm_err or "another rank could not create the communicator"
        if rccl_ok and comm_err is None:
            gctx = ctx
            do = lambda: D.allgather_shards(mine, c_counts, full_out, ctx)
        elif rccl_ok:
            do = lambda: D.allgather_shards(mine, c_counts, full_out, None)
        else:
            cpu_full, cpu_mine = full_out.cpu(), mine.cpu()
            do = lambda: D.allgather_shards(cpu_mine, c_counts, cpu_full, None)
        for _ in range(3):
            do()
        gt = []
        for _ in range(10):
            fence()
            g0 = time.perf_counter()
            do()
            if gctx is not None:
                gctx.synchronize()
            torch.cuda.synchronize()
            gt.append(time.perf_counter() - g0)
        fence()
        g_s = float(np.median(gt))
        t = torch.tensor([g_s], dtype=torch.float64, device=dev if rccl_ok else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g_s = float(t.item())
        got = full_out if rccl_ok else cpu_full.to(dev)
        ok = bool(torch.equal(got[off:off + c_counts[rank]], sets[0]["out"]))
        n_step = n_total if args.config != 2 else world * n_local
        gather = {"ms": g_s * 1e3, "bytes_per_rank_out": c_counts[rank] * 8, "bytes_total": tot * 8, "own_shard_intact": ok,
                  "through": ("uavqp_allgather_coeffs (RCCL, ctx communicator)" if gctx is not None else
                              (f"torch.distributed all-gather (uavqp_comm_create failed: {comm_err})" if rccl_ok else f"torch.distributed/{backend} stand-in")),
                  "value_with_gather": n_step / (dt / K + g_s),
                  "value_with_gather_is": "serial: one step = solve, then the exchange behind it on the same stream (solve time per step + median gather time)"}
        if gctx is not None:
            # what RCCL itself says the communicator is (ncclCommUserRank / ncclCommCount through uavqp_comm_info): the exchange ran on N ranks
            rk_, wd_ = gctx.comm_info()
            gather["rccl_rank"], gather["rccl_world"] = rk_, wd_
            gather["rccl_world_matches_n_gpus"] = bool(wd_ == world)
            # ---- overlapped: the gather of step i on the communicator's stream while step i + 1 solves on a second stream.  At 5 us per
            # solve the exchange (every rank receives the whole batch) is the step: serial, the N-rank value is the gather's; overlapped it
            # is max(solve, gather) per step -- the only way "near-linear scaling" can survive the exchange
            try:
                setup_err = None
                try:
                    s2, c2 = make_slot()                  # the solves of this leg: their own ctx and stream; the gathers stay on gctx's stream
                    launch(c2, 0)
                    torch.cuda.synchronize()
                    KO = max(K, 20)
                    fulls = [torch.zeros(tot, dtype=torch.float64, device=dev) for _ in range(2)]
                except Exception as e:  # noqa: BLE001
                    setup_err = repr(e)
                # every rank runs the leg or none does: a rank that dropped out alone would leave the others waiting in a collective
                tf = torch.tensor([0 if setup_err else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(tf, op=dist.ReduceOp.MIN)
                if int(tf.item()) == 0:
                    raise RuntimeError(setup_err or "another rank could not set the overlapped leg up")
                ow = []
                for _ in range(5):
                    fence()
                    o0 = time.perf_counter()
                    done_g = {}
                    for i in range(KO):
                        if i - S in done_g:
                            s2.wait_event(done_g.pop(i - S))      # the gather that still reads this step's output buffer has finished
                        launch(c2, i)
                        ev = torch.cuda.Event()
                        ev.record(s2)
                        stream.wait_event(ev)
                        D.allgather_shards(sets[i % S]["out"], c_counts, fulls[i & 1], gctx)
                        eg = torch.cuda.Event()
                        eg.record(stream)
                        done_g[i] = eg
                    gctx.synchronize()
                    torch.cuda.synchronize()
                    ow.append(time.perf_counter() - o0)
                    fence()
                o_s = float(np.median(ow))
                t = torch.tensor([o_s], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                o_s = float(t.item())
                gather["value_with_gather_overlapped"] = n_step * KO / o_s
                gather["overlapped"] = {"steps": KO, "ms_per_step": o_s / KO * 1e3, "own_shard_intact": bool(torch.equal(fulls[(KO - 1) & 1][off:off + c_counts[rank]], sets[(KO - 1) % S]["out"])),
                                        "note": "eager launches: step i + 1 solves on a second ctx / stream while uavqp_allgather_coeffs of step i runs on the communicator's stream "
                                                "(an event per step orders gather i behind solve i; a solve waits for the gather that still reads its output buffer); median of 5"}
            except Exception as e:  # noqa: BLE001 (the extra leg must not take the line down)
                gather["value_with_gather_overlapped"] = None
                gather["overlapped"] = {"error": repr(e)}
            gctx.comm_destroy()
        else:
            gather["rccl_world"] = None
            gather["value_with_gather_overlapped"] = None

    out = None
    if rank == 0:
        n_step = n_total if args.config != 2 else world * n_local
        per_launch_s = dt_evt / K
        achieved = bytes_local / per_launch_s / 1e9
        n_cpu = args.cpu_sample if args.cpu_sample >= 0 else 4096
        cpu = cpu_baseline(batch, r, n_cpu) if (n_cpu > 0 and world == 1 and args.config == 2) else None
        if n_cpu > 0 and world == 1 and args.config == 3:
            cpu = cpu_baseline_corridor(batch, r, c_lo, c_hi, rows_np, min(n_cpu, 1024))
        port_coef = cpu.pop("_port_coef", None) if cpu else None
        parity = None
        if cap is not None:
            # the error clause of BASELINE.json's metric, for the buffers of the timed run (after the timed region; oracle = checker only)
            so_p = np.asarray(shard["seg_offsets"], dtype=np.int64)
            n_par = n_local if (args.config == 2 and args.parity_sample < 0) else min(n_local, 256 if args.parity_sample < 0 else args.parity_sample)
            if n_par >= n_local:
                sample = np.arange(n_local)
            else:
                sample = np.sort(np.random.default_rng(args.config).choice(n_local, size=n_par, replace=False))
                sample[0], sample[-1] = 0, n_local - 1      # both ends of the batch
                sample = np.unique(sample)
            h_wp, h_T, h_bc = np.asarray(shard["waypoints"]).reshape(-1, 3), np.asarray(shard["times"]).reshape(-1), np.asarray(shard["bc"]).reshape(n_local, 2, r - 1, 3)
            if args.config in (2, 4):
                parity = parity_exact(r, so_p, h_wp, h_T, h_bc, cap["coef"], cap["st"], sample, tol=PARITY_TOL if args.config == 2 else 1e-8, port=port_coef)
                if "other_mode" in cap:
                    om, ob_, oc_ = cap["other_mode"]
                    po = parity_exact(r, so_p, np.asarray(ob_["waypoints"]).reshape(-1, 3), np.asarray(ob_["times"]).reshape(-1), np.asarray(ob_["bc"]).reshape(n_local, 2, r - 1, 3), oc_, cap["st"], sample)
                    parity["other_time_allocation"] = {"time_mode": om, "max_rel_err_vs_exact_kkt": po["max_rel_err_vs_exact_kkt"], "n_checked": po["n_checked"],
                                                       "within_tolerance": po["within_tolerance"]}
            elif args.config == 3:
                parity = parity_certificate(r, so_p, h_wp, h_T, h_bc, cap["coef"], cap["st"], sample, c_lo, c_hi, rows_np, args.rows)
            else:
                # durations span 0.3 s .. 10+ s after re-allocation: raw KKT conditioning of SURVEY App. A, hence the looser bounds
                parity = parity_certificate(r, so_p, h_wp, cap["T"], h_bc, cap["coef"], cap["st"], sample, cap["lo"], cap["hi"], tol=(1e-8, 1e-6, 1e-5))
            parity["checked"] = (f"output of the LAST timed step (buffer set {cap['set']} of {S}), copied back after the timed region; "
                                 + ("every trajectory" if n_par >= n_local else f"{sample.size} drawn trajectories (both ends of the batch included)"))
            parity["buffer_sets_bitwise_equal"] = cap["sets_equal"]
            if world > 1:
                parity["checked"] += f"; rank 0's shard of {world}"
        traffic = measure_traffic(args) if (world == 1 and not args.no_traffic) else None
        # FP64 roof ("report FP64 FLOP/s next to GB/s") and per-kernel view, from counters / traces of child runs
        fp64 = kernels = None
        if world == 1 and (not args.no_fp64 or args.kernels_only):
            prof_steps = 2 if args.config == 5 else (6 if args.config == 3 else 40)
            trace_steps = 2 if args.config == 5 else ((4 if args.rows else 20) if args.config == 3 else 1000)   # the trace is cheap: enough launches that the average is the steady state
            f64 = measure_fp64(args, prof_steps) if not args.no_fp64 else None
            kt = measure_kernel_times(args, trace_steps)
            if kt:
                kernels = []
                tot_us = sum(v["total_us"] for v in kt.values())
                for name, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_us"]):
                    k_ = {"kernel": name, "launches_per_step": v["calls"] / (trace_steps + 1), "avg_us": v["avg_us"],
                          "share_of_gpu_time": v["total_us"] / tot_us if tot_us else None}
                    if f64:
                        fl = f64.get(name, {}).get("flops_per_launch", 0.0)
                        k_.update({"fp64_flops_per_launch": fl, "fp64_tflops": fl / (v["avg_us"] * 1e-6) / 1e12 if v["avg_us"] > 0 else None,
                                   "fp64_frac_of_vector_peak": fl / (v["avg_us"] * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS if v["avg_us"] > 0 else None})
                    kernels.append(k_)
            if f64 and kt:
                dom = kernels[0]
                step_flops = sum(k_["fp64_flops_per_launch"] * k_["launches_per_step"] for k_ in kernels)
                # ONE clock per record: `achieved` / `frac` are the step's FP64 work over the SAME HIP-event time per step that
                # roofline.achieved uses (this process, this box); the traced figure of the dominant kernel alone (child run under
                # rocprofv3 on the same box, eager launches) is kept next to it, labelled
                fp64 = {"kernel": dom["kernel"], "flops_per_step": step_flops, "achieved": step_flops / per_launch_s / 1e12,
                        "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_flops / per_launch_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                        "clock": "HIP events around the timed K-step block / K -- the clock of roofline.achieved",
                        "dominant_kernel_traced": {"flops_per_launch": dom["fp64_flops_per_launch"], "avg_us": dom["avg_us"], "tflops": dom["fp64_tflops"],
                                                   "frac": dom["fp64_frac_of_vector_peak"],
                                                   "clock": "rocprofv3 --kernel-trace average of the child run on this box (eager launches)"},
                        "wave_insts_per_launch": f64.get(dom["kernel"], {}).get("wave_insts_per_launch"),
                        "note": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 wave-instructions x 64 lanes (FMA = 2 FLOP) of every uavqp kernel of a step; "
                                "issued lane-slots, idle lanes included"}
        host_e2e = None
        if world == 1 and args.config == 2 and n_cpu > 0:
            # the same batch from HOST pointers (uavqp_solve_batch_host: staging copies over PCIe, solve, copy back, synchronise):
            # what a caller without device-resident data sees.  Reported next to `value`, never as it.
            hw, hT, hbc = np.asarray(batch["waypoints"]), np.asarray(batch["times"]), np.asarray(batch["bc"])
            for _ in range(3):
                ctx.solve_batch_host(r, None, hw, hT, hbc, uniform_segments=uni)
            ht = []
            for _ in range(15):
                h0 = time.perf_counter()
                ctx.solve_batch_host(r, None, hw, hT, hbc, uniform_segments=uni)
                ht.append(time.perf_counter() - h0)
            host_e2e = {"value": n_local / float(np.median(ht)), "unit": "trajectories/s", "ms_per_call": float(np.median(ht)) * 1e3,
                        "bytes_over_the_host_link": int(bytes_local), "note": "pageable host buffers in and out, one call per batch, median of 15"}
        out = {
            "metric": {2: "trajectories/sec (8-seg 7th-order min-snap, 3-axis)",
                       3: "trajectories/sec (65536 x 16-seg min-jerk with corridor boxes" + (f" + {args.rows} general rows per segment" if args.rows else "") + ", 3-axis)", 4: "trajectories/sec (32768 ragged 4-24-seg min-snap, 3-axis, sharded)",
                       5: "trajectories/sec (16384 ragged min-snap through the SE(3)-corridor + time re-allocation pipeline, sharded)"}[args.config],
            "value": n_step * K / dt,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": n_local, "segments": args.segments if args.config in (2, 3) else "4-24", "r": r,
                       "variant": args.variant, "graph": bool(graph is not None), "buffer_sets": S, "buffer_set_bytes": int(set_bytes),
                       "repeats": R, "parallelism": f"shard{world}", "shard_bounds": bounds if world <= 16 else None},
            "timing": {"block_wall_ms_median": dt * 1e3, "block_event_ms_median": dt_evt * 1e3,
                       "block_wall_ms_p10_p90": [float(np.percentile(wall, 10)) * 1e3, float(np.percentile(wall, 90)) * 1e3],
                       "block_wall_ms_min_max": [float(np.min(wall)) * 1e3, float(np.max(wall)) * 1e3], "repeats": R,
                       "note": "K-step block between barrier + synchronize, repeated; value and ms_per_step from the median wall time (MAX over ranks)"},
            # config 5 is a pipeline of ~20 launches with host synchronisation between rounds: a pipeline-wide HBM fraction says nothing
            #  -- achieved / frac are null there, the per-kernel table (`kernels`) is the content
            "roofline": {"bound": "hbm", "achieved": achieved if args.config != 5 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if args.config != 5 else None, "traffic": traffic["bytes"] if traffic else None,
                         "traffic_detail": traffic, "algorithmic_bytes_per_launch": int(bytes_local),
                         "kernel_ms": per_launch_s * 1e3,
                         "kernel_ms_is": ("HIP events on the launch stream around the timed K-step block / K (same clock as value; includes the inter-kernel gap)"
                                          if args.config not in (3, 5) else "one corridor solve = prep + dual prelude + block solve + emission kernels (HIP events around the K-step block / K)" if args.config == 3 else "one pass of the WHOLE pipeline (about 20 launches, host-synchronised between outer rounds), not one kernel"),
                         "working_set_bytes": int(S * set_bytes),
                         "algorithmic_bytes_per_trajectory": bytes_local / max(n_local, 1),
                         "fp64": fp64},
            "cpu_baseline": cpu,
            "parity": parity,
        }
        if kernels:
            out["kernels"] = kernels   # per-kernel launches, durations and FP64 rates of one step (config 5: the pipeline's own kernels)
        if host_e2e:
            out["host_pointers"] = host_e2e
        if time_modes:
            out["time_modes"] = time_modes
        if args.config == 3:
            out["corridor"] = {"iterations_mean": float(d_it[:n_local].double().mean().item()), "iterations_max": int(d_it[:n_local].max().item()),
                               "solved": int((d_st[:n_local] == U.UAVQP_SOLVED).sum().item()), "rows_per_segment": args.rows,
                               "note": "block solves per (trajectory, axis) problem after the position-space dual prelude (uavqp_settings.corridor_initial_guess = 2)"}
        if args.config == 5 and "res" in pipe_state:
            pr = pipe_state["res"]
            out["pipeline"] = {"rounds": int(pr["rounds"]), "still_stretching": int(pr["still_stretching"]), "all_solved": bool(pr["all_solved"]),
                               "colliding_before_repair": int(pr["colliding_before_repair"]), "repairs": int(pr["repairs"])}
        if pipelined:
            out["pipelined"] = pipelined
        if gather:
            out["allgather"] = gather
    if graph is not None:
        ctx.graph_destroy(graph)
    if use_dist:
        dist.destroy_process_group()  # RCCL prints its version banner here: keep the JSON line last
    return out


OTHER_CONFIGS = (
    # (key, config, rows, steps, warmup): short runs -- what the default invocation appends so that every BASELINE.json config has a
    # figure in the driver's own record; the headline `value` / `config` stay configs[1]
    ("config3_corridor", 3, 0, 10, 3),
    ("config3_rows2", 3, 2, 4, 2),
    ("config4_ragged", 4, 0, 50, 5),
    ("config5_pipeline", 5, 0, 4, 2),
)


def config1_latency():
    """BASELINE configs[0] (plumbing): ONE 8-waypoint / 7-segment 3-axis min-snap trajectory from host pointers through the drop-in
    facades -- MinimumControl.solve per axis (the reference's call, test_minimum_jerk.cpp:75,100,125) and TrajOptimizer.solve (3 axes)."""
    import uav_motion_planning_amd as U
    from uav_motion_planning_amd import workloads as W
    b = W.uniform_batch(1, 1, 7, 4, time_mode="reference")
    wp, T, bc = b["waypoints"][0], b["times"][0], b["bc"][0]
    opt = U.MinimumControl(order=4)
    ok = True
    for _ in range(20):
        opt.solve(wp[:, 0], [bc[0, 0, 0], 0.0], [0.0, 0.0], T)
    N = 300
    t0 = time.perf_counter()
    for i in range(N):
        ok = opt.solve(wp[:, i % 3], [bc[0, 0, i % 3], 0.0], [0.0, 0.0], T) and ok
    one = (time.perf_counter() - t0) / N
    to = U.TrajOptimizer(order=4)
    to.setWaypoints(wp, n_waypoints=8); to.setTimeAllocation(T); to.setBoundary(bc.reshape(1, 2, 3, 3))
    for _ in range(20):
        to.solve()
    t0 = time.perf_counter()
    for i in range(N):
        ok = to.solve() and ok
    three = (time.perf_counter() - t0) / N
    # the same three axis calls straight through the C ABI (what a C++ caller of MinimumControl::solve pays: uavqp_solve_axis_host with
    # prepared pointers; a ctypes call adds about a microsecond) -- tools/ubench/axis_latency.cpp is the version without Python
    import ctypes
    lib, ctx = U.lib(), opt._ctx
    pos = [np.ascontiguousarray(wp[:, ax]) for ax in range(3)]
    tv, bv, ba, bj = np.ascontiguousarray(T, dtype=np.float64), np.array([0.0, 0.0]), np.zeros(2), np.zeros(2)
    coef, st = np.zeros(8 * 7), ctypes.c_int32(0)
    fn = lib.uavqp_solve_axis_host            # (argument types as _lib.py declares them: raw addresses)
    cargs = [(ctx._h, 4, 7, pos[ax].ctypes.data, bv.ctypes.data, ba.ctypes.data, bj.ctypes.data, tv.ctypes.data, coef.ctypes.data, ctypes.byref(st))
             for ax in range(3)]
    for _ in range(20):
        fn(*cargs[0])
    t0 = time.perf_counter()
    for i in range(N):
        for ax in range(3):
            ok = (fn(*cargs[ax]) == 0 and st.value == U.UAVQP_SOLVED) and ok
    c_three = (time.perf_counter() - t0) / N
    return {"workload": "configs[0]: single 8-waypoint / 7-segment 3-axis min-snap QP from host pointers (plumbing)", "all_solved": bool(ok),
            "minimum_control_solve_us_per_axis_call": one * 1e6, "three_axis_calls_us": 3 * one * 1e6,
            "traj_optimizer_solve_us_three_axes": three * 1e6, "c_abi_three_axis_calls_us": c_three * 1e6, "value": 1.0 / three, "unit": "trajectories/s",
            "note": "wall time per synchronous call through one pinned, device-mapped page (no copies; completion = a word the device writes "
                    "into that page, polled); mean of 300 calls after 20 warm-up calls; the first three figures include the numpy facades"}


def other_configs(args):
    """Short runs of the other BASELINE configs in this process, one trimmed record each (same code path as `bench.py --config X`);
    a config that fails reports its error instead of taking the headline line down."""
    import copy
    out = {}
    t_all = time.perf_counter()
    for key, cfg, rows, steps, warm in OTHER_CONFIGS:
        a = copy.copy(args)
        a.config, a.rows, a.steps, a.warmup = cfg, rows, steps, warm
        a.batch, a.sets, a.repeats, a.cpu_sample = 0, 0, 0, 0
        a.order, a.segments, a.time_mode, a.data, a.variant, a.graph = 4, 8, "distance", "astar", 0, 1
        a.no_traffic, a.no_fp64, a.kernels_only = True, True, True
        a.pipelined_streams, a.no_allgather, a.force_dist, a.inner = 0, True, False, False
        t0 = time.perf_counter()
        try:
            rec = run(a)
            rl = rec["roofline"]
            out[key] = {"command": f"bench.py --config {cfg}" + (f" --rows {rows}" if rows else "") + f" --steps {steps} --warmup {warm}",
                        "metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
                        "steps": steps, "warmup": warm, "workload": rec["config"]["workload"],
                        "roofline": {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms")},
                        "kernels": rec.get("kernels"), "wall_s": time.perf_counter() - t0}
            out[key]["parity"] = rec.get("parity")
            if "corridor" in rec:
                out[key]["corridor"] = rec["corridor"]
            if "pipeline" in rec:
                out[key]["pipeline"] = rec["pipeline"]
        except BaseException as e:  # noqa: BLE001 (SystemExit of a sub-run included)
            out[key] = {"error": repr(e), "wall_s": time.perf_counter() - t0}
    t0 = time.perf_counter()
    try:
        out["config1_latency"] = config1_latency()
    except BaseException as e:  # noqa: BLE001
        out["config1_latency"] = {"error": repr(e)}
    out["config1_latency"]["wall_s"] = time.perf_counter() - t0
    out["wall_s"] = time.perf_counter() - t_all
    return out


def respawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: this process becomes the launcher -- the same command
    line under torch.distributed.run, one rank per GPU (the shape the driver uses for N > 1), rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    port = args.master_port
    if port <= 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_ranks(args))
    out = run(args)
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0 and out is not None:
        if args.config == 2 and out["n_gpus"] == 1 and not args.no_other_configs and args.batch <= 0:
            out["other_configs"] = other_configs(args)
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
