#!/usr/bin/env python3
"""Device-resident timings of BASELINE.json's other configs (parity-test shapes, not bench.py lines):
config 3 (65536 x 16-segment min-jerk + corridors), config 4 (32768 ragged min-snap, M in [4, 24]),
evaluation (N1).  Prints one JSON line per measurement.  Run on the GPU box: python tools/bench_configs.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, stream, n=20, warm=3):
    import torch
    for _ in range(warm):
        fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n):
        fn()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    import torch
    import uav_motion_planning_amd as U
    from uav_motion_planning_amd import workloads as W
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s)
    ctx = U.Context(0)
    ctx.set_stream(s.cuda_stream)

    def up(x, dt=torch.float64):
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev)

    # ---- config 3: 65536 x (M=16, r=3), equality-only and with corridors
    r, M, n = 3, 16, 65536
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b)
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    d_lo, d_hi = up(lo), up(hi)
    out = torch.zeros(n * 3 * M * 2 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev)
    ms = timeit(lambda: ctx.solve_batch_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], out, st), s)
    print(json.dumps({"config": "3-equality", "n": n, "M": M, "r": r, "ms": ms, "traj_per_s": n / ms * 1e3,
                      "GB_per_s": n * W.algorithmic_bytes(r, M) / ms / 1e6}))
    lib = U.lib()
    import ctypes

    def corridor():
        rc = lib.uavqp_solve_corridor_batch_device(ctx._h, r, n, M, M, None, d["waypoints"].data_ptr(), d["times"].data_ptr(),
                                                   d["bc"].data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), out.data_ptr(),
                                                   st.data_ptr(), it.data_ptr())
        assert rc == 0
    ms = timeit(corridor, s, n=5, warm=1)
    bytes_c = W.algorithmic_bytes(r, M) + 8 * 2 * 3 * (M - 1)
    print(json.dumps({"config": "3-corridor", "n": n, "M": M, "r": r, "ms": ms, "traj_per_s": n / ms * 1e3,
                      "GB_per_s": n * bytes_c / ms / 1e6, "solved": int((st == 1).sum()), "iters_mean": float(it.float().mean()),
                      "iters_max": int(it.max())}))

    # ---- config 3 with its "K = 2 mid-segment samples" as GENERAL rows (uavqp_solve_rows_batch_device): a position sample at
    # mid-segment inside the chord +- 0.25 m and a velocity limit there, on top of the knot boxes
    K = 2
    wpn = b["waypoints"]
    tau = np.full((n * M, K), 0.5)
    drv = np.tile(np.array([0, 1], dtype=np.int32), (n * M, 1))
    mid = 0.5 * (wpn[:, :-1] + wpn[:, 1:]).reshape(n * M, 3)
    rlo, rhi = np.zeros((n * M, K, 3)), np.zeros((n * M, K, 3))
    rlo[:, 0], rhi[:, 0] = mid - 0.25, mid + 0.25
    rlo[:, 1], rhi[:, 1] = -3.5, 3.5
    d_tau, d_drv, d_rlo, d_rhi = up(tau), up(drv), up(rlo), up(rhi)
    ms = timeit(lambda: ctx.solve_rows_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, K, d_tau, d_drv, d_rlo, d_rhi,
                                              out, st, it), s, n=3, warm=1)
    print(json.dumps({"config": "3-corridor+rows(K=2: mid-segment position sample, velocity limit)", "n": n, "M": M, "r": r, "ms": ms,
                      "traj_per_s": n / ms * 1e3, "solved": int((st == 1).sum()), "capped": int((st == -2).sum()),
                      "iters_mean": float(it.float().mean()), "iters_max": int(it.max())}))

    # ---- config 4: 32768 ragged (M in [4, 24]), r=4
    r, n = 4, 32768
    b = W.ragged_batch(4, n, r)
    so = b["seg_offsets"]
    d_so = torch.from_numpy(so).to(dev)
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    out = torch.zeros(int(so[-1]) * 3 * 2 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    ms = timeit(lambda: ctx.solve_batch_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], out, st), s)
    byts = 8 * (3 * (so[-1] + n) + so[-1] + 18 * n) + 8 * 24 * so[-1] + 4 * n
    print(json.dumps({"config": "4-ragged", "n": n, "sum_M": int(so[-1]), "r": r, "ms": ms, "traj_per_s": n / ms * 1e3,
                      "GB_per_s": float(byts) / ms / 1e6, "solved": int((st == 1).sum())}))

    # ---- N1: evaluation of config-2 trajectories at 100 samples (pos, vel, acc)
    r, M, n, ns = 4, 8, 4096, 100
    b = W.uniform_batch(2, n, M, r, time_mode="distance")
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    out = torch.zeros(n * 3 * M * 2 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.solve_batch_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], out, st)
    ev = torch.zeros(n * ns * 9, dtype=torch.float64, device=dev)
    ms = timeit(lambda: ctx.eval_batch_device(r, n, M, None, d["times"], out, ns, 0.0, 0.05, 7, ev), s)
    print(json.dumps({"config": "N1-eval", "n": n, "samples": ns, "ms": ms, "samples_per_s": n * ns / ms * 1e3,
                      "GB_per_s": (n * ns * 72 + n * 1536) / ms / 1e6}))

    # ---- config 5: 16384 ragged (M in [4, 24]), r=4: plain solve -> corridor boxes from a pillar cloud (attitude of
    # that solve) -> corridor solve + time re-allocation, 5 outer rounds (the cap of SURVEY.md section 8-d)
    r, n, h_max = 4, 16384, 0.8
    b = W.ragged_batch(5, n, r)
    so = b["seg_offsets"]
    rows = int(so[-1]) + n
    obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
    d_so = torch.from_numpy(so).to(dev)
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    d_obs = up(obs)
    out = torch.zeros(int(so[-1]) * 3 * 2 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev)
    ch = torch.zeros(n, dtype=torch.int32, device=dev)
    d_lo = torch.zeros(rows * 3, dtype=torch.float64, device=dev)
    d_hi = torch.zeros(rows * 3, dtype=torch.float64, device=dev)
    T0 = d["times"].clone()
    ms_solve = timeit(lambda: ctx.solve_batch_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], out, st), s)
    ms_cloud = timeit(lambda: ctx.corridor_from_cloud_device(r, n, 0, d_so, rows, d["waypoints"], d["times"], out, d_obs, obs.shape[0],
                                                             0.4, 0.1, h_max, d_lo, d_hi), s, n=5, warm=1)
    width = (d_hi - d_lo).reshape(rows, 3)

    def corridor5():
        rc = lib.uavqp_solve_corridor_batch_device(ctx._h, r, n, 0, 24, d_so.data_ptr(), d["waypoints"].data_ptr(), d["times"].data_ptr(),
                                                   d["bc"].data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr())
        assert rc == 0
    ms_corr = timeit(corridor5, s, n=5, warm=1)
    it_mean = float(it.float().mean())

    def outer_loop():
        d["times"].copy_(T0)
        for _ in range(5):
            corridor5()
            ctx.time_reallocate_device(r, n, 0, d_so, d["times"], out, 7.0, 10.0, samples_per_seg=16, max_stretch=2.0, changed=ch)
    with torch.cuda.stream(s):
        ms_loop = timeit(outer_loop, s, n=3, warm=1)
    still_cold = int((ch > 0).sum())
    act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
    it_rounds = []

    def outer_loop_warm(record=False):
        d["times"].copy_(T0)
        for rnd in range(5):
            ctx.solve_corridor_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it, act, 2 if rnd > 0 else 0)
            if record:
                it_rounds.append(float(it.float().mean()))
            ctx.time_reallocate_device(r, n, 0, d_so, d["times"], out, 7.0, 10.0, samples_per_seg=16, max_stretch=2.0, changed=ch)
    with torch.cuda.stream(s):
        ms_loop_warm = timeit(outer_loop_warm, s, n=3, warm=1)
        outer_loop_warm(record=True)
    # final validation of config 5: SE(3) ellipsoid check of every trajectory at 100 samples, exhaustive vs grid
    ns = 100
    tot_T = float(d["times"].sum() / n)
    fh = torch.zeros(n, dtype=torch.int32, device=dev)
    ms_check = timeit(lambda: ctx.ellipsoid_check_device(r, n, 0, d_so, d["times"], out, ns, 0.0, tot_T / ns, d_obs, obs.shape[0], 0.4, 0.1, fh), s, n=3, warm=1)
    t0 = time.perf_counter()
    grid = ctx.obstacle_grid_build(d_obs, obs.shape[0], 0.5)
    ms_grid_build = (time.perf_counter() - t0) * 1e3
    fh2 = torch.zeros(n, dtype=torch.int32, device=dev)
    ms_check_grid = timeit(lambda: ctx.ellipsoid_check_grid_device(r, n, 0, d_so, d["times"], out, ns, 0.0, tot_T / ns, grid, 0.4, 0.1, fh2), s, n=10, warm=2)
    same = bool(torch.equal(fh, fh2))
    ctx.obstacle_grid_destroy(grid)
    print(json.dumps({"config": "5-validation", "n": n, "samples": ns, "n_obs": int(obs.shape[0]), "ms_exhaustive": ms_check,
                      "ms_grid": ms_check_grid, "ms_grid_build_host_sync": ms_grid_build, "identical_first_hits": same,
                      "colliding_trajectories": int((fh2 < ns).sum())}))
    print(json.dumps({"config": "5-pipeline", "n": n, "sum_M": int(so[-1]), "r": r, "n_obs": int(obs.shape[0]),
                      "ms_plain_solve": ms_solve, "ms_cloud_corridor": ms_cloud,
                      "cloud_pairs_per_s": rows * obs.shape[0] / ms_cloud * 1e3,
                      "ms_corridor_solve": ms_corr, "iters_mean": it_mean, "ms_5_outer_rounds_cold": ms_loop,
                      "ms_5_outer_rounds": ms_loop_warm, "iters_mean_per_round_warm": it_rounds, "still_stretching_cold": still_cold,
                      "traj_per_s_whole_pipeline": n / (ms_solve + ms_cloud + ms_loop_warm) * 1e3,
                      "solved": int((st == 1).sum()), "still_stretching": int((ch > 0).sum()),
                      "box_halfwidth_mean_xyz": [float(x) for x in (width.mean(dim=0) / 2)],
                      "rows_degenerate_frac": float((width.amax(dim=1) == 0).float().mean())}))


if __name__ == "__main__":
    main()
