#!/usr/bin/env python3
"""Timing of the corridor solve alone (BASELINE configs 3 and 5), HIP events around the launches on the ctx stream.
usage: python tools/corridor_bench.py [reps]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ctx = U.Context(0)
for cfg in (3, 5):
    if cfg == 3:
        r, n, M = 3, 65536, 16
        b = W.uniform_batch(3, n, M, r, time_mode="distance")
        uni, mx = M, M
    else:
        r, n = 4, 16384
        b = W.ragged_batch(5, n, r)
        uni, mx = 0, 24
    so = b["seg_offsets"]
    lo, hi = W.corridor_boxes(b, config_index=cfg)
    d_so = up(so)
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    d_lo, d_hi = up(lo), up(hi)
    out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev)

    def run():
        ctx.solve_corridor_device(r, n, uni, mx, None if uni else d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it)
    modes = (2, 1, 2, 1) if len(sys.argv) > 2 and sys.argv[2] == "ab" else (2,)   # A/B of uavqp_settings.corridor_initial_guess in one process (2: dual prelude, 1: closed-form set)
    for guess in modes:
        ctx.set_settings(corridor_initial_guess=guess)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        itn = it.cpu().numpy()
        print(json.dumps({"config": cfg, "initial_guess": guess, "n": n, "r": r, "ms_median": float(np.median(ms)), "ms_min": float(np.min(ms)),
                          "solved": int((st == U.UAVQP_SOLVED).sum().item()), "iters_mean": float(itn.mean()), "iters_max": int(itn.max()),
                          "checksum": float(out.abs().sum().item())}))
