#!/usr/bin/env python3
"""Cycles per knot step of the corridor solver's sweeps as a function of the segment count (r = 4: 5 own knots per lane in LDS, the
rest in the HBM workspace), from the -DUAVQP_CORRIDOR_TIMING build:  make -C uav_motion_planning_amd/csrc timing;
UAVQP_LIB_PATH=$PWD/tools/ubench/libuavqp_timing.so python tools/corridor_trip_probe.py"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402

dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ctx = U.Context(0); lib = U.lib()
for r, n, Ms in ((4, 8192, (6, 8, 10, 12, 16, 20, 24)), (3, 8192, (8, 16, 20, 24))):
    for M in Ms:
        b = W.uniform_batch(3, n, M, r, time_mode="distance")
        lo, hi = W.corridor_boxes(b, config_index=3)
        d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
        d_lo, d_hi = up(lo), up(hi)
        out = torch.zeros(n * M * 6 * r, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
        for _ in range(2):
            ctx.solve_corridor_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st)
        ctx.synchronize()
        s = (ctypes.c_longlong * 7)()
        assert lib.uavqp_debug_corridor_stamps(ctx._h, s) == 0
        v = np.array(list(s), dtype=np.float64); it = v[6]; own = (M + 1) // 2
        print(f"r={r} M={M:2d} own knots {own:2d}: iterations of wave 0 {int(it):4d}; per iteration: refill {v[0]/it:6.0f} forward {v[1]/it:6.0f} ({v[1]/it/own:5.0f}/knot) "
              f"meeting {v[2]/it:5.0f} backward {v[3]/it:6.0f} ({v[3]/it/own:5.0f}/knot) decide {v[4]/it:5.0f} hand-over {v[5]/it:5.0f}  total {v[:6].sum()/it:7.0f}", flush=True)
