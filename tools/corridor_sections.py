#!/usr/bin/env python3
"""Cycles per section of the corridor solver (wave 0), from a -DUAVQP_CORRIDOR_TIMING build of the library:
   make -C uav_motion_planning_amd/csrc timing     (-> tools/ubench/libuavqp_timing.so)
   UAVQP_LIB_PATH=$PWD/tools/ubench/libuavqp_timing.so python tools/corridor_sections.py"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402

dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ctx = U.Context(0)
lib = U.lib()
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
    for _ in range(3):
        ctx.solve_corridor_device(r, n, uni, mx, None if uni else d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st)
    s = (ctypes.c_longlong * 7)()
    assert lib.uavqp_debug_corridor_stamps(ctx._h, s) == 0
    v = np.array(list(s), dtype=np.float64)
    names = ["refill", "forward", "meeting", "backward", "decide", "hand-over"]
    tot = v[:6].sum()
    print(f"config {cfg}: wave 0 ran {int(v[6])} iterations, {tot:.0f} cycles; per iteration: " +
          ", ".join(f"{nm} {v[k] / v[6]:.0f}" for k, nm in enumerate(names)))
