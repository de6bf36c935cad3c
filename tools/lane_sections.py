"""GPU probe (build: hipcc ... -DUAVQP_LANE_TIMING -> tools/ubench/libuavqp_lanetiming.so, UAVQP_LIB_PATH): cycles block 0 of corridor_dual_lane_kernel
spends per section on config 3 (validation, forward chain, backward + G, axis inits, selection, direction + ratio test, pivot + sweep)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
r, n, M = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 16
b = W.uniform_batch(3, n, M, r, time_mode="distance")
lo, hi = W.corridor_boxes(b, config_index=3)
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d = [up(b["waypoints"].reshape(-1, 3)), up(b["times"].reshape(-1)), up(b["bc"]), up(lo.reshape(-1, 3)), up(hi.reshape(-1, 3))]
out = torch.zeros(n * M * 18, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
with U.Context(0) as ctx:
    for _ in range(3):
        ctx.solve_corridor_device(r, n, M, M, None, d[0], d[1], d[2], d[3], d[4], out, st, it)
    ctx.synchronize()
    o = (ctypes.c_longlong * 8)()
    U.lib().uavqp_debug_lane_stamps(ctx._h, o)
    v = list(o)
    trips = v[7] >> 20
    names = ["validate", "forward chain", "backward + G", "axis inits", "selection", "direction + ratio", "pivot + sweep + hand-over"]
    tot = sum(v[:7])
    for nm, c in zip(names, v[:7]):
        print(f"{nm:28s} {c:10d} cycles  {100.0 * c / tot:5.1f} %")
    print("wave trips", trips, " cycles per trip (selection..sweep)", (v[4] + v[5] + v[6]) / max(trips, 1), " total", tot, "cycles =", tot / 2.4e3, "us at 2.4 GHz")
