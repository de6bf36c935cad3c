"""Generic kernel, lane per trajectory: resident waves per CU (workspace = slots x (M-1) x F doubles) vs batch size."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream); ctx.set_variant(1)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for r, M in ((4, 14), (3, 20)):
    for n in (32768, 65536, 131072, 262144):
        b = W.uniform_batch(4, n, M, r, time_mode="distance")
        d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
        out = torch.zeros(n * 3 * M * 2 * r, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
        row = []
        for wpc in (1, 2, 3, 4, 8):
            os.environ["UAVQP_GENERIC_WPC"] = str(wpc)
            row.append("%d:%.0f" % (wpc, timeit(lambda: ctx.solve_batch_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], out, st), s, n=10) * 1e3))
        print("r=%d M=%d n=%6d  us by waves/CU  " % (r, M, n) + "  ".join(row))
