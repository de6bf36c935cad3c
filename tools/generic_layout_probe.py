"""Generic kernel: one lane per trajectory (NAX=3) vs one lane per (trajectory, axis) (NAX=1) over batch sizes.
Run on the GPU box: python tools/generic_layout_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream); ctx.set_variant(1)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r = 4
for M in (14, 24):
    for n in (64, 256, 1024, 4096, 8192, 16384, 32768, 65536):
        b = W.uniform_batch(4, n, M, r, time_mode="distance")
        d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
        out = torch.zeros(n * 3 * M * 8, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
        res = {}
        for nax in (3, 1):
            os.environ["UAVQP_GENERIC_NAX"] = str(nax)
            res[nax] = timeit(lambda: ctx.solve_batch_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], out, st), s) * 1e3
        print("M=%d n=%6d  lane/traj %.1f us   lane/(traj,axis) %.1f us" % (M, n, res[3], res[1]))
