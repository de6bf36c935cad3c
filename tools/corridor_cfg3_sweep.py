"""Config 3 (65 536 x 16-segment jerk, corridor boxes): time and iteration counts of the cold corridor solve over
uavqp_settings.corridor_pdas_rounds x corridor_initial_guess.  GPU box: python tools/corridor_cfg3_sweep.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r, n, M = 3, 65536, 16
b = W.uniform_batch(3, n, M, r, time_mode="distance")
lo, hi = W.corridor_boxes(b, config_index=3)
d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}; d_lo, d_hi = up(lo), up(hi)
out = torch.zeros(n * M * 6 * r, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
run = lambda: ctx.solve_corridor_device(r, n, M, M, None, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it)
for guess in (1, 0):
    for pdas in (0, 1, 2, 3, 4, 6):
        ctx.set_settings(corridor_pdas_rounds=pdas, corridor_initial_guess=guess)
        run(); s.synchronize()
        x = it.cpu().numpy()
        ms = timeit(run, s, n=5, warm=1)
        print(json.dumps({"initial_guess": guess, "pdas_rounds": pdas, "ms": round(ms, 4), "iters_mean": round(float(x.mean()), 2), "iters_max": int(x.max()), "solved": int((st.cpu().numpy() == 1).sum())}), flush=True)
