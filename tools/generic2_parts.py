"""Which part of the ragged pair kernel costs what: the same launch with the workspace round trip and / or the coefficient stores
compiled out (tools/ubench/libuavqp_g2_*.so, built with -DG2_NO_WS / -DG2_NO_OUT).  Run once per library:
UAVQP_LIB_PATH=tools/ubench/libuavqp_g2_NO_WS.so python tools/generic2_parts.py"""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r, n = 4, 32768
b = W.ragged_batch(4, n, r); so = b["seg_offsets"]; d_so = torch.from_numpy(so).to(dev); tot = int(so[-1])
d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
st = torch.zeros(n, dtype=torch.int32, device=dev)
out = torch.zeros(tot * 6 * r, dtype=torch.float64, device=dev)
for mode, wpc in ((2, 0), (2, 2), (2, 1)):
    ctx.set_settings(generic_lanes_per_traj=mode, generic_waves_per_cu=wpc)
    ms = timeit(lambda: ctx.solve_batch_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], out, st), s)
    print(json.dumps({"lib": os.environ.get("UAVQP_LIB_PATH", "default"), "mode": mode, "wpc": wpc, "us": ms * 1e3}), flush=True)
