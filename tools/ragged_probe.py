import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r, n = 4, 32768
# uniform M = 14 (and 24) through the generic kernel
b = W.uniform_batch(4, n, 14, r, time_mode="distance")
d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
out = torch.zeros(n * 3 * 14 * 8, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.set_variant(1)
ms = timeit(lambda: ctx.solve_batch_device(r, n, 14, 14, None, d["waypoints"], d["times"], d["bc"], out, st), s)
print("uniform M=14 generic: %.1f us" % (ms * 1e3))
b24 = W.uniform_batch(4, n, 24, r, time_mode="distance"); d24 = {k: up(b24[k]) for k in ("waypoints", "times", "bc")}; out24 = torch.zeros(n * 3 * 24 * 8, dtype=torch.float64, device=dev)
ms = timeit(lambda: ctx.solve_batch_device(r, n, 24, 24, None, d24["waypoints"], d24["times"], d24["bc"], out24, st), s)
print("uniform M=24 generic: %.1f us" % (ms * 1e3))
# ragged, as is, and pre-sorted by M on the host
b = W.ragged_batch(4, n, r); so = b["seg_offsets"]; Ms = np.diff(so)
def run(bb, tag):
    so = bb["seg_offsets"]; d_so = torch.from_numpy(so).to(dev)
    d = {k: up(bb[k]) for k in ("waypoints", "times", "bc")}
    out = torch.zeros(int(so[-1]) * 24, dtype=torch.float64, device=dev)
    ms = timeit(lambda: ctx.solve_batch_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], out, st), s)
    print(tag, "%.1f us" % (ms * 1e3))
run(b, "ragged unsorted:")
order = np.argsort(-Ms, kind="stable")
wp = np.asarray(b["waypoints"]); T = b["times"]
wps = np.concatenate([wp[so[k] + k:so[k + 1] + k + 1] for k in order]); Ts = np.concatenate([T[so[k]:so[k + 1]] for k in order])
sos = np.zeros(n + 1, dtype=np.int32); sos[1:] = np.cumsum(Ms[order])
run(dict(seg_offsets=sos, waypoints=wps, times=Ts, bc=b["bc"][order]), "ragged sorted by M (host):")
