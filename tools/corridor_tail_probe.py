"""Is config 5's corridor solve bound by its slowest problem?  Iteration histogram (per trajectory = max over the three axes) of the
cold solve and of every warm round of the outer loop, and the time of the cold solve as a function of the iteration cap
(uavqp_settings.max_iter): if the time follows the cap and not the mean, the tail is the bound.  GPU box: python tools/corridor_tail_probe.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from tools.bench_configs import timeit

dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = U.Context(0); ctx.set_stream(s.cuda_stream)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r, n, h_max = 4, 16384, 0.8
b = W.ragged_batch(5, n, r); so = b["seg_offsets"]; rows = int(so[-1]) + n
obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
d_so = torch.from_numpy(so).to(dev); d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}; d_obs = up(obs)
out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev)
st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev); ch = torch.zeros(n, dtype=torch.int32, device=dev)
d_lo = torch.zeros(rows * 3, dtype=torch.float64, device=dev); d_hi = torch.zeros(rows * 3, dtype=torch.float64, device=dev)
T0 = d["times"].clone()
ctx.solve_batch_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], out, st)
ctx.corridor_from_cloud_device(r, n, 0, d_so, rows, d["waypoints"], d["times"], out, d_obs, obs.shape[0], 0.4, 0.1, h_max, d_lo, d_hi)
act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
M = np.diff(so)


def hist(tag):
    x = it.cpu().numpy()
    q = np.percentile(x, [50, 90, 99, 99.9, 100])
    worst = np.argsort(-x)[:5]
    print(json.dumps({"what": tag, "iters_mean": float(x.mean()), "p50_p90_p99_p999_max": [float(v) for v in q], "n_over_30": int((x > 30).sum()),
                      "worst_M": [int(M[k]) for k in worst], "status_counts": {int(k): int(v) for k, v in zip(*np.unique(st.cpu().numpy(), return_counts=True))}}), flush=True)


cold = lambda: ctx.solve_corridor_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it, act, False)
cold(); s.synchronize(); hist("cold solve")
print(json.dumps({"cold_ms": timeit(cold, s, n=5, warm=1)}), flush=True)
for cap in (5, 10, 20, 30, 50, 80):
    ctx.set_settings(max_iter=cap)
    ms = timeit(cold, s, n=3, warm=1)
    print(json.dumps({"max_iter": cap, "ms": ms, "capped": int((st.cpu().numpy() != 1).sum())}), flush=True)
ctx.set_settings(max_iter=0)
for shape, pdas, WARM in ((0, 0, 2), (1, 0, 2), (0, 0, 2), (1, 0, 2)):
    ctx.set_settings(corridor_pdas_rounds_warm=pdas, corridor_tail_shape=shape)
    d["times"].copy_(T0)
    rows_ = []
    tot = 0.0
    for rnd in range(5):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(s)
        ctx.solve_corridor_device(r, n, 0, 24, d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it, act, (WARM if rnd > 0 else 0))
        ev1.record(s); s.synchronize()
        x = it.cpu().numpy(); ms = ev0.elapsed_time(ev1); tot += ms
        rows_.append((round(ms, 3), round(float(x.mean()), 2), int(x.max())))
        ctx.time_reallocate_device(r, n, 0, d_so, d["times"], out, 7.0, 10.0, samples_per_seg=16, max_stretch=2.0, changed=ch)
    print(json.dumps({"tail_shape": shape, "pdas_rounds_warm": pdas, "warm_mode": WARM, "total_ms": round(tot, 3), "per_round(ms, mean it, max it)": rows_, "solved": int((st.cpu().numpy() == 1).sum()),
                      "still_stretching": int((ch > 0).sum())}), flush=True)
ctx.set_settings(corridor_pdas_rounds_warm=0, corridor_tail_shape=1)
