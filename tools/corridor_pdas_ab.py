"""A/B of uavqp_settings.corridor_pdas_rounds (block-pivot rounds before the single-pivot phase) on BASELINE configs 3 and 5, one process,\nHIP events around the solve: 3 rounds are the optimum with the closed-form starting set as well (config 3: 12.46 iterations mean; 2: 12.79,\n4: 13.07, 5: 13.54, 1: 15.23).  GPU box: python tools/corridor_pdas_ab.py"""
import json, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
dev = torch.device("cuda", 0); up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ctx = U.Context(0)
for cfg in (3, 5):
    if cfg == 3:
        r, n, M = 3, 65536, 16; b = W.uniform_batch(3, n, M, r, time_mode="distance"); uni, mx = M, M
    else:
        r, n = 4, 16384; b = W.ragged_batch(5, n, r); uni, mx = 0, 24
    so = b["seg_offsets"]; lo, hi = W.corridor_boxes(b, config_index=cfg); d_so = up(so)
    d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}; d_lo, d_hi = up(lo), up(hi)
    out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
    run = lambda: ctx.solve_corridor_device(r, n, uni, mx, None if uni else d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it)
    for rounds in (3, 2, 4, 5, 1, 3):
        ctx.set_settings(corridor_pdas_rounds=rounds)
        for _ in range(2): run()
        torch.cuda.synchronize(); ms = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        print(json.dumps({"config": cfg, "pdas_rounds": rounds, "ms_median": float(np.median(ms)), "iters_mean": float(it.float().mean().item()), "iters_max": int(it.max().item())}))
