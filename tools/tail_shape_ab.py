#!/usr/bin/env python3
"""GPU probe: BASELINE config 5's pipeline pass with uavqp_settings.corridor_tail_shape = 1 (two waves per CU, ten own knots per lane in LDS: round 3's choice for
long iteration chains) against 0 (four waves per CU, five knots in LDS, the rest through the HBM workspace), now that a verifying solve is ONE block solve per
problem.  ms per pass, median of 9."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from uav_motion_planning_amd.pipeline import corridor_pipeline_device
r, n, mx = 4, 16384, 24
b = W.ragged_batch(5, n, r)
obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d_so, d_wp, d_bc, d_obs = up(b["seg_offsets"]), up(np.asarray(b["waypoints"]).reshape(-1, 3)), up(b["bc"]), up(obs)
T0 = up(b["times"])
with U.Context(0) as ctx:
    grid = ctx.obstacle_grid_build(d_obs, d_obs.shape[0], 0.5)
    for shape in (1, 0, 1, 0):
        ctx.set_settings(corridor_tail_shape=shape)
        out, ts = None, []
        for it in range(12):
            d_T = T0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = corridor_pipeline_device(ctx, r, d_so, d_wp, d_T, d_bc, d_obs, mx, grid=grid, repair_rounds=0, out=out)
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
            out = {k: res[k] for k in ("coeff", "status", "corr_lo", "corr_hi", "first_hit")}
        print("corridor_tail_shape", shape, "ms per pass (median of the last 9): %.3f" % (np.median(ts[3:]) * 1e3), "rounds", res["rounds"], "all solved", res["all_solved"])
