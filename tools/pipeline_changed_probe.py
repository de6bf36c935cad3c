#!/usr/bin/env python3
"""How many trajectories of config 5 does each re-allocation round stretch?  (GPU box)  uavqp_pipeline_result.still_stretching after
max_rounds = 1..5 is the count the k-th re-allocation changed: the size of the next round's re-solve."""
import sys
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from uav_motion_planning_amd.pipeline import corridor_pipeline_device
r, n, mx = 4, 16384, 24
b = W.ragged_batch(5, n, r)
obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d_so, d_wp, d_bc, d_obs = up(b["seg_offsets"]), up(np.asarray(b["waypoints"]).reshape(-1, 3)), up(b["bc"]), up(obs)
with U.Context(0) as ctx:
    for k in range(1, 6):
        d_T = up(b["times"])
        res = corridor_pipeline_device(ctx, r, d_so, d_wp, d_T, d_bc, d_obs, mx, max_rounds=k, repair_rounds=0)
        print("max_rounds", k, "rounds", res["rounds"], "changed by the last re-allocation:", res["still_stretching"])
