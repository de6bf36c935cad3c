import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from uav_motion_planning_amd.pipeline import corridor_pipeline_device
r, n, mx = 4, 16384, 24
b = W.ragged_batch(5, n, r); so = b["seg_offsets"]; wp = np.asarray(b["waypoints"]).reshape(-1, 3)
first = (so[:-1] + np.arange(n)).astype(np.int64)
obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
dev = torch.device("cuda", 0); up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for rep in (0, 2, 3):
    d_so, d_wp, d_T, d_bc, d_obs = up(so), up(wp), up(b["times"]), up(b["bc"]), up(obs)
    with U.Context(0) as ctx:
        res = corridor_pipeline_device(ctx, r, d_so, d_wp, d_T, d_bc, d_obs, mx, repair_rounds=rep)
        print(json.dumps({"repair_rounds": rep, "colliding_before": res["colliding_before_repair"], "colliding_after": int((~res["collision_free"]).sum().item()),
                          "repairs": res["repairs"], "blocked_waypoints": res["colliding_with_blocked_waypoints"], "solved": int((res["status"] == 1).sum().item())}))
