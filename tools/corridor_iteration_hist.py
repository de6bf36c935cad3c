import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
lib = U.lib()
ctx = U.Context(0)
for cfg in (3, 5):
    if cfg == 3:
        r, n, M = 3, 65536, 16
        b = W.uniform_batch(3, n, M, r, time_mode="distance"); so = b["seg_offsets"]
        lo, hi = W.corridor_boxes(b, config_index=3); uni, mx = M, M
    else:
        r, n = 4, 16384
        b = W.ragged_batch(5, n, r); so = b["seg_offsets"]; uni, mx = 0, 24
        lo, hi = W.corridor_boxes(b, config_index=5)
    d_so = up(so); d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
    d_lo, d_hi = up(lo), up(hi)
    out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
    rc = lib.uavqp_solve_corridor_batch_device(ctx._h, r, n, uni, mx, d_so.data_ptr() if not uni else None, d["waypoints"].data_ptr(), d["times"].data_ptr(),
                                               d["bc"].data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr())
    ctx.synchronize()
    itn = it.cpu().numpy()
    h = np.bincount(itn)
    print("cfg", cfg, "mean", itn.mean(), "max", itn.max(), "hist", h[:40].tolist())
    # wave = 64 lanes = 64/3 trajectories (approx: groups of 21)
    g = itn[: (n // 21) * 21].reshape(-1, 21).max(axis=1)
    print("   mean over waves of max-iteration:", g.mean(), " p50", np.median(g), "p99", np.percentile(g, 99))
    print("   frac converged within PDAS (<=3 its):", (itn <= 3).mean(), " <=4:", (itn <= 4).mean())
