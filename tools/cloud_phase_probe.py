"""GPU probe (build: hipcc ... -DUAVQP_CLOUD_STATS -> tools/ubench/libuavqp_cloudstats.so, UAVQP_LIB_PATH): config 5's cloud boxes -- blocks of
cloud_grid2d_kernel per ring, and the distribution of the radius min(g_cap, g) max(r, h) a row needs (from the exhaustive clearance)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
r, n = 4, 16384
b = W.ragged_batch(5, n, r)
so = b["seg_offsets"]
wp = np.asarray(b["waypoints"]).reshape(-1, 3)
obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rows = wp.shape[0]
with U.Context(0) as ctx:
    coef, st = ctx.solve_batch_host(r, so, b["waypoints"], b["times"], b["bc"])
    d_so, d_wp, d_T, d_coef, d_obs = up(so.astype(np.int32)), up(wp), up(np.asarray(b["times"]).reshape(-1)), up(coef), up(obs)
    lo = torch.zeros((rows, 3), dtype=torch.float64, device=dev); hi = torch.zeros_like(lo); g = torch.zeros(rows, dtype=torch.float64, device=dev)
    ctx.corridor_from_cloud_device(r, n, 0, d_so, rows, d_wp, d_T, d_coef, d_obs, obs.shape[0], 0.4, 0.1, 0.8, lo, hi, g)   # exhaustive (clearance wanted)
    ctx.synchronize()
    gg = g.cpu().numpy()
    need = np.minimum(gg, 25.0) * 0.4
    print("clearance g percentiles 10/50/90/99:", np.percentile(gg, [10, 50, 90, 99]))
    print("needed radius [m] percentiles 10/50/90/99:", np.percentile(need, [10, 50, 90, 99]), " share <= 2.5 m:", (need <= 2.5).mean(), " <= 5 m:", (need <= 5.0).mean())
    for mode in (1, 2, 3):
        ctx.set_settings(cloud_window=mode)
        for _ in range(3):
            ctx.corridor_from_cloud_device(r, n, 0, d_so, rows, d_wp, d_T, d_coef, d_obs, obs.shape[0], 0.4, 0.1, 0.8, lo, hi, None)
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream()); 
        import time; t0 = time.perf_counter()
        for _ in range(10):
            ctx.corridor_from_cloud_device(r, n, 0, d_so, rows, d_wp, d_T, d_coef, d_obs, obs.shape[0], 0.4, 0.1, 0.8, lo, hi, None)
        ctx.synchronize()
        print("cloud_window", mode, "ms per call (incl. sorts)", (time.perf_counter() - t0) / 10 * 1e3)
    lib = U.lib()
    if hasattr(lib, "uavqp_debug_cloud_phases"):
        out = (ctypes.c_uint * 4)()
        lib.uavqp_debug_cloud_phases(ctx._h if hasattr(ctx, "_h") else ctx.handle, out)
        print("blocks per ring 0/1/2:", list(out)[:3], "of", (rows + 255) // 256)

# ---- what an exact bounding-box cull would remove: metric to ANY point of the cloud >= max over axes of dist(p, bbox)_axis / sqrt((Q^-1)_axis,axis),
# (Q^-1)_aa = r^2 - (r^2 - h^2) b3_a^2
Ms = np.diff(so)
traj = np.repeat(np.arange(n), Ms + 1)
k = np.arange(rows) - (so[traj] + traj)
M_r = Ms[traj]
seg = np.minimum(k, M_r - 1)
acc = np.zeros((rows, 3))
for ax in range(3):
    base = 3 * 8 * so[traj].astype(np.int64) + (ax * M_r + seg) * 8
    c = coef[base[:, None] + np.arange(8)[None, :]]
    t = np.where(k < M_r, 0.0, np.asarray(b["times"]).reshape(-1)[so[traj] + seg])
    acc[:, ax] = sum(j * (j - 1) * c[:, j] * t ** (j - 2) for j in range(2, 8))
b3 = acc + np.array([0, 0, 9.81]); b3 /= np.linalg.norm(b3, axis=1, keepdims=True)
qinv = 0.16 - (0.16 - 0.01) * b3 ** 2
lo_b, hi_b = obs.min(axis=0), obs.max(axis=0)
dist = np.maximum(np.maximum(lo_b - wp, wp - hi_b), 0.0)
g_lb = np.max(dist / np.sqrt(qinv), axis=1)
qd = np.stack([(1 - b3[:, i] ** 2) / 0.16 + b3[:, i] ** 2 / 0.01 for i in range(3)], axis=1)      # diag of Q
gcap = 1 + 3 * 0.8 * np.sqrt(qd.max(axis=1))
culled = g_lb >= gcap
print("rows culled by the bounding-box bound:", culled.mean())
need_nc = np.minimum(gg, gcap)[~culled] * 0.4
print("non-culled rows: needed isotropic radius percentiles 10/50/90:", np.percentile(need_nc, [10, 50, 90]), "share <= 2.5:", (need_nc <= 2.5).mean(), "<= 5:", (need_nc <= 5).mean())
inside = np.all((wp >= lo_b) & (wp <= hi_b), axis=1)
print("rows inside the cloud's bounding box:", inside.mean(), " their needed radius percentiles:", np.percentile((np.minimum(gg, gcap) * 0.4)[inside], [10, 50, 90]))
