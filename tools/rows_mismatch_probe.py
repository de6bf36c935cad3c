import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from test_gpu_rows import run_rows
r, M, K, n = 4, 12, 2, 64
b = W.ragged_batch(5, n, r, m_lo=2, m_hi=M, seed=77)
so = np.asarray(b["seg_offsets"]); tot = int(so[-1]); wp = np.asarray(b["waypoints"])
mid = np.concatenate([0.5 * (wp[so[k] + k:so[k + 1] + k] + wp[so[k] + k + 1:so[k + 1] + k + 1]) for k in range(n)])
lo, hi = W.corridor_boxes(b, config_index=3)
rng = np.random.default_rng(3)
tau = np.tile(np.array([0.5, 0.5])[:K], (tot, 1))
tau[rng.random(tot) < 0.3, 0] = 0.3
tau[rng.random(tot) < 0.05, 1] = 0.0
drv = np.tile(np.array([0, 1])[:K], (tot, 1))
drv[rng.random(tot) < 0.1, K - 1] = -1
rlo, rhi = np.zeros((tot, K, 3)), np.zeros((tot, K, 3))
rlo[:, 0], rhi[:, 0] = mid - 0.25, mid + 0.25
rlo[:, 1], rhi[:, 1] = -3.5, 3.5
res = {}
with U.Context(0) as ctx:
    for guess in (2, 1):
        ctx.set_settings(corridor_initial_guess=guess)
        res[guess] = run_rows(ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, 0)
bad = np.nonzero(res[1][1] != res[2][1])[0]
print("mismatching trajectories", bad, "status prelude", res[2][1][bad], "box-start", res[1][1][bad], "iters", res[2][2][bad], res[1][2][bad], "M", np.diff(so)[bad])
print("non-solved counts: prelude", int((res[2][1] != 1).sum()), "box start", int((res[1][1] != 1).sum()))
for k in bad:
    print(k, "sets prelude", [hex(int(v)) for v in res[2][3][k].reshape(-1).astype(np.uint64)], "\n   box-start  ", [hex(int(v)) for v in res[1][3][k].reshape(-1).astype(np.uint64)])
    sl = slice(so[k], so[k + 1]); print("   tau", tau[sl].T, "drv", drv[sl].T)
