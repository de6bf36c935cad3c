#!/usr/bin/env python3
"""Debug aid (GPU box, debug build: make -C uav_motion_planning_amd/csrc dual-debug; UAVQP_LIB_PATH=tools/ubench/libuavqp_dualdbg.so python
tools/pipeline_round_probe.py): what the waves of corridor_dual_wave_kernel did in the re-solve of config 5 that takes 1500-3000
trajectories (round 3: 2184) -- trips, cycles, segment counts, when each wave entered and left."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402
from uav_motion_planning_amd.pipeline import corridor_pipeline_device  # noqa: E402

r, n = 4, 16384
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
b = W.ragged_batch(4, n, r)
so = b["seg_offsets"]
obs = W.pillar_cloud(5, n_pillars=60, resolution=0.2)
with U.Context(0) as ctx:
    d_so, d_wp, d_T, d_bc, d_obs = up(so), up(np.asarray(b["waypoints"]).reshape(-1, 3)), up(b["times"]), up(b["bc"]), up(obs)
    res = corridor_pipeline_device(ctx, r, d_so, d_wp, d_T, d_bc, d_obs, int(np.diff(so).max()), repair_rounds=0)
    ctx.synchronize()
    print("rounds", res["rounds"])
    lib = U.lib()
    dump = np.zeros((64, 2048))
    lib.uavqp_debug_corridor_dual.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert lib.uavqp_debug_corridor_dual(ctx._h, dump.ctypes.data, None, n) == 0
d = dump.reshape(-1)
rec = d[:4 * 16384].reshape(16384, 4)
end = d[4 * 16384:5 * 16384]
m = (rec[:, 1] > 0) & (rec[:, 2] >= 1) & (rec[:, 2] <= 24) & (rec[:, 0] >= 0) & (rec[:, 0] < 1000) & (rec[:, 3] > 1e12) & (end > 1e12)   # (the first solve leaves other dumps in the same area)
print("waves recorded:", int(m.sum()))
if m.any():
    tr, cy, nn, t0 = rec[m, 0], rec[m, 1], rec[m, 2], rec[m, 3]
    t1 = end[m]
    base = t0.min()
    print("trips per trajectory: mean %.1f p90 %.0f max %.0f; cycles per trajectory: mean %.0f p90 %.0f max %.0f; knots mean %.1f" % (tr.mean(), np.percentile(tr, 90), tr.max(), cy.mean(), np.percentile(cy, 90), cy.max(), nn.mean()))
    print("corr(cycles, trips) %.2f  corr(cycles, knots) %.2f" % (np.corrcoef(cy, tr)[0, 1], np.corrcoef(cy, nn)[0, 1]))
    tick = 1e-2  # wall_clock64: 100 MHz
    print("entry times us: min 0 p50 %.1f p90 %.1f max %.1f; exit: p50 %.1f p90 %.1f max %.1f" % (np.percentile(t0 - base, 50) * tick, np.percentile(t0 - base, 90) * tick, (t0 - base).max() * tick,
          np.percentile(t1 - base, 50) * tick, np.percentile(t1 - base, 90) * tick, (t1 - base).max() * tick))
    order = np.argsort(t1)
    print("the five waves that left last: trips %s cycles %s knots %s entered %s us" % (tr[order[-5:]], cy[order[-5:]], nn[order[-5:]], np.round((t0[order[-5:]] - base) * tick, 1)))
    ex = d[5 * 16384:5 * 16384 + 4]
    print("entry of block 0 / of the last block of the grid: %.1f / %.1f us; first wave of the verifying solve enters at %.1f us" % ((ex[2] - base) * tick, (ex[3] - base) * tick, (ex[1] - base) * tick))
    w = d[63 * 2048 + 1400 + 16:63 * 2048 + 1400 + 48].reshape(16, 2)
    sv = d[63 * 2048 + 1400 + 64:63 * 2048 + 1400 + 96].reshape(16, 2)
    t_first = min(x for x in list(w[:, 0]) + list(sv[:, 0]) if x > 1e12)
    print("wave preludes   (entry us, trajectories):", [(round((t - t_first) * tick, 1), int(k)) for t, k in w if t > 1e12])
    print("verifying solves (entry us, problems)  :", [(round((t - t_first) * tick, 1), int(k)) for t, k in sv if t > 1e12])
