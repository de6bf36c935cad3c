"""GPU probe: config 3 + K = 2 rows at full size; which draws do not end UAVQP_SOLVED -> gpurun_out/rows_unsolved.npz (classified on the CPU
afterwards with the OSQP port: tools/rows_unsolved_classify.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W
from test_gpu_rows import run_rows
r, n, M, K = 3, 65536, 16, 2
b = W.uniform_batch(3, n, M, r, time_mode="distance")
lo, hi = W.corridor_boxes(b, config_index=3)
tau, drv, rlo, rhi = W.config3_rows(b, K)
with U.Context(0) as ctx:
    coef, st, it, act = run_rows(ctx, r, b, lo.reshape(-1, 3), hi.reshape(-1, 3), K, tau, drv, rlo, rhi, M)
bad = np.nonzero(st != U.UAVQP_SOLVED)[0]
print("statuses", np.unique(st, return_counts=True), "iters solved mean/max", it[st == 1].mean(), it[st == 1].max(), "iters unsolved", np.unique(it[bad], return_counts=True))
np.savez(os.path.join(ROOT, "gpurun_out", "rows_unsolved.npz"), bad=bad, st=st[bad], it=it[bad], act=act[bad], coef=coef.reshape(n, -1)[bad])
