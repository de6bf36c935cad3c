#!/usr/bin/env python3
"""Debug aid for qp_rows_dual.h (GPU box, debug build: make -C uav_motion_planning_amd/csrc dual-debug;
UAVQP_LIB_PATH=tools/ubench/libuavqp_dualdbg.so python tools/rows_dual_gpu_probe.py): G on the refined grid and the unconstrained
constraint values of the first trajectories against dense numpy; starting sets against the working sets the rows solve ends with;
iterations of the rows kernel with and without the prelude."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tools')
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402
from corridor_strategy_probe import hessian  # noqa: E402

r, M, K = 3, 16, 2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
b = W.uniform_batch(3, n, M, r, time_mode="distance")
lo, hi = W.corridor_boxes(b, config_index=3)
wp = b["waypoints"]
tau = np.full((n * M, K), 0.5)
drv = np.tile(np.array([0, 1], dtype=np.int32), (n * M, 1))
mid = 0.5 * (wp[:, :-1] + wp[:, 1:]).reshape(n * M, 3)
rlo, rhi = np.zeros((n * M, K, 3)), np.zeros((n * M, K, 3))
rlo[:, 0], rhi[:, 0] = mid - 0.25, mid + 0.25
rlo[:, 1], rhi[:, 1] = -3.5, 3.5
d_wp, d_T, d_bc = up(wp.reshape(-1, 3)), up(b["times"].reshape(-1)), up(b["bc"])
d_lo, d_hi, d_tau, d_drv, d_rlo, d_rhi = up(lo.reshape(-1, 3)), up(hi.reshape(-1, 3)), up(tau), up(drv), up(rlo), up(rhi)
res = {}
have_dbg = False
with U.Context(0) as ctx:
    for guess in (2, 1):
        ctx.set_settings(corridor_initial_guess=guess)
        out = torch.zeros(n * 3 * M * 2 * r, dtype=torch.float64, device=dev)
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        it = torch.zeros(n, dtype=torch.int32, device=dev)
        act = torch.zeros((n, 3, 2 + 2 * K), dtype=torch.int64, device=dev)
        ctx.solve_rows_device(r, n, M, M, None, d_wp, d_T, d_bc, d_lo, d_hi, K, d_tau, d_drv, d_rlo, d_rhi, out, st, it, act)
        torch.cuda.synchronize()
        res[guess] = (out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy().astype(np.uint64))
        if guess == 2 and hasattr(U.lib(), "uavqp_debug_corridor_dual"):
            lib = U.lib()
            dump = np.zeros((32, 4096))
            box = np.zeros((n, 3, 2), dtype=np.uint64)
            lib.uavqp_debug_corridor_dual.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            have_dbg = lib.uavqp_debug_corridor_dual(ctx._h, dump.ctypes.data, box.ctypes.data, n) == 0
print("rows solve: iterations mean %.2f max %d with the prelude; %.2f / %d from the box set; statuses equal %s; coefficients equal to %.1e; working sets equal %s" % (
    res[2][2].mean(), res[2][2].max(), res[1][2].mean(), res[1][2].max(), np.array_equal(res[2][1], res[1][1]),
    np.max(np.abs(res[2][0] - res[1][0])) / np.max(np.abs(res[1][0])), np.array_equal(res[2][3], res[1][3])))
def hermite_functional(r, T, tau, d):
    """g_l, g_r with p^(d)(tau T) = g_l' x_start + g_r' x_end for the degree-(2r-1) polynomial matching derivatives 0..r-1 at both ends."""
    import math
    nc = 2 * r
    A = np.zeros((nc, nc))
    for k in range(r):
        A[k, k] = math.factorial(k)                                   # p^(k)(0)
        for j in range(k, nc):
            A[r + k, j] = math.factorial(j) / math.factorial(j - k) * T ** (j - k)   # p^(k)(T)
    row = np.zeros(nc)
    t = tau * T
    for j in range(d, nc):
        row[j] = math.factorial(j) / math.factorial(j - d) * t ** (j - d)
    g = np.linalg.solve(A.T, row)
    return g[:r], g[r:]


if have_dbg:
    names = ["prologue", "forward chain", "backward + G", "axis set-up", "selection", "direction + ratio", "sweep", "hand-over"]
    st = dump[:16, 3100:3109]
    print("cycles per section (mean over the waves of the first 16 trajectories, %d trajectories in flight): " % n
          + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, st[:, :8].mean(axis=0))) + "; trips %.1f; total %.0f" % (st[:, 8].mean(), st[:, :8].sum(axis=1).mean()))
    print("box part of the starting set == box part of the final set for %d / %d problems" % (int(np.sum(np.all(box == res[2][3][:, :, :2], axis=2))), 3 * n))
    for k in range(min(n, 2)):
        T = b["times"][k]
        H = hessian(r, T)
        I = np.arange(r, M * r)
        Bd = np.r_[np.arange(r), np.arange(M * r, (M + 1) * r)]
        Hi = np.linalg.inv(H[np.ix_(I, I)])
        NC = int(dump[k, 2304 + 640])
        cd = dump[k, 2304 + 576: 2304 + 576 + NC].astype(np.int64)
        kL, kind, seg = cd & 255, (cd >> 8) & 15, (cd >> 12) & 255
        C = np.zeros((NC, (M + 1) * r))                               # functionals over ALL knots 0..M
        for i in range(NC):
            if kind[i] == 0:
                C[i, kL[i] * r] = 1.0
            else:
                gl, gr = hermite_functional(r, T[seg[i]], tau[k * M + seg[i], kind[i] - 1], int(drv[k * M + seg[i], kind[i] - 1]))
                C[i, seg[i] * r:(seg[i] + 1) * r] = gl
                C[i, (seg[i] + 1) * r:(seg[i] + 2) * r] = gr
        Ci = C[:, I]
        Gref = Ci @ Hi @ Ci.T
        G = dump[k, :2304].reshape(48, 48)[:NC, :NC]
        print(f"traj {k}: NC {NC}; max |G - Gref| / max |Gref| = {np.max(np.abs(G - Gref)) / np.max(np.abs(Gref)):.2e}; asymmetry {np.max(np.abs(G - G.T)):.1e}")
        for ax in range(3):
            x0 = np.r_[wp[k, 0, ax], b["bc"][k, 0, :, ax]]; xM = np.r_[wp[k, M, ax], b["bc"][k, 1, :, ax]]
            g = -(H[np.ix_(I, Bd)] @ np.r_[x0, xM])
            yref = Ci @ (Hi @ g)
            y0 = dump[k, 2304 + 192 * ax: 2304 + 192 * ax + NC]
            trips = dump[k, 2304 + 192 * ax + 48]
            mk = [hex(int(v)) for v in dump[k, 2304 + 700 + 8 * ax: 2304 + 700 + 8 * ax + 2 + 2 * K]]
            print(f"   axis {ax}: y0 err {np.max(np.abs(y0 - yref)) / (1 + np.max(np.abs(yref))):.2e} trips {trips:.0f}  prelude set {mk}")
            print(f"           final set   {[hex(int(v)) for v in res[2][3][k, ax]]}  rows-kernel iterations {res[2][2][k]} (box start: {res[1][2][k]})")
