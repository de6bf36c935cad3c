#!/usr/bin/env python3
"""Debug aid for qp_corridor_dual.h (GPU box, debug build: make -C uav_motion_planning_amd/csrc dual-debug;
UAVQP_LIB_PATH=tools/ubench/libuavqp_dualdbg.so python tools/corridor_dual_gpu_probe.py [cfg]): what the kernel computed for
the first trajectories -- G = [H^-1]_pp, the unconstrained minimisers, trip counts, its starting sets -- against dense numpy
and against the working set the solve ends with."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tools')
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402
from corridor_strategy_probe import Problem  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ctx = U.Context(0)
if cfg == 3:
    r, M = 3, 16
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    uni, mx = M, M
else:
    r = 4
    b = W.ragged_batch(5, n, r)
    uni, mx = 0, 24
so = b["seg_offsets"]
lo, hi = W.corridor_boxes(b, config_index=cfg)
d_so = up(so)
d = {k: up(b[k]) for k in ("waypoints", "times", "bc")}
d_lo, d_hi = up(lo), up(hi)
out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev)
st = torch.zeros(n, dtype=torch.int32, device=dev)
it = torch.zeros(n, dtype=torch.int32, device=dev)
act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
ctx.solve_corridor_device(r, n, uni, mx, None if uni else d_so, d["waypoints"], d["times"], d["bc"], d_lo, d_hi, out, st, it, act, False)
torch.cuda.synchronize()
lib = U.lib()
dump = np.zeros((64, 2048))
guess = np.zeros((n, 3, 2), dtype=np.uint64)
lib.uavqp_debug_corridor_dual.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
rc = lib.uavqp_debug_corridor_dual(ctx._h, dump.ctypes.data, guess.ctypes.data, n)
assert rc == 0, rc
acts = act.cpu().numpy().astype(np.uint64)
itn = it.cpu().numpy()
wp = np.asarray(b["waypoints"]).reshape(-1, 3)
times = np.asarray(b["times"]).reshape(-1)
lo2, hi2 = lo.reshape(-1, 3), hi.reshape(-1, 3)
# ragged batches are dealt longest first: dump slot = dealing position; uniform: index order
order = np.arange(n) if uni else None
print("iterations of the solve kernel: mean %.2f max %d; guess == final set for %d / %d problems" %
      (itn.mean(), itn.max(), int(np.sum(np.all(guess == acts, axis=2))), 3 * n))
for k in range(min(n, 4 if uni else 0)):
    Mk = so[k + 1] - so[k]
    rows = slice(so[k] + k, so[k + 1] + k + 1)
    nn = Mk - 1
    G = dump[k, :1024].reshape(32, 32)[:nn, :nn]
    for ax in range(3):
        w = wp[rows, ax]
        x0 = np.r_[w[0], b["bc"][k, 0, :, ax]]
        xM = np.r_[w[Mk], b["bc"][k, 1, :, ax]]
        P = Problem(r, times[so[k]:so[k + 1]], x0, xM, lo2[rows, ax][1:Mk], hi2[rows, ax][1:Mk], w[1:Mk])
        Hinv = np.linalg.inv(P.H)
        Gref = Hinv[np.ix_(P.pos, P.pos)]
        pref = (Hinv @ P.g)[P.pos]
        if ax == 0:
            print(f"traj {k}: max |G - Gref| / max|Gref| = {np.max(np.abs(G - Gref)) / np.max(np.abs(Gref)):.2e}")
        pu = dump[k, 1024 + 96 * ax: 1024 + 96 * ax + nn]
        trips = dump[k, 1024 + 96 * ax + 32]
        print(f"   axis {ax}: p_unc err {np.max(np.abs(pu - pref)) / (1 + np.max(np.abs(pref))):.2e}  trips {trips:.0f}  guess {int(guess[k, ax, 0]):#x}/{int(guess[k, ax, 1]):#x}"
              f"  final {int(acts[k, ax, 0]):#x}/{int(acts[k, ax, 1]):#x}  solve iterations {itn[k]}")

print("trip log of trajectory 0, axis 0: q, sdir, t1, rmin, partial, pivot, p[col 0], dg[col 0]")
for t in range(12):
    print("  ", dump[0, 1400 + 8 * t: 1408 + 8 * t])
print("p at the end, axis 0:", dump[0, 1024 + 64: 1024 + 64 + 15])
print("lo:", lo2[1:16, 0])
print("hi:", hi2[1:16, 0])

# the same method in numpy on trajectory 0, axis 0
k, ax = 0, 0
Mk = so[1] - so[0]
rows = slice(so[0], so[1] + 1)
w = wp[rows, ax]
P = Problem(r, times[so[0]:so[1]], np.r_[w[0], b["bc"][0, 0, :, ax]], np.r_[w[Mk], b["bc"][0, 1, :, ax]], lo2[rows, ax][1:Mk], hi2[rows, ax][1:Mk], w[1:Mk])
Hinv = np.linalg.inv(P.H)
T = Hinv[np.ix_(P.pos, P.pos)].copy()
pp = (Hinv @ P.g)[P.pos]
nn = Mk - 1
mu = np.zeros(nn); inW = np.zeros(nn, bool); upm = np.zeros(nn, bool)
q = -1
for trip in range(12):
    if q < 0:
        v = np.maximum(P.lo - pp, pp - P.hi)
        key = np.where(inW | (v <= 1e-12), -1.0, v * v / np.diag(T))
        q = int(np.argmax(key))
        if key[q] < 0:
            print("   numpy: done"); break
        sd = 1.0 if pp[q] < P.lo[q] else -1.0
    bq = P.lo[q] if sd > 0 else P.hi[q]
    z = T[:, q].copy()
    t1 = (bq - pp[q]) * sd / z[q]
    dd = sd * z
    blocks = inW & np.where(upm, dd < 0, dd > 0)
    ratio = np.where(blocks, np.maximum(mu / np.where(blocks, dd, 1.0), 0.0), np.inf)
    i = int(np.argmin(ratio)); t2 = ratio[i]; t = min(t1, t2)
    print("   numpy trip", trip, "q", q, "sdir", sd, "t1", t1, "t2", t2, "p0", pp[0], "T00", T[0, 0])
    pp = np.where(inW, pp, pp + t * dd); mu = np.where(inW, mu - t * dd, mu); mu[q] += sd * t
    piv_k = i if t2 < t1 else q
    tcol = T[:, piv_k].copy(); piv = 1.0 / tcol[piv_k]
    T -= np.outer(tcol, tcol) * piv; T[:, piv_k] = tcol * abs(piv); T[piv_k, :] = tcol * abs(piv); T[piv_k, piv_k] = -piv
    if t2 < t1:
        inW[i] = False; mu[i] = 0.0
    else:
        inW[q] = True; upm[q] = sd < 0; pp[q] = bq; q = -1
