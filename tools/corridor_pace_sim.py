#!/usr/bin/env python3
"""CPU estimate (dense numpy, no GPU): what letting the groups of corridor_dual_kernel run at their own pace is worth (second half).

The kernel advances the 8 trajectories of a wave in lockstep: one trip = one instruction stream for all groups, the three axes one after the
other for the whole wave -- a batch costs  sum_axes max_groups trips(g, axis).  At their own pace a group starts its next axis as soon as its current
one has no violated constraint left -- a batch costs  max_groups sum_axes trips(g, axis)  trips PLUS one masked "advance" block (keep the finished set,
reload both tableau columns from G, reset the state, pick the first entering constraint) every time some group finishes an axis.
Replays the dual method of tools/corridor_dual_probe.py with the kernel's entering rule (violation^2 / T_qq) on config-3 style problems, deals them
eight trajectories to a batch and prints both costs for a few prices of the advance block.

usage: tools/corridor_pace_sim.py [config = 3] [trajectories = 512]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from corridor_strategy_probe import make_problems  # noqa: E402
from corridor_dual_probe import sweep  # noqa: E402


def exchanges(P, max_trips=400):
    n = P.M - 1
    Hinv = np.linalg.inv(P.H)
    T = Hinv[np.ix_(P.pos, P.pos)].copy()
    p = (Hinv @ P.g)[P.pos]
    lo, hi = P.lo, P.hi
    eq = lo == hi
    mu = np.zeros(n); inW = np.zeros(n, bool); up = np.zeros(n, bool)
    q, s, nex = -1, 0.0, 0
    for _ in range(max_trips):
        if q < 0:
            v = np.maximum(lo - p, p - hi)
            tol = 1e-12 * (1 + np.minimum(np.abs(lo), np.abs(hi)))
            dg = np.diag(T)
            cand = ~inW & (v > tol) & (dg > 0)
            if not cand.any():
                return nex
            key = np.where(cand, np.where(eq, 1e300, v * v / np.where(dg > 0, dg, 1.0)), -1.0)
            q = int(np.argmax(key))
            s = 1.0 if p[q] < lo[q] else -1.0
        bq = lo[q] if s > 0 else hi[q]
        z = T[:, q].copy()
        t1 = (bq - p[q]) * s / z[q]
        d = s * z
        blocks = inW & ~eq & np.where(up, d < 0, d > 0)
        ratio = np.where(blocks, np.maximum(mu / np.where(blocks, d, 1.0), 0.0), np.inf)
        i = int(np.argmin(ratio))
        t = min(t1, ratio[i])
        p = np.where(inW, p, p + t * d)
        mu = np.where(inW, mu - t * d, mu)
        mu[q] += s * t
        nex += 1
        if ratio[i] < t1:
            sweep(T, i); inW[i] = False; mu[i] = 0.0
        else:
            sweep(T, q); inW[q] = True; up[q] = s < 0; q = -1
    return nex


cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ntraj = int(sys.argv[2]) if len(sys.argv) > 2 else 512
probs = make_problems(cfg, ntraj)                 # (trajectory, axis) problems, trajectory-major
ex = np.array([exchanges(P) for P in probs])
E = ex[:len(ex) // 24 * 24].reshape(-1, 8, 3)     # batches of 8 trajectories x 3 axes
print(f"config {cfg}: {len(ex)} problems, exchanges mean {ex.mean():.2f} p90 {np.percentile(ex, 90):.0f} max {ex.max()}, none at all {100 * (ex == 0).mean():.1f} %")
sync = E.max(1).sum(1)
own = (E + (E == 0)).sum(2).max(1)               # (an axis without any exchange costs its group one idle trip)
ev = np.array([len(set(np.cumsum(b + (b == 0), axis=1).ravel().tolist())) + 1 for b in E])
print(f"{len(E)} batches: lockstep {sync.mean():.1f} trips per batch, own pace {own.mean():.1f} trips + {ev.mean():.1f} advance blocks")
TRIP, SETUP = 390, 170                            # static instructions of a trip / of the per-axis set-up (docs/measurement_log.md R4.2)
for adv in (240, 150, 80):
    print(f"   advance block of {adv} instructions: own pace {(own * TRIP + ev * adv).mean():.0f} against lockstep {(sync * TRIP + 3 * SETUP).mean():.0f} instructions per batch in the trip phase")
