"""CPU probe (dense numpy, no GPU): iteration counts of active-set strategies for the corridor QP in the Hermite
variables (the per-axis problem of qp_corridor.h) on config-3 / config-5 style problems.  A design aid: each
"iteration" = one block-tridiagonal solve on the device.  Not part of the product or of the tests."""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from uav_motion_planning_amd import workloads as W  # noqa: E402

HW = {3: np.array([[720., -360., 60.], [-360., 192., -36.], [60., -36., 9.]]),
      4: np.array([[100800., -50400., 10080., -840.], [-50400., 25920., -5400., 480.], [10080., -5400., 1200., -120.], [-840., 480., -120., 16.]])}
HV = {3: np.array([[720., -360., 60.], [360., -168., 24.], [60., -24., 3.]]),
      4: np.array([[100800., -50400., 10080., -840.], [50400., -24480., 4680., -360.], [10080., -4680., 840., -60.], [840., -360., 60., -4.]])}


def hessian(r, T):
    """Full (M+1) r x (M+1) r Hessian over knots 0..M."""
    M = len(T)
    H = np.zeros(((M + 1) * r, (M + 1) * r))
    a = np.arange(r)
    sgn = (-1.0) ** (a[:, None] + a[None, :])
    for s in range(M):
        p = T[s] ** -(2 * r - 1 - a[:, None] - a[None, :]).astype(float)
        B11 = p * HW[r]
        B01 = -p * HV[r]
        B00 = sgn * B11
        i0, i1 = s * r, (s + 1) * r
        H[i0:i0 + r, i0:i0 + r] += B00
        H[i1:i1 + r, i1:i1 + r] += B11
        H[i0:i0 + r, i1:i1 + r] += B01
        H[i1:i1 + r, i0:i0 + r] += B01.T
    return H


class Problem:
    def __init__(self, r, T, x0, xM, lo, hi, wp):
        M = len(T)
        self.r, self.M = r, M
        H = hessian(r, T)
        I = np.arange(r, M * r)
        Bd = np.r_[np.arange(r), np.arange(M * r, (M + 1) * r)]
        self.H = H[np.ix_(I, I)]
        self.g = -(H[np.ix_(I, Bd)] @ np.r_[x0, xM])     # minimise 1/2 x'Hx - g'x
        self.pos = np.arange(0, (M - 1) * r, r)            # indices of the position components
        self.lo, self.hi, self.wp = lo, hi, wp
        self.nsolve = 0

    def solve_pinned(self, pin, z):
        """Minimiser with positions `pin` (bool [M-1]) fixed at z; returns x and the multipliers of all positions."""
        self.nsolve += 1
        n = self.H.shape[0]
        fixed = np.zeros(n, bool)
        fixed[self.pos[pin]] = True
        x = np.zeros(n)
        x[fixed] = z[pin]
        free = ~fixed
        rhs = self.g[free] - self.H[np.ix_(free, fixed)] @ x[fixed]
        x[free] = np.linalg.solve(self.H[np.ix_(free, free)], rhs)
        lam = (self.H @ x - self.g)[self.pos]              # d cost / d p_k
        return x, lam


def strategy_current(P, pdas_rounds=3, max_iter=200):
    """qp_corridor.h as of round 1: all free -> PDAS rounds -> safe primal active set (one change per solve)."""
    K = P.M - 1
    lo, hi = P.lo, P.hi
    z = np.clip(P.wp, lo, hi)
    pin = np.zeros(K, bool)
    upper = np.zeros(K, bool)
    it = 0
    pdas = pdas_rounds
    while it < max_iter:
        x, lam = P.solve_pinned(pin, z)
        p = x[P.pos]
        it += 1
        tol = 1e-12 * (1 + np.abs(lo))
        below = ~pin & (p < lo - tol)
        above = ~pin & ~below & (p > hi + 1e-12 * (1 + np.abs(hi)))
        viol = np.where(upper, lam, -lam)
        wrong = pin & (viol > 1e-13 * np.abs(lam).max())
        if pdas > 0:
            npin = (pin & ~wrong) | below | above
            nupper = (upper & pin & ~wrong) | above
            if np.array_equal(npin, pin) and np.array_equal(nupper & npin, upper & pin):
                return it
            pdas -= 1
            z = np.where(npin, np.where(nupper, hi, lo), np.clip(p, lo, hi))
            pin, upper = npin, nupper
            continue
        if below.any() or above.any():
            cand = np.nonzero(below | above)[0]
            al = ((np.where(above, hi, lo) - z) / (p - z))[cand]
            j = cand[np.argmin(al)]
            a = max(0.0, al.min())
            znew = z + a * (p - z)
            znew[pin] = z[pin]
            znew[j] = hi[j] if above[j] else lo[j]
            z = znew
            pin = pin.copy(); pin[j] = True
            upper = upper.copy(); upper[j] = above[j]
        else:
            z = np.where(pin, z, np.clip(p, lo, hi))
            if not wrong.any():
                return it
            j = np.nonzero(wrong)[0][np.argmax(viol[wrong])]
            pin = pin.copy(); pin[j] = False
    return it


def strategy_all_active(P, pdas_rounds=3, max_iter=200, release_all=False):
    """Start from the EQUALITY solve (all positions pinned at the waypoints), guess every bound active on the side the
    multiplier points to, then PDAS rounds + safe phase as before."""
    K = P.M - 1
    lo, hi = P.lo, P.hi
    z = np.clip(P.wp, lo, hi)
    pin = np.ones(K, bool)
    x, lam = P.solve_pinned(pin, z)
    it = 1
    upper = lam < 0          # cost decreases when p grows -> ends at the upper face
    z = np.where(upper, hi, lo)
    pdas = pdas_rounds
    while it < max_iter:
        x, lam = P.solve_pinned(pin, z)
        p = x[P.pos]
        it += 1
        below = ~pin & (p < lo - 1e-12 * (1 + np.abs(lo)))
        above = ~pin & ~below & (p > hi + 1e-12 * (1 + np.abs(hi)))
        viol = np.where(upper, lam, -lam)
        wrong = pin & (viol > 1e-13 * np.abs(lam).max())
        if pdas > 0:
            npin = (pin & ~wrong) | below | above
            nupper = (upper & pin & ~wrong) | above
            if np.array_equal(npin, pin) and np.array_equal(nupper & npin, upper & pin):
                return it
            pdas -= 1
            z = np.where(npin, np.where(nupper, hi, lo), np.clip(p, lo, hi))
            pin, upper = npin, nupper
            continue
        if below.any() or above.any():
            cand = np.nonzero(below | above)[0]
            al = ((np.where(above, hi, lo) - z) / (p - z))[cand]
            j = cand[np.argmin(al)]
            a = max(0.0, al.min())
            znew = z + a * (p - z)
            znew[pin] = z[pin]
            znew[j] = hi[j] if above[j] else lo[j]
            z = znew
            pin = pin.copy(); pin[j] = True
            upper = upper.copy(); upper[j] = above[j]
        else:
            z = np.where(pin, z, np.clip(p, lo, hi))
            if not wrong.any():
                return it
            if release_all:
                pin = pin & ~wrong
            else:
                j = np.nonzero(wrong)[0][np.argmax(viol[wrong])]
                pin = pin.copy(); pin[j] = False
    return it


def make_problems(cfg, n, seed_off=0):
    out = []
    if cfg == 3:
        r, M = 3, 16
        b = W.uniform_batch(3, n, M, r, time_mode="distance")
        lo, hi = W.corridor_boxes(b, config_index=3)
        for k in range(n):
            for ax in range(3):
                x0 = np.r_[b["waypoints"][k, 0, ax], b["bc"][k, 0, :, ax]]
                xM = np.r_[b["waypoints"][k, M, ax], b["bc"][k, 1, :, ax]]
                out.append(Problem(r, b["times"][k], x0, xM, lo[k, 1:M, ax], hi[k, 1:M, ax], b["waypoints"][k, 1:M, ax]))
    else:
        r = 4
        b = W.ragged_batch(5, n, r)
        lo, hi = W.corridor_boxes(b, config_index=5)
        so = b["seg_offsets"]
        wp = np.asarray(b["waypoints"])
        for k in range(n):
            M = so[k + 1] - so[k]
            rows = slice(so[k] + k, so[k + 1] + k + 1)
            for ax in range(3):
                w = wp[rows, ax]
                x0 = np.r_[w[0], b["bc"][k, 0, :, ax]]
                xM = np.r_[w[M], b["bc"][k, 1, :, ax]]
                out.append(Problem(r, b["times"][so[k]:so[k + 1]], x0, xM, lo[rows, ax][1:M], hi[rows, ax][1:M], w[1:M]))
    return out


if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    for name, fn in [("current(3 pdas)", lambda P: strategy_current(P, 3)),
                     ("all-active start, 3 pdas", lambda P: strategy_all_active(P, 3)),
                     ("all-active start, 6 pdas", lambda P: strategy_all_active(P, 6)),
                     ("all-active start, 3 pdas, release all", lambda P: strategy_all_active(P, 3, release_all=True)),
                     ("all-active start, 0 pdas", lambda P: strategy_all_active(P, 0)),
                     ]:
        its = np.array([fn(P) for P in make_problems(cfg, n)])
        w = its[: (len(its) // 64) * 64].reshape(-1, 64).max(axis=1) if len(its) >= 64 else its
        print(f"{name:45s} mean {its.mean():6.2f}  max {its.max():3d}  mean-of-wave-max(64) {w.mean():6.2f}  p90 {np.percentile(its, 90):.0f}")


def simulate_refill(its, slots=32, thresh=8, c_refill=0.5, n_waves=1024):
    """Persistent waves with `slots` problems in flight each; a wave-iteration costs 1, a refill event (emission of the
    finished problems + set-up of the next ones, executed by part of the wave) costs c_refill and happens when at least
    `thresh` slots are done (or nothing is left to run).  Problems come from one global queue in index order.
    Returns wall time in wave-iterations (max over waves) and the no-refill reference (static assignment, wave max)."""
    its = list(its)
    nxt = 0
    n = len(its)
    # round-robin event simulation: waves advance in lock step (good enough: all waves have the same speed)
    rem = np.zeros((n_waves, slots), dtype=int)
    for w in range(n_waves):
        for s in range(slots):
            if nxt < n:
                rem[w, s] = its[nxt]; nxt += 1
    t = np.zeros(n_waves)
    alive = np.ones(n_waves, bool)
    while alive.any():
        for w in np.nonzero(alive)[0]:
            if (rem[w] > 0).any():
                rem[w][rem[w] > 0] -= 1
                t[w] += 1.0
            done = rem[w] == 0
            if done.sum() >= thresh or not (rem[w] > 0).any():
                if nxt < n:
                    for s in np.nonzero(done)[0]:
                        if nxt < n:
                            rem[w, s] = its[nxt]; nxt += 1
                    t[w] += c_refill
                elif not (rem[w] > 0).any():
                    t[w] += c_refill
                    alive[w] = False
    return t.max()
