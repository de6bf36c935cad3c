"""CPU probe (dense numpy, no GPU): the position-space DUAL active-set prelude of qp_corridor_dual.h replayed on
config-3 / config-5 style problems.  G = [H^-1]_pp (inverse Hessian restricted to the knot positions) is formed once per
trajectory, the Goldfarb-Idnani dual method then runs on a swept tableau of G: one symmetric sweep (rank-one update of an
n x n matrix, n = M - 1) per constraint that enters or leaves.  Prints exchanges per problem and whether the working set it
ends with is the one the exact primal method of qp_corridor.h ends with.  Design aid, not product, not a test."""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
from corridor_strategy_probe import Problem, make_problems  # noqa: E402


def reference_set(P, max_iter=400):
    """Primal active set of qp_corridor.h (safe phase only, from the clipped waypoints): returns (pin, upper)."""
    K = P.M - 1
    lo, hi = P.lo, P.hi
    z = np.clip(P.wp, lo, hi)
    pin = lo == hi
    upper = np.zeros(K, bool)
    eq = pin.copy()
    for _ in range(max_iter):
        x, lam = P.solve_pinned(pin, z)
        p = x[P.pos]
        below = ~pin & (p < lo - 1e-12 * (1 + np.abs(lo)))
        above = ~pin & ~below & (p > hi + 1e-12 * (1 + np.abs(hi)))
        viol = np.where(upper, lam, -lam)
        wrong = pin & ~eq & (viol > 1e-13 * np.abs(lam).max())
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
                return pin & ~eq, upper & pin & ~eq
            j = np.nonzero(wrong)[0][np.argmax(viol[wrong])]
            pin = pin.copy(); pin[j] = False
    raise RuntimeError("reference did not converge")


def sweep(T, k):
    """Symmetric sweep / reverse sweep on pivot k (the sign of T[k, k] says which): in place."""
    t = T[:, k].copy()
    piv = 1.0 / t[k]
    T -= np.outer(t, t) * piv
    T[:, k] = t * abs(piv)
    T[k, :] = t * abs(piv)
    T[k, k] = -piv


def dual_active_set(P, max_trips=400):
    """Goldfarb-Idnani on the tableau.  Returns (inW, upper, exchanges)."""
    n = P.M - 1
    Hinv = np.linalg.inv(P.H)
    G = Hinv[np.ix_(P.pos, P.pos)]
    p = (Hinv @ P.g)[P.pos]
    lo, hi = P.lo, P.hi
    eq = lo == hi
    T = G.copy()
    mu = np.zeros(n)
    inW = np.zeros(n, bool)
    up = np.zeros(n, bool)
    q = -1
    s = 0.0
    nex = 0
    for _ in range(max_trips):
        if q < 0:
            v = np.maximum(lo - p, p - hi)
            v = np.where(inW, -np.inf, v)
            v = np.where(eq & ~inW & (v > 0), v + 1e30, v)
            q = int(np.argmax(v))
            if v[q] <= 1e-12 * (1 + abs(lo[q] if p[q] < lo[q] else hi[q])):
                return inW & ~eq, up & inW & ~eq, nex
            s = 1.0 if p[q] < lo[q] else -1.0
        bq = lo[q] if s > 0 else hi[q]
        z = T[:, q].copy()
        t1 = (bq - p[q]) * s / z[q]
        d = s * z
        blocks = inW & ~eq & np.where(up, d < 0, d > 0)
        ratio = np.where(blocks, np.maximum(mu / np.where(blocks, d, 1.0), 0.0), np.inf)
        i = int(np.argmin(ratio))
        t2 = ratio[i]
        t = min(t1, t2)
        p = np.where(inW, p, p + t * d)
        mu = np.where(inW, mu - t * d, mu)
        mu[q] += s * t
        nex += 1
        if t2 < t1:
            sweep(T, i)
            inW[i] = False
            mu[i] = 0.0
        else:
            sweep(T, q)
            inW[q] = True
            up[q] = s < 0
            p[q] = bq
            q = -1
    return inW & ~eq, up & inW & ~eq, nex


if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    probs = make_problems(cfg, n)
    ex, same, nact = [], 0, []
    for P in probs:
        rp, ru = reference_set(P)
        dp, du, k = dual_active_set(P)
        ex.append(k)
        nact.append(rp.sum())
        same += np.array_equal(rp, dp) and np.array_equal(ru, du)
    ex = np.array(ex)
    print(f"config {cfg}: {len(probs)} problems, active at the solution mean {np.mean(nact):.2f}; exchanges mean {ex.mean():.2f} "
          f"p90 {np.percentile(ex, 90):.0f} max {ex.max()}; same working set as the primal method: {same}/{len(probs)}")
