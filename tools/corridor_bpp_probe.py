"""CPU probe (dense numpy, no GPU): block principal pivoting (Judice-Pires: full exchanges while the number of infeasibilities
falls, at most p_max exchanges without improvement, then single exchanges of the highest-index infeasible variable until it
falls again) against the current rule of qp_corridor.h, on config-3 / config-5 style problems.  Iterations = block-tridiagonal solves.
python tools/corridor_bpp_probe.py <cfg> <n_traj> [start: free|closed]"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from tools.corridor_strategy_probe import make_problems, strategy_current  # noqa: E402


def infeasible(P, x, lam, pin, upper):
    lo, hi = P.lo, P.hi
    p = x[P.pos]
    below = ~pin & (p < lo - 1e-12 * (1 + np.abs(lo)))
    above = ~pin & ~below & (p > hi + 1e-12 * (1 + np.abs(hi)))
    viol = np.where(upper, lam, -lam)
    wrong = pin & (viol > 1e-13 * max(np.abs(lam).max(), 1e-300)) & (hi > lo)   # an equality row (lo == hi) never leaves
    return below, above, wrong


def strategy_bpp(P, p_max=3, max_iter=300, start="free", single="last"):
    K = P.M - 1
    lo, hi = P.lo, P.hi
    eq = hi <= lo
    pin = eq.copy()
    upper = np.zeros(K, bool)
    if start == "closed":
        # the polynomial through the end states only (no waypoints) ~ here: the unconstrained minimiser with only equalities pinned
        x, lam = P.solve_pinned(pin, np.where(upper, hi, lo))
        p = x[P.pos]
        pin = eq | (p < lo) | (p > hi)
        upper = p > hi
        it = 1
    else:
        it = 0
    best = K + 1
    budget = p_max
    while it < max_iter:
        z = np.where(upper, hi, lo)
        x, lam = P.solve_pinned(pin, z)
        it += 1
        below, above, wrong = infeasible(P, x, lam, pin, upper)
        inf = below | above | wrong
        ninf = int(inf.sum())
        if ninf == 0:
            return it
        if ninf < best:
            best, budget, block = ninf, p_max, True
        elif budget > 0:
            budget -= 1; block = True
        else:
            block = False
        if block:
            pin = (pin & ~wrong) | below | above
            upper = (upper & pin & ~above) | above
        else:
            j = np.nonzero(inf)[0][-1] if single == "last" else np.nonzero(inf)[0][0]
            pin = pin.copy(); upper = upper.copy()
            if wrong[j]:
                pin[j] = False
            else:
                pin[j] = True; upper[j] = above[j]
    return it


if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    probs = make_problems(cfg, n)
    rules = [("current rule (3 pdas + primal active set)", lambda P: strategy_current(P, 3))]
    for pm in (1, 2, 3, 5, 10):
        rules.append((f"bpp p_max={pm} free start", lambda P, pm=pm: strategy_bpp(P, pm)))
    rules.append(("bpp p_max=3 closed start", lambda P: strategy_bpp(P, 3, start="closed")))
    rules.append(("bpp p_max=3 free start, first-index single", lambda P: strategy_bpp(P, 3, single="first")))
    for name, fn in rules:
        its = np.array([fn(P) for P in probs])
        print(f"{name:50s} mean {its.mean():6.2f}  p50 {np.percentile(its, 50):.0f} p90 {np.percentile(its, 90):.0f} p99 {np.percentile(its, 99):.0f} max {its.max():3d}  capped {int((its >= 300).sum())}", flush=True)
