#!/usr/bin/env python3
"""Exact known-answer vectors for the QP with GENERAL inequality rows (tests/golden/rows_exact.json): knot boxes plus rows
lo <= p_i^(d)(tau T_i) <= hi at in-segment times (uavqp_solve_rows_batch_device, include/uavqp.h).

No reference implementation exists for such rows (the reference only builds equality rows, minimum_control.cpp:98-125), so the
fixtures are made like corridor_exact.json: exact rational arithmetic on the reference-formulation matrices
(gen_golden.assemble = minimum_control.cpp:5-125 entry by entry) with the extra rows appended as monomial rows on their
segment's coefficients.  Enumerating all 3^(boxes + rows) assignments is too slow here, and not needed: the QP is strictly convex,
so ANY point that satisfies the KKT conditions exactly IS the unique minimiser.  A float64 dense active-set solve (numpy, written
here, sharing nothing with oracle/*.c or the device code) proposes the assignment {free, at lower, at upper} of every inequality;
the equality-constrained KKT system of that assignment is then solved in Python Fractions and the KKT conditions -- primal
feasibility of the free rows, sign of the multipliers of the active ones -- are CHECKED EXACTLY.  A proposal that fails the exact
check aborts the script.

    python tests/golden/gen_golden_rows.py      # rewrites tests/golden/rows_exact.json (about a minute)
"""
import json
import os
import random
import sys
from fractions import Fraction as Fr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import assemble, dyadic, falling  # noqa: E402
from gen_golden_corridor import solve_kkt_with_multipliers  # noqa: E402


def mono_row(r, M, seg, t, d):
    a = [Fr(0)] * (2 * r * M)
    for k in range(d, 2 * r):
        a[2 * r * seg + k] = falling(k, d) * t ** (k - d)
    return a


def propose_states(P, A, lo, hi):
    """float64 dual active set on the dense data: returns state[i] in {0, -1, +1} for every row of A (equalities: -1)."""
    Pn = np.array(P, dtype=float)
    An = np.array(A, dtype=float)
    l, u = np.array(lo, dtype=float), np.array(hi, dtype=float)
    n, m = Pn.shape[0], An.shape[0]
    eq = l == u
    act, side = eq.copy(), np.where(eq, -1, 0)
    lam_cur = np.zeros(m)
    newp = -1

    def solve():
        idx = np.nonzero(act)[0]
        K = np.zeros((n + len(idx), n + len(idx)))
        K[:n, :n] = Pn
        K[:n, n:] = An[idx].T
        K[n:, :n] = An[idx]
        sol = np.linalg.solve(K, np.r_[np.zeros(n), np.where(side[idx] > 0, u[idx], l[idx])])
        mu = np.zeros(m)
        mu[idx] = sol[n:]
        return sol[:n], mu

    for _ in range(400):
        x, mu = solve()
        bad = act & ~eq & (np.where(side > 0, -mu, mu) > 1e-11 * max(1.0, np.abs(mu).max()))
        tmin, drop = 2.0, -1
        for i in np.nonzero(bad)[0]:
            if i == newp:
                continue
            t = lam_cur[i] / (lam_cur[i] - mu[i]) if lam_cur[i] != mu[i] else 0.0
            t = min(max(t, 0.0), 1.0)
            if t < tmin:
                tmin, drop = t, i
        if drop >= 0:
            lam_cur = np.where(act, lam_cur + tmin * (mu - lam_cur), 0.0)
            act[drop] = False
            side[drop] = 0
            lam_cur[drop] = 0.0
            continue
        lam_cur = np.where(act, mu, 0.0)
        newp = -1
        v = An @ x
        viol = np.maximum(l - v, v - u)
        viol[act] = -1.0
        p = int(np.argmax(viol))
        if viol[p] <= 1e-11:
            return [int(s) for s in side]
        act[p] = True
        side[p] = 1 if v[p] > u[p] else -1
        newp = p
    raise RuntimeError("float active set did not terminate (infeasible draw?)")


def solve_axis(r, T, pos, bcs, bce, box_lo, box_hi, rows):
    """rows: list of (segment, tau, d, lo, hi) in Fractions.  Returns (x, box_state [M-1], row_state [len(rows)], P)."""
    M = len(T)
    P, A, b = assemble(r, T, pos, bcs, bce)
    lo, hi = list(b), list(b)
    wrows = [r + (r + 1) * i for i in range(M - 1)]
    for j, row in enumerate(wrows):
        lo[row], hi[row] = box_lo[j], box_hi[j]
    base = len(A)
    for (s, tau, d, l, h) in rows:
        A.append(mono_row(r, M, s, tau * T[s], d))
        lo.append(l)
        hi.append(h)
    state = propose_states(P, A, lo, hi)
    keep = [i for i in range(len(A)) if state[i] != 0]
    x, nu = solve_kkt_with_multipliers(P, [A[i] for i in keep], [hi[i] if state[i] > 0 else lo[i] for i in keep])
    # ---- the exact KKT check that makes the fixture independent of how the assignment was found
    for i in range(len(A)):
        v = sum(A[i][c] * x[c] for c in range(len(x)) if A[i][c] != 0)
        if state[i] == 0:
            assert lo[i] <= v <= hi[i], "proposal is not primal feasible in exact arithmetic"
        else:
            assert v == (hi[i] if state[i] > 0 else lo[i])
            if lo[i] != hi[i]:
                mult = nu[keep.index(i)]          # P x + A' nu = 0: lower active needs nu <= 0, upper nu >= 0
                assert (mult <= 0) if state[i] < 0 else (mult >= 0), "proposal is not dual feasible in exact arithmetic"
    box_state = [0 if lo[row] == hi[row] else state[row] for row in wrows]
    row_state = [0 if rows[e][3] == rows[e][4] else state[base + e] for e in range(len(rows))]
    return x, box_state, row_state, P


def make_case(name, r, M, K, rng, h_box=(2, 8), h_mid=(1, 3), v_lim=(20, 36), taus=(Fr(1, 2), Fr(1, 4)), derivs=(0, 1)):
    T = [dyadic(rng, 0.5, 2.0, 8) for _ in range(M)]
    pos3 = [[dyadic(rng, -3, 3, 16) for _ in range(M + 1)] for _ in range(3)]
    bc = [[[dyadic(rng, -1, 1, 8) for _ in range(3)] for _ in range(r - 1)] for _ in range(2)]
    half = [[Fr(rng.randint(*h_box), 8) for _ in range(M + 1)] for _ in range(3)]
    row_tau = [[taus[j] for j in range(K)] for _ in range(M)]
    row_d = [[derivs[j] for j in range(K)] for _ in range(M)]
    row_lo = [[[None] * 3 for _ in range(K)] for _ in range(M)]
    row_hi = [[[None] * 3 for _ in range(K)] for _ in range(M)]
    coef, box_states, row_states, cost = [], [], [], []
    for ax in range(3):
        rows = []
        for s in range(M):
            for j in range(K):
                if row_d[s][j] == 0:      # position sample: inside the interpolated waypoints +- h
                    c = (1 - row_tau[s][j]) * pos3[ax][s] + row_tau[s][j] * pos3[ax][s + 1]
                    h = Fr(rng.randint(*h_mid), 8)
                    l, u = c - h, c + h
                else:                     # derivative limit, symmetric
                    u = Fr(rng.randint(*v_lim), 8)
                    l = -u
                row_lo[s][j][ax], row_hi[s][j][ax] = l, u
                rows.append((s, row_tau[s][j], row_d[s][j], l, u))
        lo = [pos3[ax][k] - half[ax][k] for k in range(1, M)]
        hi = [pos3[ax][k] + half[ax][k] for k in range(1, M)]
        x, bs, rs, P = solve_axis(r, T, pos3[ax], [bc[0][d][ax] for d in range(r - 1)], [bc[1][d][ax] for d in range(r - 1)], lo, hi, rows)
        coef.append([float(v) for v in x])
        box_states.append(bs)
        row_states.append([[rs[s * K + j] for j in range(K)] for s in range(M)])
        cost.append(float(sum(x[i] * P[i][j] * x[j] for i in range(len(x)) for j in range(len(x)) if P[i][j] != 0) / 2))
    return dict(name=name, r=r, M=M, K=K, times=[float(t) for t in T],
                waypoints=[[float(pos3[ax][k]) for ax in range(3)] for k in range(M + 1)],
                half_width=[[float(half[ax][k]) for ax in range(3)] for k in range(M + 1)],
                bc=[[[float(bc[e][d][ax]) for ax in range(3)] for d in range(r - 1)] for e in range(2)],
                row_tau=[[float(t) for t in seg] for seg in row_tau], row_deriv=row_d,
                row_lo=[[[float(v) for v in slot] for slot in seg] for seg in row_lo],
                row_hi=[[[float(v) for v in slot] for slot in seg] for seg in row_hi],
                coef=coef, box_active=box_states, row_active=row_states, half_xPx=cost)


def main():
    rng = random.Random(20260925 + 4)
    cases = []
    for name, r, M, K, kw in [
        ("jerk_M2_pos_sample", 3, 2, 1, {}),
        ("jerk_M4_pos_and_vel", 3, 4, 2, {}),
        ("jerk_M5_tight_samples", 3, 5, 2, dict(h_mid=(1, 1), v_lim=(14, 20))),
        ("jerk_M3_acc_limit", 3, 3, 1, dict(derivs=(2,), v_lim=(8, 24), taus=(Fr(1, 2),))),
        ("snap_M3_pos_and_vel", 4, 3, 2, {}),
        ("snap_M4_vel_quarter", 4, 4, 1, dict(derivs=(1,), taus=(Fr(1, 4),), v_lim=(12, 20))),
        ("snap_M5_pos_sample", 4, 5, 1, dict(h_mid=(1, 2))),
        ("jerk_M6_mixed", 3, 6, 2, dict(h_box=(1, 3), h_mid=(2, 6), v_lim=(16, 28))),
    ]:
        for attempt in range(20):
            try:
                cases.append(make_case(name, r, M, K, rng, **kw))
                break
            except RuntimeError:
                continue          # an infeasible draw: next one
        else:
            raise SystemExit(f"no feasible draw for {name}")
        c = cases[-1]
        print(name, "boxes:", c["box_active"], "rows:", c["row_active"], flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rows_exact.json")
    with open(out, "w") as f:
        json.dump(dict(generator="tests/golden/gen_golden_rows.py",
                       layout="coef[axis][2r*seg + k], ascending powers; knot box k: waypoints[k] +- half_width[k]; row (segment i, slot j): "
                              "row_lo <= p_i^(row_deriv)(row_tau T_i) <= row_hi per axis; box_active[axis][k-1], row_active[axis][i][j] in "
                              "{0 free, -1 at lower, +1 at upper}",
                       cases=cases), f, indent=0)
    print("wrote", out)


if __name__ == "__main__":
    main()
