#!/usr/bin/env python3
"""Exact known-answer vectors for the corridor-constrained QP (tests/golden/corridor_exact.json).

The corridor extension (BASELINE configs 3 / 5) has no reference implementation at all, so its fixtures are made
the same way as kkt_exact.json: exact rational arithmetic on the reference-formulation matrices
(gen_golden.assemble = minimum_control.cpp:5-125 entry by entry), with the interior-waypoint rows turned into
lo <= a_i x <= hi.  The minimiser of a strictly convex QP is the unique point satisfying the KKT conditions, so it
is found by ENUMERATING every assignment {free, at lower, at upper} of the M-1 box rows, solving each equality-
constrained KKT system with Python Fractions and keeping the one assignment that is primal feasible with
multipliers of the right sign -- no active-set strategy, nothing shared with the device algorithm or oracle/*.c.

    python tests/golden/gen_golden_corridor.py      # rewrites tests/golden/corridor_exact.json (a few minutes)
"""
import itertools
import json
import os
import random
import sys
from fractions import Fraction as Fr

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import assemble, dyadic  # noqa: E402


def solve_kkt_with_multipliers(P, A, b):
    n, m = len(P), len(A)
    N = n + m
    K = [[Fr(0)] * (N + 1) for _ in range(N)]
    for i in range(n):
        for j in range(n):
            K[i][j] = P[i][j]
    for i in range(m):
        for j in range(n):
            K[n + i][j] = A[i][j]
            K[j][n + i] = A[i][j]
        K[n + i][N] = b[i]
    for c in range(N):
        p = next(i for i in range(c, N) if K[i][c] != 0)
        K[c], K[p] = K[p], K[c]
        inv = 1 / K[c][c]
        K[c] = [v * inv for v in K[c]]
        for i in range(N):
            if i != c and K[i][c] != 0:
                f = K[i][c]
                K[i] = [vi - f * vc for vi, vc in zip(K[i], K[c])]
    return [K[i][N] for i in range(n)], [K[n + i][N] for i in range(m)]


def solve_axis(r, T, pos, bcs, bce, lo, hi):
    """Returns (x, state) with state[i] in {0 free, -1 at lower, +1 at upper} for interior waypoint i+1."""
    M = len(T)
    P, A, b = assemble(r, T, pos, bcs, bce)
    rows = [r + (r + 1) * i for i in range(M - 1)]
    found = None
    for state in itertools.product((0, -1, 1), repeat=M - 1):
        if any(s == 1 and lo[i] == hi[i] for i, s in enumerate(state)):
            continue                                   # degenerate box: represent it as "at lower" only
        if any(s == 0 and lo[i] == hi[i] for i, s in enumerate(state)):
            continue
        keep = [i for i in range(len(A)) if i not in rows or state[rows.index(i)] != 0]
        Ae = [A[i] for i in keep]
        be = []
        for i in keep:
            if i in rows:
                j = rows.index(i)
                be.append(lo[j] if state[j] < 0 else hi[j])
            else:
                be.append(b[i])
        x, nu = solve_kkt_with_multipliers(P, Ae, be)
        ok = True
        for j, row in enumerate(rows):
            if state[j] == 0:
                v = sum(A[row][c] * x[c] for c in range(len(x)) if A[row][c] != 0)
                ok = ok and lo[j] <= v <= hi[j]
            elif lo[j] != hi[j]:
                mult = nu[keep.index(row)]          # P x + A' nu = 0: lower active needs nu <= 0, upper nu >= 0
                ok = ok and (mult <= 0 if state[j] < 0 else mult >= 0)
            if not ok:
                break
        if ok:
            assert found is None or found[0] == x, "two KKT points: impossible for a strictly convex QP"
            if found is None:
                found = (x, state)
    assert found is not None
    return found, P


def make_case(name, r, M, rng, widths=(1, 6), t_lo=0.5, t_hi=2.0, pin_one=False):
    T = [dyadic(rng, t_lo, t_hi, 8) for _ in range(M)]
    pos3 = [[dyadic(rng, -3, 3, 16) for _ in range(M + 1)] for _ in range(3)]
    bc = [[[dyadic(rng, -1, 1, 8) for _ in range(3)] for _ in range(r - 1)] for _ in range(2)]
    half = [[Fr(rng.randint(*widths), 8) for _ in range(M + 1)] for _ in range(3)]   # box half-widths 1/8 .. 6/8
    if pin_one and M > 2:
        for ax in range(3):
            half[ax][1 + ax % (M - 1)] = Fr(0)                                       # lo == hi: the reference's equality row
    coef, states, cost = [], [], []
    for ax in range(3):
        lo = [pos3[ax][k] - half[ax][k] for k in range(1, M)]
        hi = [pos3[ax][k] + half[ax][k] for k in range(1, M)]
        (x, state), P = solve_axis(r, T, pos3[ax], [bc[0][d][ax] for d in range(r - 1)], [bc[1][d][ax] for d in range(r - 1)], lo, hi)
        coef.append([float(v) for v in x])
        states.append(list(state))
        cost.append(float(sum(x[i] * P[i][j] * x[j] for i in range(len(x)) for j in range(len(x)) if P[i][j] != 0) / 2))
    return dict(name=name, r=r, M=M, times=[float(t) for t in T],
                waypoints=[[float(pos3[ax][k]) for ax in range(3)] for k in range(M + 1)],
                half_width=[[float(half[ax][k]) for ax in range(3)] for k in range(M + 1)],
                bc=[[[float(bc[e][d][ax]) for ax in range(3)] for d in range(r - 1)] for e in range(2)],
                coef=coef, active=states, half_xPx=cost)


def main():
    rng = random.Random(20260925 + 3)
    cases = []
    for name, r, M, kw in [
        ("jerk_M2_box", 3, 2, {}), ("jerk_M4_box", 3, 4, {}), ("jerk_M4_pinned_row", 3, 4, dict(pin_one=True)),
        ("jerk_M6_box", 3, 6, {}), ("snap_M3_box", 4, 3, {}), ("snap_M5_box", 4, 5, {}),
        ("snap_M5_tight", 4, 5, dict(widths=(1, 2))), ("snap_M4_wide_open", 4, 4, dict(widths=(40, 48))),
        ("jerk_M5_mixed", 3, 5, dict(widths=(6, 20))), ("snap_M4_mixed", 4, 4, dict(widths=(6, 20))),
    ]:
        cases.append(make_case(name, r, M, rng, **kw))
        print(name, "active:", cases[-1]["active"], flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corridor_exact.json")
    with open(out, "w") as f:
        json.dump(dict(generator="tests/golden/gen_golden_corridor.py",
                       layout="coef[axis][2r*seg + k], ascending powers; box of interior waypoint k: waypoints[k] +- half_width[k]; "
                              "active[axis][k-1] in {0 free, -1 at lower, +1 at upper}",
                       cases=cases), f, indent=0)
    print("wrote", out)


if __name__ == "__main__":
    main()
