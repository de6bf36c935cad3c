#!/usr/bin/env python3
"""Generate exact known-answer vectors for the reference QP (tests/golden/kkt_exact.json).

The reference holds no golden output for this path ("parity unpinned"), and
its solver stack cannot be built or imported here, so these vectors come from exact rational
arithmetic instead: the QP of minimum_control.cpp:5-125 (P, A, l=u assembled entry by entry as the
reference does; r=4 by the same pattern) is solved through its KKT system with Python Fractions --
no floating point until the final rounding to float64.  Inputs are dyadic rationals so that their
float64 representation is exact.  This script is independent of oracle/*.c (it pins the oracle) and of
the device algorithm.

    python tests/golden/gen_golden.py        # rewrites tests/golden/kkt_exact.json (~1 min)
"""
import json
import os
import random
from fractions import Fraction as Fr
from math import factorial


def falling(k, d):
    return Fr(factorial(k), factorial(k - d))


def assemble(r, T, pos, bcs, bce):
    """Dense exact P (n x n), A (m x n), b (m) following minimum_control.cpp:5-125."""
    M, R = len(T), 2 * r
    n, m = R * M, 2 * r + (r + 1) * (M - 1)
    P = [[Fr(0)] * n for _ in range(n)]
    A = [[Fr(0)] * n for _ in range(m)]
    b = [Fr(0)] * m
    for i in range(M):                                                   # getHessian :5-19
        for a in range(r, R):
            for c in range(r, R):
                e = a + c - 2 * r + 1
                P[R * i + a][R * i + c] = falling(a, r) * falling(c, r) * T[i] ** e / e
    for d in range(r):                                                   # start rows :29-31
        A[d][d] = falling(d, d)
    for i in range(M - 1):
        for k in range(R):                                               # waypoint rows :34-42
            A[r + (r + 1) * i][R * i + k] = T[i] ** k
        for d in range(r):                                               # continuity rows :45-74
            row = (r + 1) * (i + 1) + d
            for k in range(d, R):
                A[row][R * i + k] = falling(k, d) * T[i] ** (k - d)
            A[row][R * (i + 1) + d] = -falling(d, d)
    i = M - 1                                                            # end rows :77-95
    for k in range(R):
        A[r + (r + 1) * i][R * i + k] = T[i] ** k
    for d in range(1, r):
        for k in range(d, R):
            A[(r + 1) * M + d - 1][R * i + k] = falling(k, d) * T[i] ** (k - d)
    b[0] = pos[0]                                                        # getBound :98-125
    for d in range(1, r):
        b[d] = bcs[d - 1]
    b[r + (r + 1) * (M - 1)] = pos[M]
    for d in range(1, r):
        b[r + (r + 1) * (M - 1) + d] = bce[d - 1]
    for i in range(M - 1):
        b[r + (r + 1) * i] = pos[i + 1]
    return P, A, b


def solve_kkt(P, A, b):
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
    for c in range(N):                                                   # Gauss-Jordan, exact
        p = next(i for i in range(c, N) if K[i][c] != 0)
        K[c], K[p] = K[p], K[c]
        inv = 1 / K[c][c]
        K[c] = [v * inv for v in K[c]]
        for i in range(N):
            if i != c and K[i][c] != 0:
                f = K[i][c]
                K[i] = [vi - f * vc for vi, vc in zip(K[i], K[c])]
    return [K[i][N] for i in range(n)]


def dyadic(rng, lo, hi, den):
    return Fr(rng.randint(int(lo * den), int(hi * den)), den)


def make_case(name, r, M, rng, t_lo=0.5, t_hi=2.0, unit_time=False, fixed=None):
    if fixed:
        pos3, T, bc = fixed
    else:
        T = [Fr(1)] * M if unit_time else [dyadic(rng, t_lo, t_hi, 8) for _ in range(M)]
        pos3 = [[dyadic(rng, -4, 4, 16) for _ in range(M + 1)] for _ in range(3)]
        bc = [[[dyadic(rng, -1, 1, 8) for _ in range(3)] for _ in range(r - 1)] for _ in range(2)]  # [end][d][axis]
    coef, cost = [], []
    for ax in range(3):
        bcs = [bc[0][d][ax] for d in range(r - 1)]
        bce = [bc[1][d][ax] for d in range(r - 1)]
        P, A, b = assemble(r, T, pos3[ax], bcs, bce)
        x = solve_kkt(P, A, b)
        coef.append([float(v) for v in x])
        cost.append(float(sum(x[i] * P[i][j] * x[j] for i in range(len(x)) for j in range(len(x)) if P[i][j] != 0) / 2))
    return dict(name=name, r=r, M=M, times=[float(t) for t in T],
                waypoints=[[float(pos3[ax][k]) for ax in range(3)] for k in range(M + 1)],
                bc=[[[float(bc[e][d][ax]) for ax in range(3)] for d in range(r - 1)] for e in range(2)],
                coef=coef, half_xPx=cost)


def main():
    rng = random.Random(20260925)
    cases = []
    # the reference's only fixed input, test_qpsolve.cpp:10-17 (x axis; y, z: same waypoints scaled)
    kat_pos = [[Fr(1), Fr(2), Fr(3), Fr(4)], [Fr(2), Fr(4), Fr(6), Fr(8)], [Fr(-1), Fr(-2), Fr(-3), Fr(-4)]]
    kat_bc = [[[Fr(0)] * 3 for _ in range(2)] for _ in range(2)]
    cases.append(make_case("test_qpsolve", 3, 3, rng, fixed=(kat_pos, [Fr(1)] * 3, kat_bc)))
    for name, r, M, kw in [
        ("jerk_M1", 3, 1, {}), ("jerk_M2", 3, 2, {}), ("jerk_M5_unitT", 3, 5, dict(unit_time=True)),
        ("jerk_M16", 3, 16, {}), ("snap_M1", 4, 1, {}), ("snap_M2", 4, 2, {}), ("snap_M3", 4, 3, {}),
        ("snap_M7_config1", 4, 7, {}), ("snap_M8_config2", 4, 8, {}), ("snap_M8_unitT", 4, 8, dict(unit_time=True)),
        ("snap_M8_wideT", 4, 8, dict(t_lo=0.25, t_hi=4.0)), ("snap_M12", 4, 12, {}),
    ]:
        cases.append(make_case(name, r, M, rng, **kw))
        print(name, "done", flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kkt_exact.json")
    with open(out, "w") as f:
        json.dump(dict(generator="tests/golden/gen_golden.py", layout="coef[axis][2r*seg + k], ascending powers",
                       cases=cases), f, indent=0)
    print("wrote", out)


if __name__ == "__main__":
    main()
