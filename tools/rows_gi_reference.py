#!/usr/bin/env python3
"""CPU model (test infrastructure, numpy): the Goldfarb-Idnani dual active-set method on ONE (trajectory, axis) general-rows QP in dense
form, WITH the step the device solvers lacked until round 5 -- the constraint that is to enter is linearly dependent on the working set:
no primal step, a dual step along the dependency until a multiplier of the set reaches zero (that constraint leaves) or, if none does,
a Farkas certificate of infeasibility.  The QP is the reference formulation (oracle.assemble = minimum_control.cpp:5-96; rows appended as
monomial rows), reduced to the null space of its equality rows.

  python tools/rows_gi_reference.py <seed>     classifies the trajectories listed in gpurun_out/soak_rows_unsolved_seed<seed>.npy
                                               (written by tools/soak_rows.py on the GPU box): feasible / infeasible, certificate margin"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402


def mono_row(r, M, seg, t, d):
    a = np.zeros(2 * r * M)
    f = lambda k: float(np.prod(np.arange(k - d + 1, k + 1))) if d > 0 else 1.0
    for k in range(d, 2 * r):
        a[2 * r * seg + k] = f(k) * t ** (k - d)
    return a


def gi_dense(H, g, C, lo, hi, max_iter=500, dep_tol=1e-9):
    """min 1/2 y'Hy + g'y  s.t. lo <= C y <= hi (H positive definite).  Returns dict(status, y, active, farkas_margin).
    status: 'solved' | 'infeasible' | 'cap'."""
    m = C.shape[0]
    Hi = np.linalg.inv(H)
    y = -Hi @ g
    W = []            # list of (row, sign): sign +1 = at lower bound (c'y >= lo), -1 = at upper (-c'y >= -hi)
    u = []            # multipliers >= 0
    for it in range(max_iter):
        v = C @ y
        below, above = lo - v, v - hi
        viol = np.maximum(below, above) / (1.0 + np.abs(np.where(below > above, lo, hi)))
        inW = {i for i, _ in W}
        for i in inW:
            viol[i] = -1.0
        p = int(np.argmax(viol))
        if viol[p] <= 1e-10:
            return dict(status="solved", y=y, active=W, iters=it)
        sp = 1.0 if below[p] > above[p] else -1.0
        n_p = sp * C[p]
        b_p = sp * (lo[p] if sp > 0 else hi[p])
        u_p = 0.0
        while True:
            if W:
                N = np.array([s * C[i] for i, s in W]).T            # columns = normals
                G = N.T @ Hi @ N
                rr = np.linalg.solve(G, N.T @ Hi @ n_p)
                z = Hi @ (n_p - N @ rr)
            else:
                rr = np.zeros(0)
                z = Hi @ n_p
            znp = float(z @ n_p)
            dependent = znp <= dep_tol * float(n_p @ Hi @ n_p)
            eqs = np.array([lo[i] == hi[i] for i, _ in W], dtype=bool) if W else np.zeros(0, dtype=bool)
            t1, k1 = np.inf, -1
            for j in range(len(W)):
                if rr[j] > 1e-13 * max(1.0, np.max(np.abs(rr))) and not eqs[j]:
                    tj = u[j] / rr[j]
                    if tj < t1:
                        t1, k1 = tj, j
            s_p = float(n_p @ y) - b_p
            t2 = np.inf if dependent else -s_p / znp
            t = min(t1, t2)
            if not np.isfinite(t):
                # Farkas: dy = (-rr on W, +1 on p) >= 0 on inequalities, N dy = n_p - N rr = H z ~ 0, and b' dy > 0
                dy = np.concatenate([-rr, [1.0]])
                margin = -s_p / max(1.0, np.max(np.abs(dy)))
                return dict(status="infeasible", y=y, active=W, iters=it, margin=margin, dep=znp / float(n_p @ Hi @ n_p))
            if not dependent:
                y = y + t * z
            u = [uj - t * rj for uj, rj in zip(u, rr)]
            u_p += t
            if t == t2:
                W.append((p, sp))
                u.append(u_p)
                break
            W.pop(k1)
            u.pop(k1)
    return dict(status="cap", y=y, active=W, iters=max_iter)


def solve_axis(r, M, T, pos, bcs, bce, klo, khi, rows):
    """rows: list of (segment, tau, d, lo, hi).  Box bounds klo / khi on the M - 1 interior waypoint rows (None: equalities)."""
    P, A = oracle.assemble(r, T)
    l, u = oracle.bounds(r, pos, bcs, bce)
    l, u = l.copy(), u.copy()
    wrows = [r + (r + 1) * i for i in range(M - 1)]
    if klo is not None:
        l[wrows] = klo
        u[wrows] = khi
    extra = [mono_row(r, M, s, tau * T[s], d) for (s, tau, d, _, _) in rows]
    if extra:
        A = np.vstack([A, np.array(extra)])
        l = np.r_[l, [x[3] for x in rows]]
        u = np.r_[u, [x[4] for x in rows]]
    eq = l == u
    Ae, be = A[eq], l[eq]
    # null-space reduction of the equality rows
    U_, s_, Vt = np.linalg.svd(Ae, full_matrices=True)
    rank = int((s_ > 1e-10 * s_[0]).sum())
    xp = np.linalg.lstsq(Ae, be, rcond=None)[0]
    if np.max(np.abs(Ae @ xp - be)) > 1e-7 * (1 + np.max(np.abs(be))):
        return dict(status="infeasible", note="equality rows inconsistent")
    Z = Vt[rank:].T
    if Z.shape[1] == 0:
        Ai = A[~eq]
        v = Ai @ xp
        ok = np.all(v >= l[~eq] - 1e-9 * (1 + np.abs(l[~eq]))) and np.all(v <= u[~eq] + 1e-9 * (1 + np.abs(u[~eq])))
        return dict(status="solved" if ok else "infeasible", note="no free unknown")
    H = Z.T @ P @ Z
    g = Z.T @ P @ xp
    Ci = A[~eq] @ Z
    ci0 = A[~eq] @ xp
    res = gi_dense(H, g, Ci, l[~eq] - ci0, u[~eq] - ci0)
    if "y" in res:
        res["x"] = xp + Z @ res["y"]
    return res


def classify(p, k):
    so, r, K = p["so"], p["r"], p["K"]
    s0, s1 = int(so[k]), int(so[k + 1])
    M = s1 - s0
    out = []
    for ax in range(3):
        rows = [(s, p["tau"][s0 + s, j], int(p["drv"][s0 + s, j]), max(p["rlo"][s0 + s, j, ax], -1e30), min(p["rhi"][s0 + s, j, ax], 1e30))
                for s in range(M) for j in range(K) if p["drv"][s0 + s, j] >= 0]
        klo = None if p["lo"] is None else p["lo"][s0 + k + 1:s1 + k, ax]
        khi = None if p["hi"] is None else p["hi"][s0 + k + 1:s1 + k, ax]
        out.append(solve_axis(r, M, p["T"][s0:s1], p["wp"][s0 + k:s1 + k + 1, ax], p["b"]["bc"][k, 0, :, ax], p["b"]["bc"][k, 1, :, ax], klo, khi, rows))
    return out


def main():
    import soak_rows
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 91
    un = np.load(os.path.join(ROOT, "gpurun_out", "soak_rows_unsolved_seed%d.npy" % seed))
    by_draw = {}
    for row in un:
        by_draw.setdefault(int(row[0]), []).append(row)
    rng = np.random.default_rng(seed)
    tally = {}
    for draw in range(max(by_draw) + 1):
        p = soak_rows.draw_problem(rng, draw, seed)
        rng.integers(0, p["n"], size=min(p["n"], 12))      # (the soak draws its certificate sample from the same stream)
        for row in by_draw.get(draw, []):
            k = int(row[1])
            res = classify(p, k)
            verdict = "infeasible" if any(x["status"] == "infeasible" for x in res) else ("cap" if any(x["status"] == "cap" for x in res) else "solved")
            key = (int(row[2]), int(row[4]), verdict)
            tally[key] = tally.get(key, 0) + 1
            if verdict != "infeasible" or int(row[4]) != -3:
                print("draw", draw, "k", k, "M", int(row[6]), "r", int(row[7]), "K", int(row[8]), "gpu", int(row[2]), "it", int(row[3]), "port", int(row[4]), "->",
                      [(x["status"], x.get("note", ""), "%.2e" % x.get("margin", 0.0)) for x in res])
    print("(status here, port, dense GI) -> count:", dict(sorted(tally.items())))


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
