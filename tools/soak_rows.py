#!/usr/bin/env python3
"""Randomised soak of the general-rows solver (uavqp_solve_rows_batch_device) against the KKT certificate built from the
reference-formulation matrices (test infrastructure, like tests/): python tools/soak_rows.py [n_draws] [seed].
Every draw: random (r, ragged | uniform, M <= 20, K, batch size <= 400, time allocation, with / without knot boxes, warm / cold
start), random rows per segment and slot -- unused, position sample, velocity / acceleration (/ jerk) limit, at a random tau,
one- or two-sided, a few equalities.  Solved problems must pass the certificate.  EVERY trajectory that does not end UAVQP_SOLVED goes to
the OSQP port (eps 1e-9, eps_prim_inf 1e-7: tolerances at which its verdict means what it says -- at the reference's 1e-3 its
certificate test fires on feasible problems, tests/test_gpu_baseline_sizes.py): the port finds a solution -> a solver defect, the
soak fails; UAVQP_PRIMAL_INFEASIBLE needs the port's infeasibility verdict (or its failure to converge); the verdicts are tallied.
Exit code 1 on the first failure.  UAVQP_SOAK_LENIENT=1: tally only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import uav_motion_planning_amd as U  # noqa: E402
from uav_motion_planning_amd import workloads as W  # noqa: E402
from oracle import oracle  # noqa: E402
from test_gpu_rows import kkt_certificate_rows, run_rows  # noqa: E402

BIG = 1e300


def draw_problem(rng, draw, seed):
    """One random draw (deterministic in (seed, draw) given the generator state): everything the solver is handed."""
    r = int(rng.choice([3, 4]))
    K = int(rng.choice([1, 2]))
    ragged = bool(rng.integers(0, 2))
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400)], p=[0.5, 0.5]))
    warm = int(rng.integers(0, 2))
    if ragged:
        b = W.ragged_batch(draw, n, r, m_lo=1, m_hi=int(rng.integers(2, 21)), seed=seed * 100000 + draw)
        b["times"] = b["times"] * rng.uniform(0.7, 2.0, size=b["times"].shape)
        uni = 0
    else:
        M = int(rng.integers(1, 21))
        b = W.uniform_batch(draw, n, M, r, time_mode=str(rng.choice(["reference", "distance"])), seed=seed * 100000 + draw)
        uni = M
    so = np.asarray(b["seg_offsets"])
    S = int(so[-1])
    wp = np.asarray(b["waypoints"]).reshape(-1, 3)
    T = np.asarray(b["times"]).reshape(-1)
    boxes = bool(rng.integers(0, 2))
    lo = hi = None
    if boxes:
        h = 10.0 ** rng.uniform(-1.5, 0, size=wp.shape)
        h[rng.random(size=wp.shape) < 0.1] = 0.0
        lo, hi = wp - h, wp + h
    tau = rng.uniform(0.05, 0.95, size=(S, K))
    tau[rng.random(size=(S, K)) < 0.15] = 0.0
    drv = rng.integers(0, r, size=(S, K))
    drv[(tau == 0.0) & (drv == 0)] = 1                       # a position row at a knot is the knot box: not a row
    drv[rng.random(size=(S, K)) < 0.25] = -1                  # unused slots
    first = np.zeros(S, dtype=bool)
    first[so[:-1][np.diff(so) > 0]] = True
    drv[first[:, None] & (tau == 0.0)] = -1                   # tau = 0 of the first segment is the start knot (fixed by bc)
    seg_traj = np.repeat(np.arange(n), np.diff(so))
    seg_idx = np.arange(S) - so[seg_traj]
    chord = (wp[np.arange(S) + seg_traj + 1] - wp[np.arange(S) + seg_traj]) / T[:, None]          # mean velocity of the segment
    rlo, rhi = np.full((S, K, 3), -BIG), np.full((S, K, 3), BIG)
    for j in range(K):
        for d in range(r):
            sel = drv[:, j] == d
            if not sel.any():
                continue
            if d == 0:
                mid = wp[np.arange(S) + seg_traj] + tau[:, j:j + 1] * (wp[np.arange(S) + seg_traj + 1] - wp[np.arange(S) + seg_traj])
                w = 10.0 ** rng.uniform(-1.2, 0, size=(S, 3))
                rlo[sel, j], rhi[sel, j] = (mid - w)[sel], (mid + w)[sel]
            else:
                lim = (np.abs(chord).max(axis=1, keepdims=True) + 0.5) * rng.uniform(1.05, 2.5, size=(S, 1)) * (3.0 ** (d - 1))
                rlo[sel, j], rhi[sel, j] = np.broadcast_to(-lim, (S, 3))[sel], np.broadcast_to(lim, (S, 3))[sel]
        one_sided = rng.random(size=S) < 0.2
        rlo[one_sided, j] = -BIG
    return dict(r=r, K=K, ragged=ragged, n=n, warm=warm, b=b, uni=uni, so=so, S=S, wp=wp, T=T, boxes=boxes, lo=lo, hi=hi, tau=tau, drv=drv, rlo=rlo, rhi=rhi)


def port_verdict(p, k):
    """The OSQP port's verdict on trajectory k of a draw (all three axes; PORT_PRIMAL_INFEASIBLE wins)."""
    so, r, K = p["so"], p["r"], p["K"]
    s0, s1 = int(so[k]), int(so[k + 1])
    M = s1 - s0
    st_ = oracle.osqp_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=300000, eps_prim_inf=1e-7)
    kw = {}
    if p["lo"] is not None:
        kw.update(corr_lo=p["lo"][s0 + k:s1 + k + 1], corr_hi=p["hi"][s0 + k:s1 + k + 1])
    drv = p["drv"][s0:s1].copy()
    if M == 1:
        pass
    _, st, it = oracle.osqp_solve_batch(r, np.array([0, M], dtype=np.int32), p["wp"][s0 + k:s1 + k + 1], p["T"][s0:s1], p["b"]["bc"][k:k + 1], settings=st_, threads=1,
                                        rows_per_segment=K, row_tau=p["tau"][s0:s1], row_deriv=drv, row_lo=np.maximum(p["rlo"][s0:s1], -1e30),
                                        row_hi=np.minimum(p["rhi"][s0:s1], 1e30), **kw)
    return int(st[0]), int(it[0])


def main():
    n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = U.Context(0)
    # every certificate counts (the default margin, the reference's 1e-3, leaves barely infeasible draws "undecided"); UAVQP_SOAK_LANES=1: the one-lane kernel
    ctx.set_settings(eps_prim_inf=float(os.environ.get("UAVQP_SOAK_EPS", "1e-9")), rows_lanes_per_problem=int(os.environ.get("UAVQP_SOAK_LANES", "0")))
    worst = np.zeros(3)
    n_solved = n_capped = n_checked = n_bind = n_inf = 0
    lenient = os.environ.get("UAVQP_SOAK_LENIENT", "0") == "1"
    tally, unsolved = {}, []
    for draw in range(n_draws):
        p = draw_problem(rng, draw, seed)
        r, K, ragged, n, b, uni, so, wp, T, boxes, lo, hi, tau, drv, rlo, rhi = (p[k_] for k_ in ("r", "K", "ragged", "n", "b", "uni", "so", "wp", "T", "boxes", "lo", "hi", "tau", "drv", "rlo", "rhi"))
        ctx.set_settings(warm_start=p["warm"])
        got, st, it, act = run_rows(ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, uni)
        Ms = np.diff(so)
        if not np.all((st == U.UAVQP_SOLVED) | (st == U.UAVQP_MAX_ITER_REACHED) | (st == U.UAVQP_PRIMAL_INFEASIBLE) | ((Ms > 63) & (st == U.UAVQP_INVALID_INPUT))):
            print("STATUS FAILURE draw", draw, np.unique(st, return_counts=True))
            return 1
        n_solved += int((st == U.UAVQP_SOLVED).sum())
        n_capped += int((st == U.UAVQP_MAX_ITER_REACHED).sum())
        n_inf += int((st == U.UAVQP_PRIMAL_INFEASIBLE).sum())
        for k in np.nonzero((st == U.UAVQP_MAX_ITER_REACHED) | (st == U.UAVQP_PRIMAL_INFEASIBLE))[0]:
            pv, pit = port_verdict(p, int(k))
            key = (int(st[k]), pv)
            tally[key] = tally.get(key, 0) + 1
            unsolved.append((draw, int(k), int(st[k]), int(it[k]), pv, pit, int(Ms[k]), r, K))
            if pv == oracle.PORT_SOLVED and not lenient:
                # the port finds a solution of a problem this back-end gave up on (or called infeasible): a solver defect
                print("UNSOLVED BUT FEASIBLE draw", draw, dict(r=r, K=K, ragged=ragged, n=n, k=int(k), M=int(Ms[k]), boxes=boxes, status=int(st[k]), iters=int(it[k]), port_iters=pit))
                return 1
        nc = 2 * r
        for k in np.unique(rng.integers(0, n, size=min(n, 12))):
            if st[k] != U.UAVQP_SOLVED:
                continue
            s0, s1 = int(so[k]), int(so[k + 1])
            M = s1 - s0
            c = got[3 * nc * s0:3 * nc * s1].reshape(3, M * nc)
            for ax in range(3):
                rows = [(s, tau[s0 + s, j], int(drv[s0 + s, j]), rlo[s0 + s, j, ax], rhi[s0 + s, j, ax]) for s in range(M) for j in range(K)
                        if drv[s0 + s, j] >= 0 and M >= 2]
                prim, stat, comp = kkt_certificate_rows(oracle, r, M, T[s0:s1], c[ax], wp[s0 + k:s1 + k + 1, ax], b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax],
                                                        None if lo is None else lo[s0 + k + 1:s1 + k, ax], None if hi is None else hi[s0 + k + 1:s1 + k, ax], rows)
                worst = np.maximum(worst, [prim, stat, comp])
                n_checked += 1
                if not (prim < 1e-8 and stat < 1e-5 and comp < 1e-4):
                    print("ROWS FAILURE draw", draw, dict(r=r, K=K, ragged=ragged, n=n, k=int(k), M=M, ax=ax, boxes=boxes, prim=prim, stat=stat, comp=comp,
                                                          iters=int(it[k]), warm=ctx.get_settings().warm_start))
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez(os.path.join(ROOT, "gpurun_out", "soak_rows_fail.npz"), r=r, K=K, uni=uni, so=so, wp=wp, T=T, bc=b["bc"], lo=np.zeros(0) if lo is None else lo,
                             hi=np.zeros(0) if hi is None else hi, tau=tau, drv=drv, rlo=rlo, rhi=rhi, got=got, st=st, it=it, act=act, k=k, ax=ax)
                    return 1
            n_bind += sum(bin(int(v) & 0xFFFFFFFFFFFFFFFF).count("1") for v in act[k, :, 2::2].ravel())
    ctx.set_settings(warm_start=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "soak_rows_unsolved_seed%d.npy" % seed), np.array(unsolved, dtype=np.int64).reshape(-1, 9))
    print("rows soak verdicts (status here, OSQP-port status at eps 1e-9 / eps_prim_inf 1e-7) -> count:", dict(sorted(tally.items())), "| primal infeasible here:", n_inf)
    print("rows soak ok: %d draws, seed %d, %d solved, %d capped, %d (trajectory, axis) certificates, %d active rows in them, worst prim %.2e stat %.2e comp %.2e"
          % (n_draws, seed, n_solved, n_capped, n_checked, n_bind, worst[0], worst[1], worst[2]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
