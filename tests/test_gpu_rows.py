"""-m gpu: GENERAL inequality rows through the C ABI (uavqp_solve_rows_batch_device): knot boxes + rows
lo <= p_i^(d)(tau T_i) <= hi at in-segment times (position samples, velocity / acceleration limits).

Checkers (no reference code exists for these rows -- the reference only ever builds equality rows, minimum_control.cpp:98-125):
 (1) with every row wide open the result equals the corridor solve / the plain equality solve;
 (2) an exact optimality certificate from the REFERENCE-FORMULATION matrices (oracle.assemble = minimum_control.cpp:5-96) with
     the extra rows appended as monomial rows on the segment's coefficients: primal feasibility, stationarity
     P x + A' nu = 0, multipliers zero on inactive rows and right-signed on active ones;
 (3) the OSQP-faithful port with the same extra rows at eps 1e-10 (1e-5 relative = what ADMM reaches);
 (4) the exact-rational fixtures of tests/golden/rows_exact.json including the active sets (test_rows_golden.py)."""
import math

import numpy as np
import pytest

import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W

pytestmark = pytest.mark.gpu
BIG = 1e300


def mono_row(r, M, seg, t, d):
    from oracle.certificates import mono_row as f
    return f(r, M, seg, t, d)


def kkt_certificate_rows(oracle, r, M, T, coef, pos, bcs, bce, lo, hi, rows):
    """rows: list of (segment, tau, d, lo, hi).  Returns (primal violation, stationarity residual, complementarity violation):
    oracle/certificates.py (shared with bench.py's `parity` record)."""
    from oracle.certificates import kkt_certificate_rows as cert
    return cert(r, M, T, coef, pos, bcs, bce, lo, hi, rows)


def run_rows(ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, uniform):
    import torch
    dev = torch.device("cuda", 0)
    up = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    so = np.asarray(b["seg_offsets"])
    n = so.size - 1
    out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev)
    act = torch.zeros((n, 3, 2 + 2 * K), dtype=torch.int64, device=dev)
    ctx.solve_rows_device(r, n, uniform, int(np.max(np.diff(so))), None if uniform else up(so), up(np.asarray(b["waypoints"]).reshape(-1, 3)),
                          up(np.asarray(b["times"]).reshape(-1)), up(b["bc"]), up(lo), up(hi), K, up(tau), up(drv.astype(np.int32)), up(rlo), up(rhi),
                          out, st, it, act)
    ctx.synchronize()
    return out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy()


@pytest.mark.parametrize("r,M,K", [(3, 8, 2), (4, 6, 1), (3, 5, 1), (4, 4, 2)])
def test_wide_open_rows_reproduce_the_corridor_and_the_equality_solve(gpu_ctx, r, M, K):
    n = 40
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b, config_index=3)
    tau = np.tile(np.array([0.5, 0.25])[:K], (n * M, 1))
    drv = np.tile(np.array([1, 0])[:K], (n * M, 1))
    rlo, rhi = np.full((n * M, K, 3), -BIG), np.full((n * M, K, 3), BIG)
    got, st, it, act = run_rows(gpu_ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, M)
    ref, st2, _ = gpu_ctx.solve_corridor_batch_host(r, None, b["waypoints"], b["times"], b["bc"], lo, hi, uniform_segments=M)
    assert np.all(st == U.UAVQP_SOLVED) and np.all(st2 == U.UAVQP_SOLVED)
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))
    assert not act[:, :, 2:].any()
    got2, st3, _, _ = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, M)
    eq, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    assert np.all(st3 == U.UAVQP_SOLVED) and np.max(np.abs(got2 - eq)) < 1e-9 * np.max(np.abs(eq))


def _rows_problem(b, M, K, h_pos, v_lim, n):
    """Config-3 style extra rows: a position sample at mid-segment inside chord +- h_pos and (K = 2) a velocity limit there."""
    wp = b["waypoints"]
    tau = np.tile(np.array([0.5, 0.5])[:K], (n * M, 1))
    drv = np.tile(np.array([0, 1])[:K], (n * M, 1))
    rlo, rhi = np.zeros((n * M, K, 3)), np.zeros((n * M, K, 3))
    mid = 0.5 * (wp[:, :-1] + wp[:, 1:]).reshape(n * M, 3)
    rlo[:, 0], rhi[:, 0] = mid - h_pos, mid + h_pos
    if K == 2:
        rlo[:, 1], rhi[:, 1] = -v_lim, v_lim
    return tau, drv, rlo, rhi


@pytest.mark.parametrize("r,M,K,n", [(3, 8, 2, 48), (3, 16, 2, 24), (4, 6, 2, 48), (3, 5, 1, 48), (4, 8, 1, 32)])
def test_rows_kkt_certificate_and_osqp_port(gpu_ctx, oracle, r, M, K, n):
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b, config_index=3)
    tau, drv, rlo, rhi = _rows_problem(b, M, K, 0.2, 3.2, n)
    got, st, it, act = run_rows(gpu_ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, M)
    solved = st == U.UAVQP_SOLVED
    assert solved.mean() >= 0.9, np.unique(st, return_counts=True)     # (a few of these random problems are infeasible)
    assert np.all(solved | (st == U.UAVQP_MAX_ITER_REACHED) | (st == U.UAVQP_PRIMAL_INFEASIBLE))
    g = got.reshape(n, 3, 2 * r * M)
    worst, n_active_rows = np.zeros(3), 0
    for k in np.nonzero(solved)[0]:
        for ax in range(3):
            rows = [(s, tau[k * M + s, j], int(drv[k * M + s, j]), rlo[k * M + s, j, ax], rhi[k * M + s, j, ax]) for s in range(M) for j in range(K)]
            prim, stat, comp = kkt_certificate_rows(oracle, r, M, b["times"][k], g[k, ax], b["waypoints"][k, :, ax], b["bc"][k, 0, :, ax],
                                                    b["bc"][k, 1, :, ax], lo[k, 1:M, ax], hi[k, 1:M, ax], rows)
            worst = np.maximum(worst, [prim, stat, comp])
        n_active_rows += sum(bin(int(v) & 0xFFFFFFFFFFFFFFFF).count("1") for v in act[k, :, 2::2].ravel())
    assert worst[0] < 1e-9 and worst[1] < 1e-7 and worst[2] < 1e-6, worst
    assert n_active_rows > n                                          # the extra rows really bind
    # OSQP-faithful port with the same rows
    s = oracle.osqp_settings(eps_abs=1e-10, eps_rel=1e-10, max_iter=400000, eps_prim_inf=1e-7)
    ref, st_ref, _ = oracle.osqp_solve_batch(r, b["seg_offsets"], b["waypoints"], b["times"], b["bc"], settings=s, corr_lo=lo, corr_hi=hi,
                                             rows_per_segment=K, row_tau=tau, row_deriv=drv, row_lo=rlo, row_hi=rhi, threads=8)
    good = solved & (st_ref == oracle.PORT_SOLVED)
    assert good.sum() >= 0.8 * n
    rr = ref.reshape(n, 3, 2 * r * M)
    err = np.max(np.abs(g - rr), axis=(1, 2)) / np.max(np.abs(rr), axis=(1, 2))
    assert err[good].max() < 1e-5, err[good].max()


def test_rows_on_a_ragged_batch_and_knot_derivative_limits(gpu_ctx, oracle):
    """Ragged batch; row slot 0 = acceleration limit AT the knots (tau = 0, d = 2), waypoints as equalities (no boxes)."""
    r, n, K = 4, 60, 1
    b = W.ragged_batch(4, n, r, m_lo=2, m_hi=10)
    so = np.asarray(b["seg_offsets"])
    S = int(so[-1])
    tau = np.zeros((S, K))
    drv = np.full((S, K), 2)
    drv[so[:-1], 0] = -1                      # the first segment's slot: its tau = 0 knot is the start knot (fixed by bc)
    a_lim = 6.0
    rlo, rhi = np.full((S, K, 3), -a_lim), np.full((S, K, 3), a_lim)
    got, st, it, act = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, 0)
    assert np.all((st == U.UAVQP_SOLVED) | (st == U.UAVQP_MAX_ITER_REACHED) | (st == U.UAVQP_PRIMAL_INFEASIBLE)) and (st == U.UAVQP_SOLVED).mean() > 0.9
    wp = np.asarray(b["waypoints"]).reshape(-1, 3)
    T = np.asarray(b["times"])
    n_bind = 0
    for k in np.nonzero(st == U.UAVQP_SOLVED)[0][::3]:
        M = so[k + 1] - so[k]
        c = got[24 * so[k]:24 * so[k + 1]].reshape(3, M, 8)
        acc_knots = 2.0 * c[:, 1:, 2]                                  # acceleration at the start of segments 1..M-1
        assert np.all(np.abs(acc_knots) <= a_lim + 1e-8)
        n_bind += int((np.abs(np.abs(acc_knots) - a_lim) < 1e-8).sum())
        for ax in range(3):
            rows = [(s, 0.0, 2, -a_lim, a_lim) for s in range(1, M)]
            prim, stat, comp = kkt_certificate_rows(oracle, r, M, T[so[k]:so[k + 1]], c[ax].ravel(), wp[so[k] + k:so[k + 1] + k + 1, ax],
                                                    b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax], None, None, rows)
            assert prim < 1e-9 and stat < 1e-6 and comp < 1e-5, (prim, stat, comp)
    assert n_bind > 5


def test_rows_invalid_input_is_flagged(gpu_ctx):
    r, M, n, K = 3, 4, 6, 1
    b = W.uniform_batch(3, n, M, r)
    tau = np.full((n * M, K), 0.5)
    drv = np.zeros((n * M, K))
    rlo, rhi = np.full((n * M, K, 3), -BIG), np.full((n * M, K, 3), BIG)
    tau[2 * M + 1, 0] = 1.5                   # outside [0, 1)
    rlo[4 * M + 2, 0, 1], rhi[4 * M + 2, 0, 1] = 1.0, -1.0   # lo > hi
    drv[5 * M, 0] = 3                         # derivative order >= r
    got, st, it, act = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, M)
    assert list(st[[2, 4, 5]]) == [U.UAVQP_INVALID_INPUT] * 3 and np.all(st[[0, 1, 3]] == U.UAVQP_SOLVED)


def test_conflicting_rows_are_reported_not_solved(gpu_ctx):
    """Found by tools/soak_rows.py: when the rows of a working set become (numerically) dependent on the free unknowns -- an
    infeasible or degenerate problem -- its KKT system is singular, the solve no longer puts the active rows on their bounds, and
    such a problem used to come back UAVQP_SOLVED with violated rows.  Here the same functional twice with contradictory bounds
    (p(0.3 T) <= a and p(0.3 T) >= a + 0.1): the second row depends on the first, no multiplier of the working set blocks the dual
    direction -- a Farkas certificate with margin 0.1: UAVQP_PRIMAL_INFEASIBLE (round 4: UAVQP_MAX_ITER_REACHED, the verdict OSQP's
    eps_prim_inf test exists for, minimum_control.cpp:161, was thrown away); the feasible neighbours unaffected."""
    r, M, K, n = 3, 6, 2, 8
    b = W.uniform_batch(3, n, M, r, time_mode="reference")
    eq, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    c = eq.reshape(n, 3, M, 2 * r)
    tau = np.tile(np.array([0.3, 0.3]), (n * M, 1))
    drv = np.tile(np.array([0, 0]), (n * M, 1))
    rlo, rhi = np.full((n * M, K, 3), -BIG), np.full((n * M, K, 3), BIG)
    bad = [1, 5]
    for k in range(n):
        t = 0.3 * b["times"][k, 2]
        p = sum(c[k, :, 2, q] * t ** q for q in range(2 * r))                   # the unconstrained path at the sample of segment 2
        if k in bad:
            rhi[k * M + 2, 0] = p - 0.05                                         # slot 0: p(0.3 T) <= a
            rlo[k * M + 2, 1] = p + 0.05                                         # slot 1: p(0.3 T) >= a + 0.1
        else:
            rhi[k * M + 2, 0] = p + 0.5                                          # slack rows
            rlo[k * M + 2, 1] = p - 0.5
    got, st, it, act = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, M)
    good = np.setdiff1d(np.arange(n), bad)
    assert np.all(st[good] == U.UAVQP_SOLVED) and np.all(st[bad] == U.UAVQP_PRIMAL_INFEASIBLE), st
    assert np.all(np.isfinite(got))
    g = got.reshape(n, 3, M, 2 * r)
    assert np.max(np.abs(g[good] - c[good])) < 1e-9 * np.max(np.abs(c))         # their rows are slack: the equality solution
    assert it[bad].max() < 20                                                    # detected at once, not at the iteration cap


def test_rows_from_host_pointers_equal_the_device_entry(gpu_ctx):
    """uavqp_solve_rows_batch_host = the device entry behind staging copies: same coefficients, statuses and iteration counts;
    uniform and ragged, with and without knot boxes."""
    for r, ragged, boxes, K in ((3, False, True, 2), (4, True, False, 1), (4, True, True, 2)):
        n = 30
        b = W.ragged_batch(4, n, r, m_lo=2, m_hi=9) if ragged else W.uniform_batch(3, n, 7, r, time_mode="distance")
        so = np.asarray(b["seg_offsets"])
        S = int(so[-1])
        wp = np.asarray(b["waypoints"]).reshape(-1, 3)
        lo = hi = None
        if boxes:
            lo, hi = wp - 0.4, wp + 0.4
        rng = np.random.default_rng(7 + r)
        tau = rng.uniform(0.2, 0.8, size=(S, K))
        drv = np.tile(np.array([1, 0])[:K], (S, 1))
        T = np.asarray(b["times"]).reshape(-1)
        seg_traj = np.repeat(np.arange(n), np.diff(so))
        chord = (wp[np.arange(S) + seg_traj + 1] - wp[np.arange(S) + seg_traj]) / T[:, None]
        rlo, rhi = np.full((S, K, 3), -BIG), np.full((S, K, 3), BIG)
        lim = np.abs(chord).max(axis=1, keepdims=True) * 1.3 + 0.3
        rlo[:, 0], rhi[:, 0] = -lim, lim
        uni = 0 if ragged else 7
        dev_c, dev_st, dev_it, _ = run_rows(gpu_ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, uni)
        c, st, it = gpu_ctx.solve_rows_batch_host(r, None if uni else so, wp, T, b["bc"], lo, hi, K, tau, drv, rlo, rhi, uniform_segments=uni)
        assert np.array_equal(st, dev_st) and np.array_equal(it, dev_it)
        solved = np.repeat(st == U.UAVQP_SOLVED, np.diff(so) * 6 * r)
        assert np.array_equal(c[solved], dev_c[solved]) and solved.mean() > 0.8


def test_single_segment_rows_are_checked(gpu_ctx):
    """M = 1: the polynomial is fixed by the boundary data, its rows cannot be enforced, only checked -- a violated one reports
    UAVQP_PRIMAL_INFEASIBLE, a satisfied one UAVQP_SOLVED; the coefficients are the equality solution either way."""
    r, n, K = 4, 6, 1
    b = W.uniform_batch(3, n, 1, r, time_mode="reference")
    eq, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=1)
    c = eq.reshape(n, 3, 2 * r)
    t = 0.5 * b["times"][:, 0]
    v_mid = np.stack([sum(q * c[:, ax, q] * t ** (q - 1) for q in range(1, 2 * r)) for ax in range(3)], axis=1)
    tau, drv = np.full((n, K), 0.5), np.ones((n, K), dtype=np.int32)
    rlo, rhi = np.full((n, K, 3), -BIG), np.full((n, K, 3), BIG)
    rhi[:, 0] = v_mid + 0.1
    rhi[[2, 4], 0] = v_mid[[2, 4]] - 0.1                      # violated by the only possible trajectory
    got, st, it, act = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, 1)
    assert list(st) == [U.UAVQP_SOLVED, U.UAVQP_SOLVED, U.UAVQP_PRIMAL_INFEASIBLE, U.UAVQP_SOLVED, U.UAVQP_PRIMAL_INFEASIBLE, U.UAVQP_SOLVED]
    assert np.max(np.abs(got - eq)) < 1e-12 * np.max(np.abs(eq))


def test_python_traj_optimizer_facade_with_rows():
    """TrajOptimizer.setRows (the Python mirror of the C++ facade's method): a mid-segment velocity limit 15 % under the
    unconstrained speed; checked on the emitted polynomials."""
    r = 4
    b = W.ragged_batch(4, 25, r, m_lo=2, m_hi=9)
    so = np.asarray(b["seg_offsets"])
    n, S = so.size - 1, int(so[-1])
    wp_off = so + np.arange(n + 1)
    opt = U.TrajOptimizer(order=r)
    opt.setWaypoints(b["waypoints"], wp_off)
    opt.setTimeAllocation(b["times"])
    opt.setBoundary(b["bc"])
    assert opt.solve()
    T = np.asarray(b["times"]).reshape(-1)

    def vel_mid(o):
        out = np.zeros((S, 3))
        for k in range(n):
            c = o.getPolyCoeff(k)                                     # [3][M][2r]
            for s in range(so[k + 1] - so[k]):
                t = 0.5 * T[so[k] + s]
                out[so[k] + s] = sum(q * c[:, s, q] * t ** (q - 1) for q in range(1, 2 * r))
        return out
    v0 = vel_mid(opt)
    lim = 0.85 * np.abs(v0) + 0.2
    opt.setRows(1, np.full((S, 1), 0.5), np.ones((S, 1), dtype=np.int32), -lim[:, None, :], lim[:, None, :])
    ok = opt.solve()
    solved = opt.status == U.UAVQP_SOLVED
    assert solved.mean() > 0.7 and (ok == bool(solved.all()))
    v1 = vel_mid(opt)
    seg_solved = np.repeat(solved, np.diff(so))
    assert np.all(np.abs(v1[seg_solved]) <= lim[seg_solved] * (1 + 1e-9) + 1e-9)
    assert (np.abs(np.abs(v1[seg_solved]) - lim[seg_solved]) < 1e-7).sum() > 10      # the rows bind
    opt.setRows(0)
    assert opt.solve() and np.allclose(vel_mid(opt), v0, rtol=0, atol=1e-12)


@pytest.mark.parametrize("r,M,K", [(3, 16, 2), (4, 10, 2), (3, 7, 1)])
def test_pair_and_one_lane_rows_kernels_take_the_same_path(gpu_ctx, r, M, K):
    """uavqp_settings.rows_lanes_per_problem: the pair kernel (default, qp_rows2.h) and the one-lane kernel kept as its cross-check
    partner (qp_rows.h) are the same method -- statuses, iteration counts and working sets must be identical problem by problem,
    coefficients to rounding (the A/B existed only as tools/rows_ab.py and could rot)."""
    n = 96
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b, config_index=3)
    tau, drv, rlo, rhi = _rows_problem(b, M, K, 0.25, 3.5, n)
    res = {}
    try:
        for lanes in (2, 1):
            # (the starting set of qp_rows_dual.h exists for the pair kernel only: both from the box set here)
            gpu_ctx.set_settings(rows_lanes_per_problem=lanes, corridor_initial_guess=1)
            res[lanes] = run_rows(gpu_ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, M)
    finally:
        gpu_ctx.set_settings(rows_lanes_per_problem=0, corridor_initial_guess=2)
    assert np.array_equal(res[1][1], res[2][1]) and np.array_equal(res[1][2], res[2][2])
    ok = res[2][1] == U.UAVQP_SOLVED
    assert ok.mean() > 0.9 and np.array_equal(res[1][3][ok], res[2][3][ok])
    assert np.max(np.abs(res[1][0] - res[2][0])) <= 1e-9 * np.max(np.abs(res[2][0]))


def test_corridor_tail_shape_changes_the_launch_not_the_result(gpu_ctx):
    """uavqp_settings.corridor_tail_shape (two waves per CU with twice the sweep state on chip for small batches of long r = 4
    problems): bit-identical coefficients, statuses, iteration counts and working sets with and without it; with the primal
    method from its closed-form set, where the solve kernel iterates (the default cold start leaves it one verifying solve)."""
    import torch
    r, n = 4, 512
    b = W.ragged_batch(5, n, r, m_lo=12, m_hi=24)
    so = b["seg_offsets"]
    lo, hi = W.corridor_boxes(b, config_index=5)
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_so, d_wp, d_T, d_bc, d_lo, d_hi = up(so), up(b["waypoints"]), up(b["times"]), up(b["bc"]), up(lo), up(hi)
    res = {}
    try:
        for guess in (1, 2):
            for shape in (1, 0):
                gpu_ctx.set_settings(corridor_tail_shape=shape, corridor_initial_guess=guess)
                out = torch.zeros(int(so[-1]) * 6 * r, dtype=torch.float64, device=dev)
                st = torch.zeros(n, dtype=torch.int32, device=dev)
                it = torch.zeros(n, dtype=torch.int32, device=dev)
                act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
                gpu_ctx.solve_corridor_device(r, n, 0, 24, d_so, d_wp, d_T, d_bc, d_lo, d_hi, out, st, it, act, False)
                gpu_ctx.synchronize()
                res[(guess, shape)] = (out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy())
    finally:
        gpu_ctx.set_settings(corridor_tail_shape=1, corridor_initial_guess=2)
    for guess in (1, 2):
        for k in range(4):
            assert np.array_equal(res[(guess, 0)][k], res[(guess, 1)][k]), (guess, k)
    assert np.array_equal(res[(1, 1)][0], res[(2, 1)][0]) and np.array_equal(res[(1, 1)][3], res[(2, 1)][3])
    assert res[(1, 1)][2].max() > 3 and res[(2, 1)][2].mean() < 1.3


@pytest.mark.parametrize("r,M,K,ragged", [(3, 16, 2, False), (4, 10, 2, False), (3, 20, 1, False), (4, 12, 2, True), (3, 20, 2, False)])
def test_rows_prelude_changes_iterations_not_results(gpu_ctx, r, M, K, ragged):
    """uavqp_settings.corridor_initial_guess = 2 also gives the general-rows solve its starting set (qp_rows_dual.h: the dual method of
    the corridor prelude on the dense matrix G_ij = c_i' H^-1 c_j of ALL constraints -- boxes and rows as two-knot functionals, the rows
    QP itself): statuses and working sets identical, coefficients to rounding, and where it applies ((M - 1) + used rows <= 48
    constraints) barely more than the one verifying block solve per problem; the last case is beyond that: same counts as from the box
    set.  Segments with two different row times, unused slots and rows at tau = 0 are mixed in."""
    n = 64
    if ragged:
        b = W.ragged_batch(5, n, r, m_lo=2, m_hi=M, seed=77)
        so = np.asarray(b["seg_offsets"])
        tot = int(so[-1])
        wp = np.asarray(b["waypoints"])
        mid = np.concatenate([0.5 * (wp[so[k] + k:so[k + 1] + k] + wp[so[k] + k + 1:so[k + 1] + k + 1]) for k in range(n)])
        uni = 0
    else:
        b = W.uniform_batch(3, n, M, r, time_mode="distance")
        tot = n * M
        mid = 0.5 * (b["waypoints"][:, :-1] + b["waypoints"][:, 1:]).reshape(tot, 3)
        uni = M
    lo, hi = W.corridor_boxes(b, config_index=3)
    rng = np.random.default_rng(3)
    tau = np.tile(np.array([0.5, 0.5])[:K], (tot, 1))
    tau[rng.random(tot) < 0.3, 0] = 0.3                      # some segments with two different row times
    if K == 2:
        tau[rng.random(tot) < 0.05, 1] = 0.0                   # a velocity row AT a knot (tau = 0, d = 1)
    drv = np.tile(np.array([0, 1])[:K], (tot, 1))
    drv[rng.random(tot) < 0.1, K - 1] = -1                     # some unused slots
    rlo, rhi = np.zeros((tot, K, 3)), np.zeros((tot, K, 3))
    rlo[:, 0], rhi[:, 0] = mid - 0.25, mid + 0.25
    if K == 2:
        rlo[:, 1], rhi[:, 1] = -3.5, 3.5
    res = {}
    try:
        for guess in (2, 1):
            gpu_ctx.set_settings(corridor_initial_guess=guess)
            res[guess] = run_rows(gpu_ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, uni)
    finally:
        gpu_ctx.set_settings(corridor_initial_guess=2)
    # (round 4: a degenerate vertex -- as many active constraints as free variables -- ended the dual method from the box set as "capped"
    # where the verified starting set was accepted, seen on a 2-segment snap trajectory; round 5: the dependent constraint takes
    # Goldfarb-Idnani's zero-primal-step route and both starts end on the same verdict)
    assert np.array_equal(res[2][1], res[1][1]), (res[2][1][res[2][1] != res[1][1]], res[1][1][res[2][1] != res[1][1]])
    ok = (res[2][1] == U.UAVQP_SOLVED) & (res[1][1] == U.UAVQP_SOLVED)
    assert ok.mean() > 0.8
    assert np.array_equal(res[1][3][ok], res[2][3][ok])
    so_ = np.asarray(b["seg_offsets"])
    co = np.repeat(ok, np.diff(so_) * 6 * r)
    assert np.max(np.abs(res[1][0][co] - res[2][0][co])) <= 1e-9 * np.max(np.abs(res[1][0][co]))
    if (M - 1) + K * M <= 48:
        assert res[2][2][ok].mean() < 1.5 and res[2][2][ok].mean() < 0.3 * res[1][2][ok].mean(), (res[2][2][ok].mean(), res[1][2][ok].mean())
    else:
        assert np.array_equal(res[1][2], res[2][2])


def test_rows_prelude_deals_by_ticket_or_round_robin_with_the_same_results():
    """Round 6: rows_dual_kernel draws the trajectories of a wave from a ticket counter (the launch used to last as long as the unluckiest
    round-robin sum of solves); UAVQP_DEAL_TICKETS=0, read when a context is created, deals round-robin as before.  Which wave solves a
    trajectory decides nothing: everything identical bit for bit, on a batch of more trajectories than the launch has waves (so that
    tickets ARE drawn) whose size is not a multiple of anything."""
    import os
    r, M, K, n = 3, 8, 2, 5003
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    tot = n * M
    mid = 0.5 * (b["waypoints"][:, :-1] + b["waypoints"][:, 1:]).reshape(tot, 3)
    lo, hi = W.corridor_boxes(b, config_index=3)
    tau = np.tile(np.array([0.5, 0.5]), (tot, 1))
    drv = np.tile(np.array([0, 1]), (tot, 1))
    rlo, rhi = np.zeros((tot, K, 3)), np.zeros((tot, K, 3))
    rlo[:, 0], rhi[:, 0] = mid - 0.25, mid + 0.25
    rlo[:, 1], rhi[:, 1] = -3.5, 3.5
    out = {}
    for tag in ("1", "0"):
        os.environ["UAVQP_DEAL_TICKETS"] = tag
        try:
            with U.Context(0) as ctx:
                out[tag] = run_rows(ctx, r, b, lo, hi, K, tau, drv, rlo, rhi, M)
        finally:
            os.environ.pop("UAVQP_DEAL_TICKETS", None)
    assert (out["1"][1] == U.UAVQP_SOLVED).mean() > 0.8 and out["1"][2].mean() < 1.5      # (the prelude's sets were the solution's)
    for a, c in zip(out["1"], out["0"]):
        assert np.array_equal(a, c, equal_nan=True)


def _one_row_problem(n, gap, duplicate=False):
    """n copies of a 6-segment jerk problem whose segment 2 carries the SAME position sample twice: slot 0 bounds it from above at
    a, slot 1 from below at a + gap (gap > 0: no common point; gap = 0: one point, the two rows are the same equation)."""
    r, M, K = 3, 6, 2
    b = W.uniform_batch(3, n, M, r, time_mode="reference")
    tau = np.tile(np.array([0.3, 0.3]), (n * M, 1))
    drv = np.tile(np.array([0, 0]), (n * M, 1))
    rlo, rhi = np.full((n * M, K, 3), -BIG), np.full((n * M, K, 3), BIG)
    return r, M, K, b, tau, drv, rlo, rhi


def test_eps_prim_inf_is_the_margin_of_the_infeasibility_certificate(gpu_ctx):
    """uavqp_settings.eps_prim_inf (the reference's setPrimalInfeasibilityTollerance(1e-3), minimum_control.cpp:161) is the test OSQP
    applies to a certificate dy: u' max(dy, 0) + l' min(dy, 0) <= -eps |dy|_inf.  Two rows p(0.3 T) <= a and p(0.3 T) >= a + gap have
    the certificate (1, -1) with value -gap: PRIMAL_INFEASIBLE for gap >= eps, undecided (MAX_ITER_REACHED) below it -- and with a
    smaller eps the same problem is called infeasible."""
    n = 6
    r, M, K, b, tau, drv, rlo, rhi = _one_row_problem(n, 0.0)
    eq, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    c = eq.reshape(n, 3, M, 2 * r)
    gaps = [1e-1, 1e-2, 2e-3, 5e-4, 1e-5, 1e-7]
    for k, gap in enumerate(gaps):
        t = 0.3 * b["times"][k, 2]
        p = sum(c[k, :, 2, q] * t ** q for q in range(2 * r))
        rhi[k * M + 2, 0] = p - 0.05
        rlo[k * M + 2, 1] = p - 0.05 + gap
    try:
        _, st, it, _ = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, M)
        assert list(st) == [U.UAVQP_PRIMAL_INFEASIBLE] * 3 + [U.UAVQP_MAX_ITER_REACHED] * 3, st      # default eps_prim_inf = 1e-3
        gpu_ctx.set_settings(eps_prim_inf=1e-6)
        _, st, it, _ = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, M)
        assert list(st) == [U.UAVQP_PRIMAL_INFEASIBLE] * 5 + [U.UAVQP_MAX_ITER_REACHED], st
    finally:
        gpu_ctx.set_settings(eps_prim_inf=1e-3)
    assert it.max() < 20


def test_dependent_but_consistent_rows_are_solved(gpu_ctx, oracle):
    """The degenerate side of the same coin: the same functional bounded from above AND from below at the SAME value (two inequality
    rows that are one equation) and, next to it, a row that repeats an active one with a looser bound.  The second row of the pair
    depends on the first; the multiplier of the first blocks the dual direction, it leaves, the other enters: SOLVED, on the point
    the certificate accepts (round 4: UAVQP_MAX_ITER_REACHED)."""
    n = 4
    r, M, K, b, tau, drv, rlo, rhi = _one_row_problem(n, 0.0)
    eq, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    c = eq.reshape(n, 3, M, 2 * r)
    for k in range(n):
        t = 0.3 * b["times"][k, 2]
        p = sum(c[k, :, 2, q] * t ** q for q in range(2 * r))
        if k < 2:
            rhi[k * M + 2, 0] = p - 0.05            # p <= a
            rlo[k * M + 2, 1] = p - 0.05            # p >= a: together p = a
        else:
            rhi[k * M + 2, 0] = p - 0.05            # p <= a (active)
            rhi[k * M + 2, 1] = p - 0.05 + (0.0 if k == 2 else 0.02)     # the same row again / a looser copy
    got, st, it, act = run_rows(gpu_ctx, r, b, None, None, K, tau, drv, rlo, rhi, M)
    assert np.all(st == U.UAVQP_SOLVED), st
    g = got.reshape(n, 3, 2 * r * M)
    for k in range(n):
        for ax in range(3):
            rows = [(2, 0.3, 0, rlo[k * M + 2, j, ax], rhi[k * M + 2, j, ax]) for j in range(K)]
            prim, stat, comp = kkt_certificate_rows(oracle, r, M, b["times"][k], g[k, ax], b["waypoints"][k, :, ax], b["bc"][k, 0, :, ax],
                                                    b["bc"][k, 1, :, ax], None, None, rows)
            assert prim < 1e-9 and stat < 1e-7, (k, ax, prim, stat)
