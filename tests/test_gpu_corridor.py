"""-m gpu: corridor-constrained solve (north-star extension, BASELINE configs 3 / 5) through the C ABI.

Checkers: (1) with lo = hi = waypoints it must reproduce the plain equality solve; (2) the OSQP-faithful
port with the waypoint rows turned into l <= p <= u, at tight eps (it converges to the QP's minimiser;
tolerance 1e-5 relative = what ADMM reaches, the north star's budget); (3) an exact optimality certificate
from the reference-formulation matrices (oracle.assemble): primal feasibility, stationarity P x + A' nu = 0
with multipliers that vanish on inactive corridor rows and have the right sign on active ones."""
import numpy as np
import pytest

import uav_motion_planning_amd as U
from uav_motion_planning_amd import workloads as W

pytestmark = pytest.mark.gpu


def kkt_certificate(oracle, r, M, T, coef, pos, bcs, bce, lo, hi):
    """(max primal violation, max stationarity residual, max complementarity violation) of the corridor QP in the reference formulation:
    oracle/certificates.py (shared with bench.py's `parity` record)."""
    from oracle.certificates import kkt_certificate as cert
    return cert(r, M, T, coef, pos, bcs, bce, lo, hi)


@pytest.mark.parametrize("r,M", [(3, 8), (4, 8), (3, 16)])
def test_degenerate_corridor_equals_equality_solve(gpu_ctx, r, M):
    b = W.uniform_batch(3, 200, M, r, time_mode="distance")
    ref, st0 = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    got, st, it = gpu_ctx.solve_corridor_batch_host(r, None, b["waypoints"], b["times"], b["bc"], b["waypoints"], b["waypoints"],
                                                    uniform_segments=M)
    assert np.all(st == U.UAVQP_SOLVED) and np.all(it <= 1)
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


@pytest.mark.parametrize("r,M,n", [(3, 16, 48), (4, 8, 48), (3, 5, 64), (4, 3, 64)])
def test_corridor_vs_osqp_port_and_kkt_certificate(gpu_ctx, oracle, r, M, n):
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b, h_lo=0.1, h_hi=0.6)
    got, st, it = gpu_ctx.solve_corridor_batch_host(r, None, b["waypoints"], b["times"], b["bc"], lo, hi, uniform_segments=M)
    assert np.all(st == U.UAVQP_SOLVED), st
    assert it.max() <= 8 * M + 20
    g = got.reshape(n, 3, 2 * r * M)
    # (3) optimality certificate on every trajectory / axis
    worst = np.zeros(3)
    n_active = 0
    for k in range(n):
        for ax in range(3):
            prim, stat, comp = kkt_certificate(oracle, r, M, b["times"][k], g[k, ax], b["waypoints"][k, :, ax],
                                               b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax], lo[k, 1:M, ax], hi[k, 1:M, ax])
            worst = np.maximum(worst, [prim, stat, comp])
    assert worst[0] < 1e-9 and worst[1] < 1e-7 and worst[2] < 1e-6, worst
    # (2) OSQP port at tight eps
    s = oracle.osqp_settings(eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, eps_prim_inf=1e-7)
    ref, st_ref, _ = oracle.osqp_solve_batch(r, b["seg_offsets"], b["waypoints"], b["times"], b["bc"], settings=s,
                                             corr_lo=lo, corr_hi=hi, threads=8)
    good = st_ref == oracle.PORT_SOLVED
    assert good.sum() >= 0.9 * n
    rr = ref.reshape(n, 3, 2 * r * M)
    err = np.max(np.abs(g - rr), axis=(1, 2)) / np.max(np.abs(rr), axis=(1, 2))
    assert err[good].max() < 1e-5, err[good].max()
    # the corridor must actually bind somewhere, and loosening constraints can only lower the cost
    eq, _ = gpu_ctx.solve_batch_host(r, None, b["waypoints"], b["times"], b["bc"], uniform_segments=M)
    e = eq.reshape(n, 3, 2 * r * M)
    for k in range(0, n, 7):
        for ax in range(3):
            assert oracle.cost(r, b["times"][k], g[k, ax]) <= oracle.cost(r, b["times"][k], e[k, ax]) * (1 + 1e-9) + 1e-12
    # (binding = the solution differs from the equality solve; the iteration count no longer shows it: the default cold start
    # begins at the set the position-space dual method found, and one solve verifies it)
    assert np.max(np.abs(g - e)) > 1e-3 * np.max(np.abs(e))
    assert it.max() >= 1


def test_corridor_ragged_and_wide_open(gpu_ctx, oracle):
    """Ragged batch; boxes so wide that no bound is active: the optimum has free interior positions."""
    r, n = 4, 60
    b = W.ragged_batch(4, n, r, m_lo=2, m_hi=10)
    wp = np.asarray(b["waypoints"])
    lo, hi = W.corridor_boxes(b, h_lo=1e3, h_hi=2e3)
    got, st, it = gpu_ctx.solve_corridor_batch_host(r, b["seg_offsets"], wp, b["times"], b["bc"], lo, hi)
    assert np.all(st == U.UAVQP_SOLVED) and it.max() <= 1
    so = b["seg_offsets"]
    for k in range(0, n, 5):
        M = so[k + 1] - so[k]
        for ax in range(3):
            c = got[24 * so[k]:24 * so[k + 1]].reshape(3, 8 * M)[ax]
            w = wp[so[k] + k:so[k + 1] + k + 1, ax]
            prim, stat, comp = kkt_certificate(oracle, r, M, b["times"][so[k]:so[k + 1]], c, w, b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax],
                                               lo[so[k] + k + 1:so[k + 1] + k, ax], hi[so[k] + k + 1:so[k + 1] + k, ax])
            assert prim < 1e-9 and stat < 1e-7 and comp < 1e-6


def test_corridor_invalid_box_is_flagged(gpu_ctx):
    b = W.uniform_batch(3, 6, 4, 3)
    lo, hi = W.corridor_boxes(b)
    lo[2, 2, 1], hi[2, 2, 1] = 1.0, -1.0
    got, st, it = gpu_ctx.solve_corridor_batch_host(3, None, b["waypoints"], b["times"], b["bc"], lo, hi, uniform_segments=4)
    assert st[2] == U.UAVQP_INVALID_INPUT and np.all(np.delete(st, 2) == U.UAVQP_SOLVED)


def test_config5_shape_corridor_plus_time_reallocation_outer_loop(oracle):
    """BASELINE config 5 in miniature: ragged batch + corridors + <= 5 outer time re-allocations, all on device
    buffers.  No reference behaviour exists for the outer loop (SURVEY.md section 8-a'); checked properties:
    durations only grow, the loop stops, peak speed / acceleration end within the limits (5 % sampling slack), and
    the final inner solve is the corridor QP's minimiser for the final time allocation (KKT certificate)."""
    import torch
    r, n, v_max, a_max = 4, 96, 7.0, 10.0   # max_velocity / max_accelration of test_kino_astar_searching.launch:49-50
    b = W.ragged_batch(5, n, r, m_lo=3, m_hi=12)
    so = b["seg_offsets"]
    lo, hi = W.corridor_boxes(b, config_index=5)
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_so, d_wp, d_T, d_bc, d_lo, d_hi = up(so), up(b["waypoints"]), up(b["times"]), up(b["bc"]), up(lo), up(hi)
    d_out = torch.zeros(int(so[-1]) * 24, dtype=torch.float64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_it = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ch = torch.zeros(n, dtype=torch.int32, device=dev)
    lib = U.lib()
    with U.Context(0) as ctx:
        T_prev = b["times"].copy()
        outer = 0
        for outer in range(1, 7):
            rc = lib.uavqp_solve_corridor_batch_device(ctx._h, r, n, 0, 12, d_so.data_ptr(), d_wp.data_ptr(), d_T.data_ptr(), d_bc.data_ptr(),
                                                       d_lo.data_ptr(), d_hi.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), d_it.data_ptr())
            assert rc == 0
            ctx.time_reallocate_device(r, n, 0, d_so, d_T, d_out, v_max, a_max, samples_per_seg=16, max_stretch=2.0, changed=d_ch)
            ctx.synchronize()
            assert bool((d_st == U.UAVQP_SOLVED).all())
            T_now = d_T.cpu().numpy()
            assert np.all(T_now >= T_prev * (1 - 1e-15))
            T_prev = T_now
            if int(d_ch.sum().item()) == 0:
                break
        assert outer <= 6 and int(d_ch.sum().item()) == 0, "time re-allocation did not settle"
        coef = d_out.cpu().numpy()
        # limits hold on a finer grid than the one used for the update
        n_s = 400
        tot = np.array([T_prev[so[k]:so[k + 1]].sum() for k in range(n)])
        d_ev = torch.zeros(n * n_s * 6, dtype=torch.float64, device=dev)
        for k_dt in (tot.max() / (n_s - 1),):
            ctx.eval_batch_device(r, n, 0, d_so, d_T, d_out, n_s, 0.0, float(k_dt), 6, d_ev)
            ctx.synchronize()
            ev = d_ev.cpu().numpy().reshape(n, n_s, 2, 3)
            assert np.max(np.linalg.norm(ev[:, :, 0], axis=2)) <= 1.05 * v_max
            assert np.max(np.linalg.norm(ev[:, :, 1], axis=2)) <= 1.05 * a_max
    # final inner solve is optimal for the final allocation
    wp = np.asarray(b["waypoints"])
    for k in range(0, n, 11):
        M = so[k + 1] - so[k]
        for ax in range(3):
            c = coef[24 * so[k]:24 * so[k + 1]].reshape(3, 8 * M)[ax]
            prim, stat, comp = kkt_certificate(oracle, r, M, T_prev[so[k]:so[k + 1]], c, wp[so[k] + k:so[k + 1] + k + 1, ax],
                                               b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax],
                                               lo[so[k] + k + 1:so[k + 1] + k, ax], hi[so[k] + k + 1:so[k + 1] + k, ax])
            assert prim < 1e-9 and stat < 1e-7 and comp < 1e-6


@pytest.mark.parametrize("r,ragged", [(3, False), (4, True)])
def test_warm_started_corridor_solve(gpu_ctx, r, ragged):
    """uavqp_solve_corridor_warm_device: (1) restarting from the working set of the solution takes exactly one
    iteration and reproduces the cold solve bit for bit (same pins, same pinned values, same arithmetic);
    (2) ANY bit pattern is an admissible guess -- random sets still end at the same minimiser (1e-9 relative);
    (3) after a 10 % time re-allocation the previous working set is a good guess: far fewer iterations than cold;
    (4) warm_start = 2 additionally reads the previous polynomials from coeff_out as the starting point: same minimiser whatever
    is found there (the previous solution, NaNs, huge numbers), and about as few iterations as (3) from the previous solution (its benefit shows after large changes: config 5's
    outer loop, tools/corridor_tail_probe.py)."""
    import torch
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    if ragged:
        n = 150
        b = W.ragged_batch(5, n, r, m_lo=2, m_hi=20)
        uni, mx = 0, 20
    else:
        n, M = 200, 16
        b = W.uniform_batch(3, n, M, r, time_mode="distance")
        uni, mx = M, M
    so = b["seg_offsets"]
    lo, hi = W.corridor_boxes(b, config_index=3)
    d_so = up(so) if ragged else None
    d_wp, d_T, d_bc, d_lo, d_hi = up(b["waypoints"]), up(b["times"]), up(b["bc"]), up(lo), up(hi)
    nco = int(so[-1]) * 6 * r

    def run(active, warm, times=d_T, out_init=None):
        out = torch.zeros(nco, dtype=torch.float64, device=dev) if out_init is None else out_init.clone()
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        it = torch.zeros(n, dtype=torch.int32, device=dev)
        gpu_ctx.solve_corridor_device(r, n, uni, mx, d_so, d_wp, times, d_bc, d_lo, d_hi, out, st, it, active, warm)
        gpu_ctx.synchronize()
        assert bool((st == U.UAVQP_SOLVED).all())
        return out.cpu().numpy(), it.cpu().numpy()

    act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
    c_cold, it_cold = run(act, False)
    c_plain, _ = run(None, False)
    assert np.array_equal(c_cold, c_plain)
    sets = act.cpu().numpy().copy()
    assert np.any(sets[:, :, 0] != 0)                       # some boxes are active at the solution
    assert np.all((sets[:, :, 1] & ~sets[:, :, 0]) == 0)    # "upper" only where active
    # (1) exact restart
    c_warm, it_warm = run(act, True)
    assert np.array_equal(c_warm, c_cold)
    assert np.all(it_warm == 1)
    assert np.array_equal(act.cpu().numpy(), sets)
    # (2) garbage guesses
    rng = np.random.default_rng(9)
    junk = torch.from_numpy(rng.integers(-2**63, 2**63 - 1, size=(n, 3, 2), dtype=np.int64)).to(dev)
    c_junk, it_junk = run(junk, True)
    assert np.max(np.abs(c_junk - c_cold)) <= 1e-9 * np.max(np.abs(c_cold))
    assert np.array_equal(junk.cpu().numpy(), sets)
    # (3) outer-loop situation
    d_T2 = d_T * 1.1
    act2 = act.clone()
    c2_warm, it2_warm = run(act2, True, d_T2)
    c2_cold, it2_cold = run(None, False, d_T2)
    assert np.max(np.abs(c2_warm - c2_cold)) <= 1e-9 * np.max(np.abs(c2_cold))
    # (the default cold start already begins at the set the position-space dual method found -- ~1 solve; "cold" in the sense of
    # this comparison is the primal method from its closed-form set)
    gpu_ctx.set_settings(corridor_initial_guess=1)
    try:
        c2_cold1, it2_cold1 = run(None, False, d_T2)
    finally:
        gpu_ctx.set_settings(corridor_initial_guess=2)
    assert np.array_equal(c2_cold1, c2_cold)
    assert it2_warm.mean() < 0.5 * it2_cold1.mean()
    assert it2_cold.mean() < 1.5
    # (4) the previous polynomials as the starting point
    prev = torch.from_numpy(c_cold).to(dev)
    act3 = act.clone()
    c3, it3 = run(act3, 2, d_T2, out_init=prev)
    assert np.max(np.abs(c3 - c2_cold)) <= 1e-9 * np.max(np.abs(c2_cold))
    assert np.array_equal(act3.cpu().numpy(), act2.cpu().numpy())
    assert it3.mean() <= 1.1 * it2_warm.mean() + 0.1            # (a 10 % stretch barely moves the set: both need 1-3 iterations)
    for junk_val in (float("nan"), 1e300, -1e300):
        c4, _ = run(act.clone(), 2, d_T2, out_init=torch.full((nco,), junk_val, dtype=torch.float64, device=dev))
        assert np.max(np.abs(c4 - c2_cold)) <= 1e-9 * np.max(np.abs(c2_cold))
    c5, _ = run(None, 0, d_T2, out_init=torch.full((nco,), float("nan"), dtype=torch.float64, device=dev))   # cold: coeff_out is write-only
    assert np.array_equal(c5, c2_cold)


def test_mid_segment_samples_by_knot_insertion(gpu_ctx, oracle):
    """adapters.refine_with_mid_knots + the ordinary corridor solve: positions at the inserted knots (the
    mid-segment sample times of the original segments) stay inside chord +- h, the refined solve is KKT-optimal for
    the refined problem, and a corridor that is wide at the mids reproduces nothing tighter than the plain corridor."""
    from uav_motion_planning_amd import adapters as A
    r, n, M, h_mid = 3, 24, 6, 0.15
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b, config_index=3)
    ref = A.refine_with_mid_knots(b["seg_offsets"], b["waypoints"], b["times"], lo, hi, k_mid=2, mid_half_width=h_mid)
    so2 = ref["seg_offsets"]
    coef, st, it = gpu_ctx.solve_corridor_batch_host(r, so2, ref["waypoints"], ref["times"], b["bc"], ref["corr_lo"], ref["corr_hi"])
    assert np.all(st == U.UAVQP_SOLVED)
    M2 = 3 * M
    c = coef.reshape(n, 3, M2, 2 * r)
    knots = c[:, :, 1:, 0]                                            # positions at interior knots 1..M2-1: [n,3,M2-1]
    lo2 = ref["corr_lo"].reshape(n, M2 + 1, 3)[:, 1:M2].transpose(0, 2, 1)
    hi2 = ref["corr_hi"].reshape(n, M2 + 1, 3)[:, 1:M2].transpose(0, 2, 1)
    assert np.all(knots >= lo2 - 1e-9) and np.all(knots <= hi2 + 1e-9)
    tight = np.isclose(hi2 - lo2, 2 * h_mid)
    assert (np.isclose(knots, lo2, atol=1e-9) | np.isclose(knots, hi2, atol=1e-9))[tight].sum() > 10   # mids really bind
    for k in range(0, n, 5):
        for ax in range(3):
            prim, stat, comp = kkt_certificate(oracle, r, M2, ref["times"][k * M2:(k + 1) * M2], c[k, ax].ravel(),
                                               ref["waypoints"].reshape(n, M2 + 1, 3)[k, :, ax], b["bc"][k, 0, :, ax], b["bc"][k, 1, :, ax],
                                               lo2[k, ax], hi2[k, ax])
            assert prim < 1e-9 and stat < 1e-7 and comp < 1e-6


@pytest.mark.parametrize("r,M", [(3, 10), (4, 6), (4, 22), (3, 30)])
def test_corridor_stress_time_allocation_and_tiny_boxes(gpu_ctx, oracle, r, M):
    """Stress: durations in [0.2, 5] s (block entries spanning T^-7..T^-1) and boxes from 1 mm to 1 m, some degenerate.
    Every result must be feasible and pass the optimality certificate of the reference-formulation matrices
    (stationarity tolerance 1e-6 here: the raw KKT condition numbers reach 1e13 for this allocation)."""
    n = 48
    b = W.uniform_batch(3, n, M, r, time_mode="wide")
    rng = np.random.default_rng(123)
    wp = b["waypoints"]
    h = 10.0 ** rng.uniform(-3, 0, size=wp.shape)
    h[rng.random(size=wp.shape) < 0.1] = 0.0
    lo, hi = wp - h, wp + h
    got, st, it = gpu_ctx.solve_corridor_batch_host(r, None, wp, b["times"], b["bc"], lo, hi, uniform_segments=M)
    assert np.all(st == U.UAVQP_SOLVED), np.unique(st, return_counts=True)
    g = got.reshape(n, 3, 2 * r * M)
    worst = np.zeros(3)
    for k in range(n):
        for ax in range(3):
            prim, stat, comp = kkt_certificate(oracle, r, M, b["times"][k], g[k, ax], wp[k, :, ax], b["bc"][k, 0, :, ax],
                                               b["bc"][k, 1, :, ax], lo[k, 1:M, ax], hi[k, 1:M, ax])
            worst = np.maximum(worst, [prim, stat, comp])
    assert worst[0] < 1e-9 and worst[1] < 1e-6 and worst[2] < 1e-5, worst


def test_cold_start_guess_changes_iterations_not_results(gpu_ctx):
    """uavqp_settings.corridor_initial_guess: 1 = the closed-form starting set of a cold solve (knots whose boxes the end-state
    polynomial misses), 2 (default) = the set the position-space dual method ends with (qp_corridor_dual.h).  Any starting set is
    admissible -- the minimiser and the reported working set are the same to the last bit whichever is used; on config-3-like
    problems 1 saves iterations on average and 2 leaves (almost always) the one solve that verifies its set."""
    import torch
    r, M, n = 3, 16, 600
    b = W.uniform_batch(3, n, M, r, time_mode="distance")
    lo, hi = W.corridor_boxes(b, config_index=3)
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_wp, d_T, d_bc, d_lo, d_hi = up(b["waypoints"].reshape(-1, 3)), up(b["times"].reshape(-1)), up(b["bc"]), up(lo.reshape(-1, 3)), up(hi.reshape(-1, 3))
    res = {}
    assert gpu_ctx.get_settings().corridor_initial_guess == 2
    try:
        for g in (2, 1, 0):
            gpu_ctx.set_settings(corridor_initial_guess=g)
            out = torch.zeros(n * 3 * M * 2 * r, dtype=torch.float64, device=dev)
            st = torch.zeros(n, dtype=torch.int32, device=dev)
            it = torch.zeros(n, dtype=torch.int32, device=dev)
            act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
            gpu_ctx.solve_corridor_device(r, n, M, M, None, d_wp, d_T, d_bc, d_lo, d_hi, out, st, it, act, False)
            gpu_ctx.synchronize()
            res[g] = (out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy())
    finally:
        gpu_ctx.set_settings(corridor_initial_guess=2)
    assert np.all(res[1][1] == U.UAVQP_SOLVED) and np.all(res[0][1] == U.UAVQP_SOLVED) and np.all(res[2][1] == U.UAVQP_SOLVED)
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][3], res[0][3])
    assert np.array_equal(res[2][0], res[0][0]) and np.array_equal(res[2][3], res[0][3])
    assert res[1][2].mean() < res[0][2].mean()
    assert res[2][2].mean() < 1.2 and res[2][2].max() <= 4, (res[2][2].mean(), res[2][2].max())


@pytest.mark.gpu
@pytest.mark.parametrize("r,m_lo,m_hi", [(3, 2, 17), (4, 2, 17), (4, 4, 24), (3, 18, 33), (4, 25, 33), (4, 30, 40)])
def test_dual_prelude_on_ragged_batches(gpu_ctx, r, m_lo, m_hi):
    """The starting set of qp_corridor_dual.h on ragged batches -- all three group shapes (8 lanes x 16 rows, 16 x 24, 16 x 32) and
    the fall-back beyond 33 segments -- against the primal method from its closed-form set: same coefficients and working sets
    bit for bit, and (where the dual method runs) barely more than the one verifying solve per problem.  Equality rows (lo == hi)
    and wide-open boxes are mixed in."""
    import torch
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    n = 300
    b = W.ragged_batch(5, n, r, m_lo=m_lo, m_hi=m_hi, seed=1234 + r + m_hi)
    so = b["seg_offsets"]
    lo, hi = W.corridor_boxes(b, config_index=5)
    rng = np.random.default_rng(5)
    rows = lo.shape[0]
    eq = rng.random(rows) < 0.1
    lo[eq] = hi[eq] = 0.5 * (lo[eq] + hi[eq])
    wide = rng.random(rows) < 0.1
    lo[wide] = -np.inf
    hi[wide] = np.inf
    d_so, d_wp, d_T, d_bc, d_lo, d_hi = up(so), up(b["waypoints"]), up(b["times"]), up(b["bc"]), up(lo), up(hi)
    nco = int(so[-1]) * 6 * r
    res = {}
    try:
        for g in (2, 1):
            gpu_ctx.set_settings(corridor_initial_guess=g)
            out = torch.zeros(nco, dtype=torch.float64, device=dev)
            st = torch.zeros(n, dtype=torch.int32, device=dev)
            it = torch.zeros(n, dtype=torch.int32, device=dev)
            act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
            gpu_ctx.solve_corridor_device(r, n, 0, m_hi, d_so, d_wp, d_T, d_bc, d_lo, d_hi, out, st, it, act, False)
            gpu_ctx.synchronize()
            res[g] = (out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy())
    finally:
        gpu_ctx.set_settings(corridor_initial_guess=2)
    assert np.array_equal(res[2][1], res[1][1])
    ok = res[1][1] == U.UAVQP_SOLVED
    assert ok.mean() > 0.9
    assert np.array_equal(res[2][3][ok], res[1][3][ok])
    co = np.repeat(ok, np.diff(so) * 6 * r)
    assert np.array_equal(res[2][0][co], res[1][0][co])
    if m_hi <= 33:
        assert res[2][2][ok].mean() < 1.3 and res[2][2][ok].mean() < 0.5 * res[1][2][ok].mean(), (res[2][2][ok].mean(), res[1][2][ok].mean())
    else:
        assert np.array_equal(res[2][2], res[1][2])


@pytest.mark.gpu
@pytest.mark.parametrize("r,m_hi", [(3, 12), (4, 22), (3, 30)])
def test_validation_inside_the_dual_prelude_matches_the_prep_kernel(gpu_ctx, r, m_hi):
    """With the dual prelude in front (corridor_initial_guess = 2) there is no corridor_reset_kernel / corridor_prep_kernel launch: the
    prelude resets, validates and describes the trajectories it visits and emits the one-segment ones.  Same verdicts and the same
    bytes as the two-kernel path (guess = 1) on a ragged batch with one-segment trajectories, a zero / negative / NaN / infinite
    duration, a box with lo > hi on ONE axis, a NaN bound, and a segment count beyond max_segments: invalid trajectories keep the
    sentinel the output buffer was filled with, on every axis."""
    import torch
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    n = 240
    b = W.ragged_batch(5, n, r, m_lo=1, m_hi=m_hi, seed=77 + r + m_hi)
    so = b["seg_offsets"]
    Ms = np.diff(so)
    assert (Ms == 1).sum() >= 2
    times = np.array(b["times"], dtype=np.float64)
    lo, hi = W.corridor_boxes(b, config_index=5)
    multi = np.flatnonzero(Ms >= 3)
    bad_T = {multi[0]: 0.0, multi[1]: -1.0, multi[2]: np.nan, multi[3]: np.inf}
    for k, v in bad_T.items():
        times[so[k] + Ms[k] // 2] = v
    k_box, k_nan = multi[4], multi[5]
    row = so[k_box] + k_box + 1                       # first interior knot of that trajectory
    lo[row, 1], hi[row, 1] = 1.0, -1.0                # one axis only
    lo[so[k_nan] + k_nan + 1, 2] = np.nan
    one = np.flatnonzero(Ms == 1)
    times[so[one[0]]] = -2.0                          # an invalid one-segment trajectory
    max_seg = m_hi - 1                                # the longest trajectories are beyond max_segments: invalid
    too_long = np.flatnonzero(Ms > max_seg)
    assert too_long.size >= 1
    d_so, d_wp, d_T, d_bc, d_lo, d_hi = up(so), up(b["waypoints"]), up(times), up(b["bc"]), up(lo), up(hi)
    nco = int(so[-1]) * 6 * r
    res = {}
    try:
        for g in (2, 1):
            gpu_ctx.set_settings(corridor_initial_guess=g)
            out = torch.full((nco,), -777.0, dtype=torch.float64, device=dev)
            st = torch.full((n,), 99, dtype=torch.int32, device=dev)
            it = torch.full((n,), 99, dtype=torch.int32, device=dev)
            act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
            gpu_ctx.solve_corridor_device(r, n, 0, max_seg, d_so, d_wp, d_T, d_bc, d_lo, d_hi, out, st, it, act, False)
            gpu_ctx.synchronize()
            res[g] = (out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy())
    finally:
        gpu_ctx.set_settings(corridor_initial_guess=2)
    st2 = res[2][1]
    invalid = sorted(list(bad_T) + [k_box, k_nan, one[0]] + list(too_long))
    assert np.all(st2[invalid] == U.UAVQP_INVALID_INPUT)
    assert np.all(np.delete(st2, invalid) != U.UAVQP_INVALID_INPUT) and (st2 == U.UAVQP_SOLVED).mean() > 0.8
    assert np.array_equal(st2, res[1][1])
    co = res[2][0].reshape(-1)
    for k in invalid:                                  # untouched, all three axes
        assert np.all(co[so[k] * 6 * r:so[k + 1] * 6 * r] == -777.0), k
    for k in one[1:]:                                  # one-segment trajectories: emitted (by the prelude / by the prep kernel)
        assert np.all(co[so[k] * 6 * r:so[k + 1] * 6 * r] != -777.0), k
    ok = st2 == U.UAVQP_SOLVED
    sel = np.repeat(ok | (st2 == U.UAVQP_INVALID_INPUT), Ms * 6 * r)
    assert np.array_equal(res[2][0][sel], res[1][0][sel])           # bit for bit
    assert np.array_equal(res[2][3][ok], res[1][3][ok])
    assert np.all(res[2][2][st2 == U.UAVQP_INVALID_INPUT] == res[1][2][st2 == U.UAVQP_INVALID_INPUT])


@pytest.mark.parametrize("r,ragged", [(3, False), (4, False), (3, True)])
def test_one_lane_per_trajectory_prelude_gives_the_same_results(gpu_ctx, r, ragged):
    """uavqp_settings.corridor_prelude_lanes = 1: the dual prelude with one lane per trajectory and a single-precision tableau in
    lane-private LDS (qp_corridor_lane.h) instead of groups of eight lanes (qp_corridor_dual.h).  Neither decides a result: coefficients,
    statuses and working sets are bit-identical; the rounding of the float tableau may cost a near-degenerate problem one more verifying
    solve (config 3: 5 of 196 608 problems), never more.  Validation duties included: a bad duration, a bad box, a single segment."""
    import torch
    if not U.has_experiments():
        # the default library carries no experimental kernels: the setting must be refused, not silently ignored
        with pytest.raises(U.UavqpError):
            gpu_ctx.set_settings(corridor_prelude_lanes=1)
        assert gpu_ctx.get_settings().corridor_prelude_lanes == 0
        pytest.skip("library built without -DUAVQP_EXPERIMENTS (make -C uav_motion_planning_amd/csrc experiments)")
    n = 700
    if ragged:
        b = W.ragged_batch(5, n, r, m_lo=1, m_hi=16, seed=99)
        uni, mx = 0, 16
    else:
        b = W.uniform_batch(3, n, 16, r, time_mode="distance")
        uni, mx = 16, 16
    so = np.asarray(b["seg_offsets"])
    lo, hi = W.corridor_boxes(b, config_index=3)
    lo, hi = np.asarray(lo).reshape(-1, 3).copy(), np.asarray(hi).reshape(-1, 3).copy()
    T = np.asarray(b["times"]).reshape(-1).copy()
    T[so[5]] = -1.0                                            # an invalid duration
    k6 = int(so[6]) + 6 + 1
    if so[7] - so[6] >= 2:
        lo[k6, 1], hi[k6, 1] = 1.0, -1.0                        # lo > hi on one axis of one interior knot
    k9 = int(so[9]) + 9 + 1
    if so[10] - so[9] >= 2:
        lo[k9] = hi[k9]                                        # an equality row
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d = [None if uni else up(so.astype(np.int32)), up(np.asarray(b["waypoints"]).reshape(-1, 3)), up(T), up(b["bc"]), up(lo), up(hi)]
    res = {}
    try:
        for lanes in (8, 1):
            gpu_ctx.set_settings(corridor_prelude_lanes=lanes)
            out = torch.full((int(so[-1]) * 6 * r,), 7.0, dtype=torch.float64, device=dev)
            st = torch.zeros(n, dtype=torch.int32, device=dev)
            it = torch.zeros(n, dtype=torch.int32, device=dev)
            act = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
            gpu_ctx.solve_corridor_device(r, n, uni, mx, d[0], d[1], d[2], d[3], d[4], d[5], out, st, it, act, False)
            gpu_ctx.synchronize()
            res[lanes] = (out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy())
    finally:
        gpu_ctx.set_settings(corridor_prelude_lanes=0)
    assert np.array_equal(res[8][0], res[1][0]) and np.array_equal(res[8][1], res[1][1]) and np.array_equal(res[8][3], res[1][3])
    assert res[8][1][5] == U.UAVQP_INVALID_INPUT and (res[8][1] == U.UAVQP_SOLVED).sum() >= n - 2
    assert res[1][2].max() <= 3 and res[1][2][res[1][1] == U.UAVQP_SOLVED].mean() < 1.05 and res[8][2].max() <= 1
