"""CPU: the oracle's restatement of the reference's QP assembly against the reference's OWN source.

oracle/_ref/libref_minimum_control.so is /root/reference/src/planner/traj_optimization/src/minimum_control.cpp compiled
unmodified, from where it lies, against stand-in headers for the libraries that are absent from this image
(oracle/ref_shim/: a minimal Eigen surface, an osqp-eigen facade that records what solve() hands to the solver and solves
the all-equality QP exactly).  Built by __graft_entry__.build() / `make -C oracle ref` where /root/reference is mounted;
the prebuilt file travels to the GPU box.  Skipped where neither exists.

This pins the part of the path whose arithmetic lives in the reference repository itself -- getHessian,
getConstraintMatrix, getBound, the solver settings (minimum_control.cpp:5-125,160-162).  The ADMM iteration lives in
OSQP (absent, unpinned): parity stays "unpinned" for that part."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not (oracle.build_ref() or oracle.ref_available()):
        pytest.skip("oracle/_ref not built and /root/reference not mounted")
    return oracle


def test_reference_source_kat_matches_exact_fixture(ref):
    """test_qpsolve.cpp:10-17 through the reference's own solve(): the call succeeds, the settings are the ones SURVEY.md
    8-a8 lists (warm start, eps_prim_inf 1e-3, max_iter 1000 -- also the defaults of oracle.osqp_settings()), P carries
    BOTH triangles (27 = 9 per segment insertions, minimum_control.cpp:9-17), and the exact minimiser of the reference's
    own data is the rational table of BASELINE.md section 4."""
    r = ref.ref_solve([1, 2, 3, 4], [0, 0], [0, 0], [1, 1, 1])
    assert r["ok"] and (r["n"], r["m"]) == (18, 14)
    assert r["warm_start"] is True and r["eps_prim_inf"] == 1e-3 and r["max_iter"] == 1000
    s = ref.osqp_settings()
    assert s.max_iter == r["max_iter"] and s.eps_prim_inf == r["eps_prim_inf"]
    assert r["p_inserted"] == 27 and np.array_equal(r["P"], r["P"].T)
    exp = np.array([1, 0, 0, 190 / 51, -65 / 17, 56 / 51, 2, 70 / 51, -40 / 51, -10 / 17, 5 / 3, -2 / 3,
                    3, 70 / 51, 40 / 51, -10 / 17, -5 / 3, 56 / 51])
    assert np.max(np.abs(r["coef"] - exp)) < 1e-13
    assert np.all(r["l"] == r["u"])                      # every row an equality


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 16])
def test_oracle_assembly_equals_reference_assembly_entry_by_entry(ref, M):
    """oracle_assemble_P / _A / oracle_bounds (r = 3) vs the matrices the reference code builds for the same inputs:
    identical sparsity, values equal to the last bit (dyadic durations) or to 2 ulp (arbitrary durations: the reference
    calls pow(T, k), the restatement multiplies)."""
    rng = np.random.default_rng(100 + M)
    for dyadic in (True, False):
        T = rng.integers(2, 24, size=M) / 8.0 if dyadic else rng.uniform(0.2, 5.0, size=M)
        pos = rng.uniform(-4, 4, size=M + 1)
        vel, acc = rng.uniform(-2, 2, size=2), rng.uniform(-2, 2, size=2)
        r = ref.ref_solve(pos, vel, acc, T)
        assert r["ok"]
        P, A = ref.assemble(3, T)
        l, u = ref.bounds(3, pos, [vel[0], acc[0]], [vel[1], acc[1]])
        assert np.array_equal(P != 0, r["P"] != 0) and np.array_equal(A != 0, r["A"] != 0)
        if dyadic:
            assert np.array_equal(P, r["P"]) and np.array_equal(A, r["A"])
        else:
            assert np.allclose(P, r["P"], rtol=4e-16, atol=0) and np.allclose(A, r["A"], rtol=4e-16, atol=0)
        assert np.array_equal(l, r["l"]) and np.array_equal(u, r["u"])
        assert r["a_inserted"] >= np.count_nonzero(r["A"])           # the reference also inserts explicit zeros (:55,61,64...)
        # and the exact minimiser of the reference's own data is the oracle's
        mine = ref.solve_exact(3, pos, [vel[0], acc[0]], [vel[1], acc[1]], T)
        assert np.max(np.abs(mine - r["coef"])) <= 1e-9 * max(1.0, np.max(np.abs(mine)))


def test_reference_source_rejects_nothing_and_indexes_out_of_range_for_one_waypoint(ref):
    """with a single waypoint (no segment) the reference's getBound indexes out of range; the stand-in
    containers are bounds-checked, so the wrapper reports the throw instead of corrupting memory."""
    r = ref.ref_solve([1.0], [0, 0], [0, 0], [])
    assert r["rc"] in (-1, 0)


def test_poly_eval_restatement_equals_reference_polytraj_header(ref):
    """oracle/poly_eval.c vs the reference's own header-only PolyTraj (traj_utils/poly_traj.hpp:74-168) compiled against the
    stand-in Eigen: positions, velocities and accelerations at times that exercise the segment-search rule (inside
    segments, exactly on a boundary, boundary + 1e-4 on either side of the slack, before 0 and past the end).  1e-13
    relative: the only difference is rounding in how the monomial powers are accumulated."""
    rng = np.random.default_rng(7)
    for nc, M in ((6, 1), (6, 4), (8, 7), (8, 12)):
        T = rng.uniform(0.3, 2.5, size=M)
        c = rng.normal(size=(3, M, nc))
        edges = np.cumsum(T)
        ts = np.concatenate([np.linspace(-0.7, edges[-1] + 1.5, 150), edges, edges + 1e-4, edges + 1.0001e-4, edges + 0.9999e-4, edges - 1e-9])
        for t in ts:
            a = ref.ref_polytraj_eval(nc, T, c, float(t))
            b = ref.poly_eval(nc, T, c, float(t), 7)
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-13, (nc, M, t)


def test_traj_length_restatement_equals_reference_polytraj_header(ref):
    """oracle/poly_eval.c::oracle_traj_length vs the reference's own PolyTraj::getTraj / getLength / getMeanVel
    (traj_utils/poly_traj.hpp:175-207) compiled from where it lies: the same sample COUNT -- including total times that are
    multiples of the 0.01 s step, where the reference's floating-point accumulation of t decides whether the last sample
    exists -- and length / mean velocity to 1e-12."""
    from uav_motion_planning_amd import workloads as W
    for r, M, mode in [(3, 3, "reference"), (4, 7, "reference"), (3, 5, "distance"), (4, 8, "wide")]:
        b = W.uniform_batch(2, 6, M, r, time_mode=mode)
        coef, _ = ref.solve_exact_batch(r, b["seg_offsets"], b["waypoints"], b["times"], b["bc"])
        c = coef.reshape(6, -1)
        for k in range(6):
            la, va, na = ref.traj_length(2 * r, b["times"][k], c[k])
            lb, vb, nb = ref.ref_polytraj_length(2 * r, b["times"][k], c[k])
            assert na == nb
            assert abs(la - lb) <= 1e-12 * lb and abs(va - vb) <= 1e-12 * vb
