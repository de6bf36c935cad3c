"""CPU: pins the oracle (oracle/qp_oracle.c) against exact rational known answers (tests/golden/) and
checks that its matrices are the reference's (minimum_control.cpp:5-125), entry for entry."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kkt_exact.json")


def load_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_exact_oracle_matches_rational_golden(oracle, case):
    r, M = case["r"], case["M"]
    wp = np.array(case["waypoints"])
    bc = np.array(case["bc"])
    for ax in range(3):
        c = oracle.solve_exact(r, wp[:, ax], bc[0, :, ax], bc[1, :, ax], case["times"])
        ref = np.array(case["coef"][ax])
        assert np.max(np.abs(c - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))
        assert abs(oracle.cost(r, case["times"], c) - case["half_xPx"][ax]) <= 1e-10 * max(1.0, case["half_xPx"][ax])


def test_reference_kat_exact_rationals(oracle):
    """test_qpsolve.cpp:10-17 -> BASELINE.md section 4 table."""
    c = oracle.solve_exact(3, [1, 2, 3, 4], [0, 0], [0, 0], [1, 1, 1])
    exp = np.array([1, 0, 0, 190 / 51, -65 / 17, 56 / 51,
                    2, 70 / 51, -40 / 51, -10 / 17, 5 / 3, -2 / 3,
                    3, 70 / 51, 40 / 51, -10 / 17, -5 / 3, 56 / 51])
    assert np.max(np.abs(c - exp)) < 1e-14
    assert abs(oracle.cost(3, [1, 1, 1], c) - 1560 / 17) < 1e-11


def test_hessian_literals_of_the_reference(oracle):
    """getHessian, minimum_control.cpp:9-17: 36T, 72T^2, 120T^3 / 192T^3, 360T^4 / 720T^5 on rows/cols 6i+3..5."""
    T = np.array([0.75, 1.5])
    P, _ = oracle.assemble(3, T)
    for i, t in enumerate(T):
        blk = P[6 * i + 3:6 * i + 6, 6 * i + 3:6 * i + 6]
        exp = np.array([[36 * t, 72 * t**2, 120 * t**3], [72 * t**2, 192 * t**3, 360 * t**4],
                        [120 * t**3, 360 * t**4, 720 * t**5]])
        assert np.allclose(blk, exp, rtol=1e-15, atol=0)
    mask = np.ones_like(P, dtype=bool)
    for i in range(2):
        mask[6 * i + 3:6 * i + 6, 6 * i + 3:6 * i + 6] = False
    assert np.all(P[mask] == 0.0)


def test_constraint_rows_of_the_reference(oracle):
    """getConstraintMatrix / getBound, minimum_control.cpp:26-125, M=3: 14 rows, 66 numerically non-zero entries."""
    T = np.array([1.0, 2.0, 0.5])
    _, A = oracle.assemble(3, T)
    assert A.shape == (14, 18) and oracle.dims(3, 3) == (18, 14)
    assert np.count_nonzero(A) == 66
    assert A[0, 0] == 1 and A[1, 1] == 1 and A[2, 2] == 2                       # :29-31
    t = T[0]
    assert np.allclose(A[3, :6], [1, t, t**2, t**3, t**4, t**5])               # waypoint row :36-41
    assert np.allclose(A[4, :7], [1, t, t**2, t**3, t**4, t**5, -1])           # continuity p :47-53
    assert np.allclose(A[5, :8], [0, 1, 2 * t, 3 * t**2, 4 * t**3, 5 * t**4, 0, -1])
    assert np.allclose(A[6, :9], [0, 0, 2, 6 * t, 12 * t**2, 20 * t**3, 0, 0, -2])
    t = T[2]
    assert np.allclose(A[11, 12:], [1, t, t**2, t**3, t**4, t**5])             # end position :77-82
    assert np.allclose(A[12, 12:], [0, 1, 2 * t, 3 * t**2, 4 * t**3, 5 * t**4])
    assert np.allclose(A[13, 12:], [0, 0, 2, 6 * t, 12 * t**2, 20 * t**3])
    l, u = oracle.bounds(3, [1, 2, 3, 4], [0.5, -0.25], [0.125, 0.75])
    assert np.array_equal(l, u)                                                  # every row is an equality
    assert list(l) == [1, 0.5, -0.25, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0.125, 0.75]


def test_snap_extension_dimensions(oracle):
    """r = 4: n = 8M, m = 5M + 3 (SURVEY.md section 8-a')."""
    for M in (1, 2, 7, 8, 24):
        assert oracle.dims(4, M) == (8 * M, 5 * M + 3)
    P, A = oracle.assemble(4, [1.0, 1.0])
    assert np.linalg.matrix_rank(A) == A.shape[0]
    assert P[4, 4] == 576.0 and P[7, 7] == pytest.approx(840.0**2 / 7)          # (4!)^2 T, (7!/3!)^2 T^7/7


@pytest.mark.parametrize("r,M", [(3, 4), (4, 8)])
def test_oracle_solution_properties(oracle, r, M):
    rng = np.random.default_rng(5)
    pos = np.cumsum(rng.uniform(-2, 2, M + 1))
    T = rng.uniform(0.4, 2.5, M)
    bs, be = rng.uniform(-1, 1, r - 1), rng.uniform(-1, 1, r - 1)
    c = oracle.solve_exact(r, pos, bs, be, T)
    assert oracle.residual(r, pos, bs, be, T, c) < 1e-9
    # translation: only c0 of every segment moves
    c2 = oracle.solve_exact(r, pos + 3.0, bs, be, T)
    d = (c2 - c).reshape(M, 2 * r)
    assert np.allclose(d[:, 0], 3.0, atol=1e-9) and np.max(np.abs(d[:, 1:])) < 1e-8 * max(1, np.max(np.abs(c)))
    # time scaling T -> sT with derivative BCs scaled by s^-d maps c_k -> c_k / s^k
    s = 1.7
    sc = np.array([s ** -(d + 1) for d in range(r - 1)])
    c3 = oracle.solve_exact(r, pos, bs * sc, be * sc, T * s).reshape(M, 2 * r)
    assert np.allclose(c3, c.reshape(M, 2 * r) / s ** np.arange(2 * r), rtol=1e-8, atol=1e-9)
    # optimality: feasible perturbations in null(A) never lower the cost
    P, A = oracle.assemble(r, T)
    _, _, vt = np.linalg.svd(A)
    null = vt[A.shape[0]:].T
    base = oracle.cost(r, T, c)
    for _ in range(5):
        dx = null @ rng.normal(size=null.shape[1]) * 1e-3
        assert oracle.cost(r, T, c + dx) >= base - 1e-9 * max(1, base)


def test_exact_oracle_against_fresh_rational_solves_property_based(oracle):
    """Beyond the 13 committed fixtures: hypothesis draws small dyadic problems, tests/golden/gen_golden.py solves them
    in exact rationals on the spot (its assemble() is minimum_control.cpp:5-125 entry by entry), and the binary128
    oracle must agree to 1e-12 -- a different problem set on every hypothesis database, same pinning argument."""
    import sys
    from fractions import Fraction as Fr

    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import gen_golden as G

    dy = lambda lo, hi, den: st.integers(int(lo * den), int(hi * den)).map(lambda k: Fr(k, den))

    @st.composite
    def problems(draw):
        r = draw(st.sampled_from([3, 4]))
        M = draw(st.integers(1, 3))
        T = [draw(dy(0.25, 3.0, 8)) for _ in range(M)]
        pos = [draw(dy(-4, 4, 16)) for _ in range(M + 1)]
        bcs = [draw(dy(-2, 2, 8)) for _ in range(r - 1)]
        bce = [draw(dy(-2, 2, 8)) for _ in range(r - 1)]
        return r, T, pos, bcs, bce

    @settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(problems())
    def check(p):
        r, T, pos, bcs, bce = p
        P, A, b = G.assemble(r, T, pos, bcs, bce)
        exact = np.array([float(v) for v in G.solve_kkt(P, A, b)])
        got = oracle.solve_exact(r, [float(v) for v in pos], [float(v) for v in bcs], [float(v) for v in bce], [float(t) for t in T])
        assert np.max(np.abs(got - exact)) <= 1e-12 * max(1.0, np.max(np.abs(exact)))

    check()
