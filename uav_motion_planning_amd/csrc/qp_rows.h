// qp_rows.h -- GENERAL inequality rows on device (north-star extension; the reference hands any l <= A x <= u to OSQP,
// minimum_control.cpp:146-147,164-180, but only ever builds equality rows, :98-125).
//
// On top of the corridor boxes lo <= p_i(T_i) <= hi at the knots (qp_corridor.h) every segment carries up to K rows
//     lo <= p_i^(d)(tau T_i) <= hi          d = 0 (position sample), 1 (velocity), 2 (acceleration)[, 3 (jerk)],  0 <= tau < 1,
// per axis -- position samples at mid-segment times (BASELINE config 3's "K = 2 mid-segment samples"), per-axis velocity /
// acceleration limits, or any other row a' c_i on the coefficients of ONE segment: in the Hermite variables a row is a linear
// functional g_l' x_i + g_r' x_{i+1} of the two end knots of its segment (row_functional below).
//
// Exact solve, no ADMM tolerance: a batched ADMM on the reduced system was prototyped first and needs 500 - 2000+ iterations for
// 1e-7 on these problems (the reduced Hessian in position space is far from an M-matrix); the QP is strictly convex, so a DUAL
// active-set method (Goldfarb-Idnani, driven by nothing but "solve for a working set") is the natural exact method once a feasible
// starting point is no longer free (a clipped waypoint satisfies boxes, not velocity rows):
//   * working set W of knot boxes and rows held at a bound, multipliers lambda(W) all of the right sign (invariant; W = the
//     equality rows to begin with);
//   * the most violated constraint p joins W; the multipliers move linearly from lambda(W) towards lambda(W + p) while p's bound
//     is approached; the first one to reach zero leaves W (one more solve), until the full step is possible;
//   * no violated constraint left: optimal.  A cap on the iterations ends with UAVQP_MAX_ITER_REACHED;
//   * p is linearly DEPENDENT on W (the solve with W + p is singular: an active row no longer sits on its bound): Goldfarb-Idnani's
//     zero-primal-step case (round 5).  One extra solve with W and the right-hand side -+c_p, all data zeroed ("direction mode") gives
//     the rates d lambda(W) / d |lambda_p| -- the coefficients of c_p in the rows of W -- and a primal direction z that must vanish.
//     The multipliers move along those rates until the first of them reaches zero: that constraint leaves and p enters for good; if
//     none ever does, (rates, 1) is a Farkas certificate: UAVQP_PRIMAL_INFEASIBLE when it passes OSQP's test at eps_prim_inf
//     (|violation of p| >= eps |dy|_inf, |H z| <~ eps |dy|_inf), UAVQP_MAX_ITER_REACHED (undecided) otherwise;
//   * a STARTING set that is singular (equality rows, a warm start) is dropped once: the iteration restarts from the empty set and
//     lets the equality rows enter like any violated constraint (a redundant duplicate never does).
// Every solve is one block-Thomas pass over the knots with blocks [x_k ; mu_(rows of segment k-1)] of size R + K: the rows'
// multipliers ride in the block of the knot that closes their segment, so the KKT matrix stays block tridiagonal; a free knot
// position, an inactive row (mu fixed at 0) and a pinned position are the same thing to the elimination -- a component with a
// known value -- exactly as pinned positions are in qp_corridor.h.  Pivot order inside a block is x (positive definite Schur
// complement) then mu (negative definite): LDL' without pivoting.
//
// One lane per (trajectory, axis); the sweep state lives in an HBM workspace [wave][knot][field][lane] (this path is about
// generality, the knot-box-only corridor solver of qp_corridor.h stays the fast path).  The Hermite solution goes to
// corridor_emit_kernel like the corridor solver's.
#pragma once
#include "qp_corridor.h"

namespace uavqp {

struct RowsArgs {
    int n_traj, uniform, max_segments, max_iter;
    double eps_prim_inf;        // acceptance margin of an infeasibility certificate (uavqp_settings.eps_prim_inf: OSQP's test, minimum_control.cpp:161)
    const int32_t* seg_offsets;
    const double* waypoints;
    const double* times;
    const double* bc;
    const double* corr_lo;      // [waypoint rows][3] knot boxes (may be null: the reference's equalities at the waypoints)
    const double* corr_hi;
    const double* row_tau;      // [segments][K] position of the row inside its segment, fraction of T in [0, 1)
    const int32_t* row_deriv;   // [segments][K] derivative order 0..R-1; < 0: slot unused
    const double* row_lo;       // [segments][K][3]
    const double* row_hi;
    double* xsol;               // [waypoint rows][3][R] Hermite solution (hand-off to corridor_emit_kernel)
    int32_t* status;            // pre-filled with UAVQP_SOLVED
    int32_t* iters;             // pre-filled with 0 (may be null)
    double* ws;                 // [wave][knot 1..M][F][lane]
    unsigned long long* active; // [n_traj][3][2 + 2 K]: knot boxes (active, upper), then per row slot (active, upper); may be null
    const unsigned long long* warm;   // [n_traj][3][2]: initial working set of the knot boxes (the box-only solution's: uavqp.hip), may be null
    const unsigned long long* warm_rows;   // [n_traj][3][2 K]: initial working set of the rows, (active, upper) per slot (qp_rows_dual.h; pair kernel only), may be null
    unsigned int* queue;              // work counter, zeroed before the launch
};

// g_l, g_r with  p^(d)(tau T) = g_l' x_k + g_r' x_{k+1}   (x = derivatives 0..R-1 at the two end knots of the segment).
// From segment_coeffs: c_m = x_k[m] / m! (m < R), c_{R+j} = T^-(R+j) sum_d' K[j][d'] e[d'],  e = s1 - C s0, s*[d'] = T^d' x*[d']:
//   w[d']  = T^-d sum_j (R+j)!/(R+j-d)! tau^(R+j-d) K[j][d']
//   g_r[d'] = T^d' w[d'],      g_l[k'] = [k' >= d] k'!/(k'-d)! (tau T)^(k'-d) / k'!  -  T^k' sum_{d' <= k'} w[d'] / (k'-d')!
template <int R>
__device__ __forceinline__ void row_functional(double T, double tau, int d, double (&gl)[R], double (&gr)[R]) {
    double Tp[R];  // T^e
    Tp[0] = 1.0;
#pragma unroll
    for (int e = 1; e < R; ++e) Tp[e] = Tp[e - 1] * T;
    const double it = fast_rcp(T);
    double itd = 1.0;  // T^-d
    for (int e = 0; e < d; ++e) itd *= it;
    double w[R];
#pragma unroll
    for (int dp = 0; dp < R; ++dp) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            double f = 1.0;  // (R+j)! / (R+j-d)!
            for (int q = 0; q < d; ++q) f *= (double)(R + j - q);
            double tp = 1.0;  // tau^(R+j-d)
            for (int q = 0; q < R + j - d; ++q) tp *= tau;
            acc += f * tp * Tab<R>::K(j, dp);
        }
        w[dp] = acc * itd;
    }
#pragma unroll
    for (int dp = 0; dp < R; ++dp) gr[dp] = Tp[dp] * w[dp];
#pragma unroll
    for (int kp = 0; kp < R; ++kp) {
        double mono = 0.0;
        if (kp >= d) {
            double f = 1.0;  // k'! / (k'-d)!
            for (int q = 0; q < d; ++q) f *= (double)(kp - q);
            double tp = 1.0;  // (tau T)^(k'-d)
            for (int q = 0; q < kp - d; ++q) tp *= tau * T;
            mono = f * tp;
        }
        double s = 0.0;
#pragma unroll
        for (int dp = 0; dp <= kp; ++dp) s += w[dp] * inv_fact(kp - dp);
        gl[kp] = mono * inv_fact(kp) - Tp[kp] * s;
    }
}

template <int R, int K>
__global__ __launch_bounds__(64, 1) void rows_solve_kernel(RowsArgs a) {
    constexpr int ND = R - 1, B = R + K, NL = B * (B + 1) / 2, NCN = 1 + K;  // constraints per block: the knot box + K rows
    constexpr int F_L = 0, F_Y = NL, F_LC = NL + B, F_LN = NL + B + NCN, F = NL + B + 2 * NCN;
    constexpr int NONE = 1 << 30;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int kmax = a.uniform > 0 ? a.uniform : a.max_segments;   // blocks 1..M
    double* ws = a.ws + (size_t)(slot >> 6) * (size_t)kmax * F * 64 + (slot & 63);
    auto Wf = [&](int k, int f) -> double& { return ws[((size_t)(k - 1) * F + f) * 64]; };   // block k = 1..M

    // Dynamic dealing: a lane that has finished its problem takes the next one from a global counter at once (every lane runs
    // ONE active-set iteration per trip of the outer loop), so a wave does not run as long as the slowest of 64 problems per
    // round -- iteration counts spread from 1 to 80+ (config 3 with K = 2 rows: 25 mean, 82 max).
    const long long total = (long long)a.n_traj * 3;
    long long g = -1;            // problem = 3 * trajectory + axis, -1: none
    bool exhausted = false;      // the counter has run past the last problem
    int b = 0, ax = 0, s0 = 0, M = 0;
    long long base3 = 0;
    const double* wp = a.waypoints;
    const double* T = a.times;
    const double* bc = a.bc;
    auto klo = [&](int k) -> double { return a.corr_lo ? a.corr_lo[base3 + 3 * k] : wp[3 * k]; };
    auto khi = [&](int k) -> double { return a.corr_hi ? a.corr_hi[base3 + 3 * k] : wp[3 * k]; };
    auto rdv = [&](int s, int j) -> int { return a.row_deriv ? a.row_deriv[(size_t)(s0 + s) * K + j] : -1; };
    auto rta = [&](int s, int j) -> double { return a.row_tau[(size_t)(s0 + s) * K + j]; };
    auto rlo = [&](int s, int j) -> double { return a.row_lo[((size_t)(s0 + s) * K + j) * 3 + ax]; };
    auto rhi = [&](int s, int j) -> double { return a.row_hi[((size_t)(s0 + s) * K + j) * 3 + ax]; };
    unsigned long long eqmask = 0ull, pin = 0ull, upper = 0ull;
    unsigned long long rused[K], req[K], ract[K], rup[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { rused[j] = 0ull; req[j] = 0ull; ract[j] = 0ull; rup[j] = 0ull; }
    double x0[R], xM[R];
#pragma unroll
    for (int c = 0; c < R; ++c) { x0[c] = 0.0; xM[c] = 0.0; }
    int it = 0;
    bool done = false, capped = false;
    double tpend = 1.0;
    unsigned long long ppin = 0ull, pract[K];
#pragma unroll
    for (int j = 0; j < K; ++j) pract[j] = 0ull;
    int new_kind = -1, new_idx = -1;
    // direction mode (the entering constraint depends on the working set) -- see the header
    bool dirm = false, restarted = false;
    int fail = 0;                // status of a problem that ends without a solution (0: none)
    bool pend_inf = false;       // a Farkas certificate was found: its margin is taken in the closing solve (violation of q at the minimiser for W)
    double dyn_keep = 1.0;       // |dy|_inf of that certificate
#ifdef UAVQP_ROWS_REASON         // probe build: why a problem ended without a solution -> iters += 1000 * reason
    int reason = 0;
#define ROWS_REASON(x) reason = (x)
#else
#define ROWS_REASON(x) do {} while (0)
#endif

    for (;;) {
      if (g < 0 && !exhausted) {
        const long long q = (long long)atomicAdd(a.queue, 1u);
        if (q >= total) {
            exhausted = true;
        } else {
        g = q;
        b = (int)(g / 3);
        ax = (int)(g - 3LL * b);
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        base3 = 3LL * ((long long)s0 + b) + ax;
        wp = a.waypoints + base3;
        T = a.times + s0;
        bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
        // ---- validation, permanent (equality) constraints
        bool ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;   // working-set masks are 64-bit
        if (ok)
            for (int i = 0; i < M; ++i) ok = ok && (T[i] > 0.0) && (T[i] < INFINITY);
        eqmask = 0ull; pin = 0ull; upper = 0ull;
#pragma unroll
        for (int j = 0; j < K; ++j) { rused[j] = 0ull; req[j] = 0ull; ract[j] = 0ull; rup[j] = 0ull; }
        if (ok) {
            for (int k = 1; k < M; ++k) {
                const double l = klo(k), h = khi(k);
                ok = ok && (l <= h);
                if (l == h) eqmask |= 1ull << k;
            }
            for (int s = 0; s < M; ++s)
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int d = rdv(s, j);
                    if (d < 0) continue;
                    const double tau = rta(s, j), l = rlo(s, j), h = rhi(s, j);
                    ok = ok && (d < R) && (tau >= 0.0) && (tau < 1.0) && (l <= h) && !(tau == 0.0 && d == 0);   // (a position row AT a knot is the knot box)
                    rused[j] |= 1ull << s;
                    if (l == h) req[j] |= 1ull << s;
                }
        }
        if (!ok) {
            atomicMin(&a.status[b], (int32_t)UAVQP_INVALID_INPUT);
            g = -1;            // takes another problem on the next trip
        } else {
        x0[0] = wp[0];
        xM[0] = wp[3 * M];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            x0[d + 1] = bc[d * 3];
            xM[d + 1] = bc[(ND + d) * 3];
        }
        pin = eqmask;
        if (a.warm) {
            // Start from the working set of the box-only problem (solved first by the corridor kernel): at its solution all box
            // multipliers have the right sign, which is what the dual method needs of a starting set.  (Any set is admissible:
            // the first solve starts every multiplier from 0, so a wrong-signed one reaches zero at t = 0 and leaves.)
            const unsigned long long valid = M >= 2 ? ((1ull << M) - 2ull) : 0ull;
            const unsigned long long w0 = a.warm[(size_t)g * 2] & valid & ~eqmask;
            pin |= w0;
            upper = a.warm[(size_t)g * 2 + 1] & w0;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) ract[j] = req[j];
        for (int k = 1; k <= M; ++k)
#pragma unroll
            for (int c = 0; c < NCN; ++c) { Wf(k, F_LC + c) = 0.0; Wf(k, F_LN + c) = 0.0; }

        // ---- dual active-set iterations
        it = 0;
        done = (M == 1);   // a single segment has no free knot: nothing to decide -- its rows can only be CHECKED
        capped = false;
        if (M == 1) {
            // the polynomial is fixed by the boundary data; a row it violates makes the problem infeasible (reported like every
            // other infeasible problem: UAVQP_MAX_ITER_REACHED, the trajectory itself is still emitted)
            bool viol = false;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (!(rused[j] & 1ull)) continue;
                double gl[R], gr[R];
                row_functional<R>(T[0], rta(0, j), rdv(0, j), gl, gr);
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < R; ++c) v += gl[c] * x0[c] + gr[c] * xM[c];
                const double l = rlo(0, j), h = rhi(0, j);
                viol = viol || (l - v > 1e-9 * (1.0 + fabs(l))) || (v - h > 1e-9 * (1.0 + fabs(h)));
            }
            capped = viol;
        }
        // what the next backward sweep does to the stored multipliers: lam_cur <- lam_cur + tpend (lam_new - lam_cur) for the
        // constraints in the masks of the PREVIOUS solve (ppin, pract), 0 for a constraint that has just joined
        tpend = 1.0;
        ppin = 0ull;
#pragma unroll
        for (int j = 0; j < K; ++j) pract[j] = 0ull;
        new_kind = -1; new_idx = -1;   // the constraint being added (kind 0: knot box, 1 + j: row slot j), not subject to the sign test
        dirm = false; restarted = false; fail = capped ? (int)UAVQP_PRIMAL_INFEASIBLE : 0; pend_inf = false; dyn_keep = 1.0;
        }      // valid problem
        }      // ticket in range
      }        // refill
      if (__ballot(g >= 0) == 0ull) {
          if (__ballot(!exhausted) == 0ull) break;
          continue;
      }
      if (g >= 0) {
        bool finish = done;    // (M == 1: nothing to iterate)

        if (!done) {
            const double dat = dirm ? 0.0 : 1.0;      // direction mode: homogeneous system (every bound, pinned value and boundary state is 0)
            // direction mode: the entering constraint q = (new_kind, new_idx), unit step of its multiplier towards the violated side:
            // H z + G_W' dmu = -+ c_q (row) / +- e_0 (box); qs = -1: upper side, +1: lower side
            const double qs = dirm ? ((new_kind == 0 ? ((upper >> new_idx) & 1ull) : ((rup[new_kind > 0 ? new_kind - 1 : 0] >> new_idx) & 1ull)) ? -1.0 : 1.0) : 0.0;
            double qgl[R], qgr[R];
#pragma unroll
            for (int c = 0; c < R; ++c) { qgl[c] = 0.0; qgr[c] = 0.0; }
            int qk = -100;              // the functional sits on knots qk (qgl) and qk + 1 (qgr)
            if (dirm) {
                if (new_kind == 0) { qk = new_idx - 1; qgr[0] = 1.0; }
                else { qk = new_idx; row_functional<R>(T[new_idx], rta(new_idx, new_kind - 1), rdv(new_idx, new_kind - 1), qgl, qgr); }
            }
            // ================= forward sweep: blocks k = 1..M =================
            {
                FullBlocks<R> sa;
                sa.build(T[0]);
                SmallLDL<B> lprev;
                LDLPack<B>::zero(lprev);   // knot 0: nothing free, h = its Hermite data
                double hprev[B];
#pragma unroll
                for (int i = 0; i < B; ++i) hprev[i] = (i < R && !dirm) ? x0[i] : 0.0;
                bool pprev = false;         // knot k-1 position pinned (k-1 >= 1)
                double zprev = 0.0;
                for (int k = 1; k <= M; ++k) {
                    const bool last = (k == M);
                    FullBlocks<R> sb;
                    if (!last) sb.build(T[k]);
                    // rows of segment k-1 (this block's mu part)
                    double gl[K][R], gr[K][R], rb[K];
                    bool racv[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        racv[j] = (ract[j] >> (k - 1)) & 1ull;
                        rb[j] = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) { gl[j][c] = 0.0; gr[j][c] = 0.0; }
                        if (racv[j]) {
                            row_functional<R>(T[k - 1], rta(k - 1, j), rdv(k - 1, j), gl[j], gr[j]);
                            rb[j] = dirm ? 0.0 : (((rup[j] >> (k - 1)) & 1ull) ? rhi(k - 1, j) : rlo(k - 1, j));
                        }
                    }
                    const bool pk = !last && ((pin >> k) & 1ull);
                    const double zc = (pk && !dirm) ? (((upper >> k) & 1ull) ? khi(k) : klo(k)) : 0.0;
                    const bool pnext = (k + 1 < M) && ((pin >> (k + 1)) & 1ull);
                    const double zn = (pnext && !dirm) ? (((upper >> (k + 1)) & 1ull) ? khi(k + 1) : klo(k + 1)) : 0.0;
                    // ---- block matrix (lower triangle), right-hand side
                    double D[B][B], rhs[B];
#pragma unroll
                    for (int i = 0; i < B; ++i) {
                        rhs[i] = 0.0;
#pragma unroll
                        for (int c = 0; c < B; ++c) D[i][c] = 0.0;
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) D[i][c] = sa.B11[i][c] + (last ? 0.0 : sb.B00(i, c));
#pragma unroll
                    for (int j = 0; j < K; ++j) {
#pragma unroll
                        for (int c = 0; c < R; ++c) D[R + j][c] = gr[j][c];
                        rhs[R + j] = rb[j];
                    }
                    if (dirm && (k == qk || k == qk + 1)) {
#pragma unroll
                        for (int i = 0; i < R; ++i) rhs[i] += qs * (k == qk ? qgl[i] : qgr[i]);
                    }
                    // known values of the neighbours: pinned position of knot k-1, pinned position of knot k+1 / the end knot
                    if (pprev) {
#pragma unroll
                        for (int i = 0; i < R; ++i) rhs[i] -= sa.B01[0][i] * zprev;
#pragma unroll
                        for (int j = 0; j < K; ++j) rhs[R + j] -= gl[j][0] * zprev;
                    }
                    if (!last) {
                        if (k + 1 == M) {
#pragma unroll
                            for (int i = 0; i < R; ++i)
#pragma unroll
                                for (int c = 0; c < R; ++c) rhs[i] -= sb.B01[i][c] * (dat * xM[c]);
                        } else if (pnext) {
#pragma unroll
                            for (int i = 0; i < R; ++i) rhs[i] -= sb.B01[i][0] * zn;
                        }
                    }
                    // known values inside the block: the end knot (all of x), a pinned position, inactive rows (mu = 0)
                    bool fx[B];
                    double vx[B];
#pragma unroll
                    for (int i = 0; i < B; ++i) { fx[i] = false; vx[i] = 0.0; }
                    if (last) {
#pragma unroll
                        for (int c = 0; c < R; ++c) { fx[c] = true; vx[c] = dat * xM[c]; }
                    } else if (pk) {
                        fx[0] = true;
                        vx[0] = zc;
                    }
#pragma unroll
                    for (int j = 0; j < K; ++j) fx[R + j] = !racv[j];
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int c = 0; c < B; ++c)
                            if (fx[c] && !fx[i]) rhs[i] -= (i >= c ? D[i][c] : D[c][i]) * vx[c];
                    // coupling to the previous block: rows = x_{k-1} (its mu part does not reach this block), masked
                    double Mp[R][B];
#pragma unroll
                    for (int c = 0; c < R; ++c) {
#pragma unroll
                        for (int i = 0; i < R; ++i) Mp[c][i] = sa.B01[c][i];
#pragma unroll
                        for (int j = 0; j < K; ++j) Mp[c][R + j] = gl[j][c];
                    }
#pragma unroll
                    for (int c = 0; c < R; ++c)
#pragma unroll
                        for (int i = 0; i < B; ++i)
                            if ((pprev && c == 0) || fx[i]) Mp[c][i] = 0.0;
                    double Ep[B][B];   // S_{k-1}^-1 [Mp ; 0]
#pragma unroll
                    for (int i = 0; i < B; ++i) {
                        double col[B];
#pragma unroll
                        for (int c = 0; c < B; ++c) col[c] = c < R ? Mp[c][i] : 0.0;
                        lprev.solve(col);
#pragma unroll
                        for (int c = 0; c < B; ++c) Ep[c][i] = col[c];
                    }
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int q = 0; q < R; ++q) {
#pragma unroll
                            for (int c = 0; c <= i; ++c) D[i][c] -= Mp[q][i] * Ep[q][c];
                            rhs[i] -= Mp[q][i] * hprev[q];
                        }
                    // known components: identity rows
#pragma unroll
                    for (int i = 0; i < B; ++i)
                        if (fx[i]) {
#pragma unroll
                            for (int c = 0; c < B; ++c) {
                                if (c <= i) D[i][c] = 0.0;
                                if (c >= i) D[c][i] = 0.0;
                            }
                            D[i][i] = 1.0;
                            rhs[i] = vx[i];
                        }
                    SmallLDL<B> ldl;
                    ldl.factor(D);
                    ldl.solve(rhs);
                    {
                        double e[NL];
                        LDLPack<B>::get(ldl, e);
#pragma unroll
                        for (int q = 0; q < NL; ++q) Wf(k, F_L + q) = e[q];
#pragma unroll
                        for (int q = 0; q < B; ++q) Wf(k, F_Y + q) = rhs[q];
                    }
#pragma unroll
                    for (int i = 0; i < B; ++i) hprev[i] = rhs[i];
                    lprev = ldl;
                    sa = sb;
                    pprev = pk;
                    zprev = zc;
                }
            }
            // ================= backward sweep k = M..1 (+ the late parts at k = 0), decisions =================
            double vmax = 0.0, vq_now = 0.0;   // most violated inactive constraint (normalised violation); violation of the constraint (new_kind, new_idx)
            int vkind = -1, vidx = NONE;
            bool vupper = false;
            double tmin = dirm ? 1e300 : 2.0;   // first multiplier to reach zero on the way to the new point (direction mode: along the dependency, unbounded)
            double hz = 0.0, dymax = 0.0;       // direction mode: |H z| (diagonal proxy) and the largest rate of a multiplier of the working set
            int tkind = -1, tidx = NONE;
            bool inconsistent = false;   // an active row is not on its bound: singular working set
            {
                double yn[B];            // block k+1
#pragma unroll
                for (int i = 0; i < B; ++i) yn[i] = 0.0;
                double gln[K][R], grn[K][R];   // functionals of the rows of segment k (block k+1's rows), all slots in use
                bool usedn[K];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    usedn[j] = false;
#pragma unroll
                    for (int c = 0; c < R; ++c) { gln[j][c] = 0.0; grn[j][c] = 0.0; }
                }
                double lamA = 0.0, magA = 0.0;   // knot (k+1)'s box multiplier without its left-hand segment
                FullBlocks<R> sn;                 // segment k (between knot k and k+1)
                for (int k = M; k >= 0; --k) {
                    double y[B];
                    FullBlocks<R> sk;             // segment k-1
                    double glk[K][R], grk[K][R];
                    bool usedk[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        usedk[j] = false;
#pragma unroll
                        for (int c = 0; c < R; ++c) { glk[j][c] = 0.0; grk[j][c] = 0.0; }
                    }
                    if (k >= 1) {
                        sk.build(T[k - 1]);
#pragma unroll
                        for (int j = 0; j < K; ++j)
                            if ((rused[j] >> (k - 1)) & 1ull) {
                                usedk[j] = true;
                                row_functional<R>(T[k - 1], rta(k - 1, j), rdv(k - 1, j), glk[j], grk[j]);
                            }
                        // ---- y_k = h_k - S_k^-1 (M_k y_{k+1})
                        double h[B], t[B];
#pragma unroll
                        for (int i = 0; i < B; ++i) { h[i] = Wf(k, F_Y + i); t[i] = 0.0; }
                        if (k < M) {
                            const bool pk = (pin >> k) & 1ull;
                            const bool nlast = (k + 1 == M);
                            const bool pn = !nlast && ((pin >> (k + 1)) & 1ull);
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                if (pk && i == 0) continue;
                                double acc = 0.0;
                                if (!nlast) {
#pragma unroll
                                    for (int c = 0; c < R; ++c)
                                        if (!(pn && c == 0)) acc += sn.B01[i][c] * yn[c];
                                }
#pragma unroll
                                for (int j = 0; j < K; ++j)
                                    if ((ract[j] >> k) & 1ull) acc += gln[j][i] * yn[R + j];
                                t[i] = acc;
                            }
                            SmallLDL<B> ldl;
                            double e[NL];
#pragma unroll
                            for (int q = 0; q < NL; ++q) e[q] = Wf(k, F_L + q);
                            LDLPack<B>::set(ldl, e);
                            ldl.solve(t);
                        }
#pragma unroll
                        for (int i = 0; i < B; ++i) y[i] = h[i] - t[i];
#pragma unroll
                        for (int i = 0; i < B; ++i) Wf(k, F_Y + i) = y[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < B; ++i) y[i] = i < R ? dat * x0[i] : 0.0;
                    }
                    // ---- late parts, now that x_k is known: knot k+1's box multiplier, the values of the rows of segment k
                    if (k + 1 <= M - 1) {
                        const int kj = k + 1;
                        double lam = lamA, mag = magA;
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const double t1 = sn.B01[c][0] * y[c];
                            lam += t1;
                            mag += fabs(t1);
                        }
                        const bool pj = (pin >> kj) & 1ull, ej = (eqmask >> kj) & 1ull, uj = (upper >> kj) & 1ull;
                        // (direction mode: the injected right-hand side is no part of the multiplier; lam is then the RATE of this multiplier)
                        if (dirm) lam -= qs * (kj == qk ? qgl[0] : (kj == qk + 1 ? qgr[0] : 0.0));
                        // stored multipliers of this constraint: previous lam_new -> lam_cur by the pending interpolation
                        double lc = Wf(kj, F_LC), ln = Wf(kj, F_LN);
                        const bool was = (ppin >> kj) & 1ull;
                        lc = was ? (tpend == 0.0 ? lc : lc + tpend * (ln - lc)) : 0.0;
                        const double lnew = dirm ? lc + lam : lam;
                        if (pj && !ej) {
                            // lower bound active: need lam >= 0, upper: lam <= 0 (lam = d cost / d p, up to the factor 2)
                            const double bad = uj ? lam : -lam;
                            if (bad > (dirm ? 1e-9 + 1e-11 * mag : 1e-13 * mag) && !(new_kind == 0 && new_idx == kj)) {
                                const double den = lc - lnew;
                                double t = den != 0.0 ? lc / den : 0.0;
                                t = t < 0.0 ? 0.0 : ((t > 1.0 && !dirm) ? 1.0 : t);
                                if (t < tmin || (t == tmin && tkind == 0 && kj < tidx)) { tmin = t; tkind = 0; tidx = kj; }
                            }
                        }
                        if (dirm && pj) dymax = fmax(dymax, fabs(lam));
                        Wf(kj, F_LC) = pj ? lc : 0.0;
                        Wf(kj, F_LN) = pj ? lnew : ((dirm && new_kind == 0 && new_idx == kj) ? qs : 0.0);     // (the entering box: unit rate)
                    }
                    if (k <= M - 1) {
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            if (!usedn[j]) continue;
                            double v = 0.0;
#pragma unroll
                            for (int c = 0; c < R; ++c) v += gln[j][c] * y[c] + grn[j][c] * (k + 1 == M ? dat * xM[c] : yn[c]);
                            if ((ract[j] >> k) & 1ull) {
                                // a row of the working set must sit ON its bound.  If it does not, the working set's KKT system is
                                // singular to working precision: its rows are (numerically) dependent on the free unknowns -- an
                                // infeasible or degenerate problem (e.g. a position sample right behind the fixed start state,
                                // which no free derivative can move).  The solve is then worthless; the iteration ends as the
                                // iteration cap does (found by tools/soak_rows.py: such a problem came back "solved").
                                const double bnd = dirm ? 0.0 : (((rup[j] >> k) & 1ull) ? rhi(k, j) : rlo(k, j));
                                if (!(fabs(v - bnd) <= 1e-8 * (1.0 + fabs(bnd)))) inconsistent = true;
                                continue;
                            }
                            if (dirm) continue;      // (no constraint is looked for: the one that enters is known)
                            const double l = rlo(k, j), h = rhi(k, j);
                            const double below = l - v, above = v - h;
                            const double viol = (below > above ? below : above);
                            const double sc = viol / (1.0 + fabs(below > above ? l : h));
                            if (sc > 1e-12 && (sc > vmax || (sc == vmax && (1 + j < vkind || (1 + j == vkind && k < vidx))))) {
                                vmax = sc; vkind = 1 + j; vidx = k; vupper = above > below;
                            }
                            if (new_kind == 1 + j && new_idx == k) vq_now = viol;
                        }
                    }
                    // ---- this block's own constraints
                    if (k >= 1) {
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const int s = k - 1;
                            const bool aj = (ract[j] >> s) & 1ull, ej = (req[j] >> s) & 1ull, uj = (rup[j] >> s) & 1ull;
                            double lc = Wf(k, F_LC + 1 + j), ln = Wf(k, F_LN + 1 + j);
                            const bool was = (pract[j] >> s) & 1ull;
                            lc = was ? (tpend == 0.0 ? lc : lc + tpend * (ln - lc)) : 0.0;
                            const double mu = y[R + j];                   // (direction mode: the RATE of this multiplier)
                            const double lnew = dirm ? lc + mu : mu;
                            if (aj && !ej) {
                                // stationarity H x + G' mu = 0: lower bound active needs mu <= 0, upper mu >= 0
                                const double bad = uj ? -mu : mu;
                                if (bad > (dirm ? 1e-9 : 1e-13 * (fabs(lc) + fabs(mu))) && bad > 0.0 && !(new_kind == 1 + j && new_idx == s)) {
                                    const double den = lc - lnew;
                                    double t = den != 0.0 ? lc / den : 0.0;
                                    t = t < 0.0 ? 0.0 : ((t > 1.0 && !dirm) ? 1.0 : t);
                                    if (t < tmin || (t == tmin && (tkind < 0 || 1 + j < tkind || (1 + j == tkind && s < tidx)))) { tmin = t; tkind = 1 + j; tidx = s; }
                                }
                            }
                            if (dirm && aj) dymax = fmax(dymax, fabs(mu));
                            Wf(k, F_LC + 1 + j) = aj ? lc : 0.0;
                            Wf(k, F_LN + 1 + j) = aj ? lnew : ((dirm && new_kind == 1 + j && new_idx == s) ? -qs : 0.0);     // (the entering row: unit rate)
                        }
                        if (k < M) {
                            // first part of knot k's box multiplier: row 0 of its own block and of the right-hand segment
                            lamA = 0.0;
                            magA = 0.0;
#pragma unroll
                            for (int c = 0; c < R; ++c) {
                                const double d0 = sk.B11[0][c] + sn.B00(0, c);
                                const double t2 = d0 * y[c], t3 = sn.B01[0][c] * (k + 1 == M ? dat * xM[c] : yn[c]);
                                lamA += t2 + t3;
                                magA += fabs(t2) + fabs(t3);
                            }
#pragma unroll
                            for (int j = 0; j < K; ++j) {
                                if ((ract[j] >> (k - 1)) & 1ull) { const double t4 = grk[j][0] * y[R + j]; lamA += t4; magA += fabs(t4); }
                                if ((ract[j] >> k) & 1ull) { const double t5 = gln[j][0] * yn[R + j]; lamA += t5; magA += fabs(t5); }
                            }
                            if (dirm) {
                                // |H z|_inf by its diagonal part: the primal direction of a dependent constraint must vanish
#pragma unroll
                                for (int c = 0; c < R; ++c) hz = fmax(hz, fabs((sk.B11[c][c] + sn.B00(c, c)) * y[c]));
                            }
                            // free knot position outside its box?
                            if (!dirm && !((pin >> k) & 1ull)) {
                                const double l = klo(k), h = khi(k), v = y[0];
                                const double below = l - v, above = v - h;
                                const double viol = below > above ? below : above;
                                const double sc = viol / (1.0 + fabs(below > above ? l : h));
                                if (sc > 1e-12 && (sc > vmax || (sc == vmax && (0 < vkind || k < vidx)))) { vmax = sc; vkind = 0; vidx = k; vupper = above > below; }
                                if (new_kind == 0 && new_idx == k) vq_now = viol;
                            }
                        }
                    }
                    // ---- shift: block k becomes "next"
#pragma unroll
                    for (int i = 0; i < B; ++i) yn[i] = y[i];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        usedn[j] = usedk[j];
#pragma unroll
                        for (int c = 0; c < R; ++c) { gln[j][c] = glk[j][c]; grn[j][c] = grk[j][c]; }
                    }
                    sn = sk;
                }
            }
            ++it;
            // ================= dual active-set step =================
            ppin = pin;
#pragma unroll
            for (int j = 0; j < K; ++j) pract[j] = ract[j];
            if (capped) {
                finish = true;   // the last solve (for the working set it stopped with) is what is handed over
                // (a pending certificate: accepted as OSQP accepts one -- its support-function value, the violation of q at this minimiser, is at
                // least eps_prim_inf |dy|_inf; below that margin the verdict stays "undecided")
                if (pend_inf && vq_now >= a.eps_prim_inf * dyn_keep) fail = (int)UAVQP_PRIMAL_INFEASIBLE;
            } else if (dirm) {
                // ---- the direction-mode solve: rates of the multipliers of W along the dependency of q on W
                dirm = false;
                const unsigned long long qbit = 1ull << new_idx;
                const double dyn = fmax(1.0, dymax);                                   // |dy|_inf: the entering constraint's own rate is 1
                const bool dep = !inconsistent && hz <= fmax(a.eps_prim_inf, 1e-6) * dyn;   // (OSQP: |A' dy| <= eps |dy|; NaNs fail the test)
                if (!inconsistent && tkind >= 0) {
                    // a multiplier of W reaches zero first: that constraint leaves, q enters for good with the multiplier it has by then.
                    // (Also when z did not vanish -- W + q is regular but so ill-conditioned that its solve was worthless: the full step cannot
                    // be computed, leaving with the constraint that blocks the dual direction is the way on; what counts as SOLVED is decided by
                    // the KKT conditions of a later solve, not here.)
                    tpend = tmin;
                    if (tkind == 0) pin &= ~(1ull << tidx);
                    else ract[tkind - 1] &= ~(1ull << tidx);
                    if (new_kind == 0) { pin |= qbit; ppin |= qbit; }
                    else { ract[new_kind - 1] |= qbit; pract[new_kind - 1] |= qbit; }
                } else {
                    // none ever does: (rates, 1) is a Farkas certificate of infeasibility (a direction that did not vanish proves nothing)
                    fail = (int)UAVQP_MAX_ITER_REACHED;
                    pend_inf = dep;
                    dyn_keep = dyn;
                    ROWS_REASON(inconsistent ? 1 : (dep ? 3 : 2));
                    tpend = 0.0;
                    capped = true;      // one more solve for W: the minimiser of the last regular working set is what is handed over
                }
            } else if (inconsistent) {
                if (new_kind >= 0) {
                    // the constraint that has just entered depends on the working set: take it out again and solve for the direction
                    if (new_kind == 0) pin &= ~(1ull << new_idx);
                    else ract[new_kind - 1] &= ~(1ull << new_idx);
                    dirm = true;
                    tpend = 0.0;        // (the multipliers of W stay where they are: the solve that went singular left garbage in their "new" slots)
                } else if (!restarted) {
                    // a singular STARTING set (equality rows that depend on each other, a warm start): once more from the boxes' equalities
                    // alone; the equality rows enter like any violated constraint -- a redundant one never does
                    restarted = true;
                    pin = eqmask; upper = 0ull; ppin = 0ull;
#pragma unroll
                    for (int j = 0; j < K; ++j) { ract[j] = 0ull; rup[j] = 0ull; pract[j] = 0ull; }
                    tpend = 1.0;
                } else {
                    tpend = 1.0;
                    capped = true;
                    fail = (int)UAVQP_MAX_ITER_REACHED;
                    ROWS_REASON(4);
                }
            } else if (tkind >= 0) {
                // a multiplier of W reaches zero before the new point: it leaves, the others stop at that fraction of the way
                tpend = tmin;
                if (tkind == 0) pin &= ~(1ull << tidx);
                else ract[tkind - 1] &= ~(1ull << tidx);
            } else {
                tpend = 1.0;
                new_kind = -1;
                new_idx = -1;
                if (vkind < 0) {
                    done = true;
                } else if (vkind == 0) {
                    pin |= 1ull << vidx;
                    if (vupper) upper |= 1ull << vidx; else upper &= ~(1ull << vidx);
                    new_kind = 0; new_idx = vidx;
                } else {
                    ract[vkind - 1] |= 1ull << vidx;
                    if (vupper) rup[vkind - 1] |= 1ull << vidx; else rup[vkind - 1] &= ~(1ull << vidx);
                    new_kind = vkind; new_idx = vidx;
                }
            }
            if (!done && !finish && !capped && it >= a.max_iter) {
                // give up: re-solve once for the working set as it stands WITHOUT the constraint that was being added
                if (!dirm) {
                    if (new_kind == 0) pin &= ~(1ull << new_idx);
                    else if (new_kind > 0) ract[new_kind - 1] &= ~(1ull << new_idx);
                }
                dirm = false;
                new_kind = -1;
                capped = true;
                pend_inf = false;
                fail = (int)UAVQP_MAX_ITER_REACHED;
                ROWS_REASON(5);
            }
            if (done) finish = true;
        }

        // ================= hand-over: Hermite solution of the interior knots =================
        if (finish) {
            for (int k = 1; k < M; ++k) {
                double* o = a.xsol + (base3 + 3LL * k) * R;
#pragma unroll
                for (int c = 0; c < R; ++c) o[c] = Wf(k, F_Y + c);
            }
            if (capped) atomicMin(&a.status[b], (int32_t)(fail != 0 ? fail : (int)UAVQP_MAX_ITER_REACHED));      // (UAVQP_PRIMAL_INFEASIBLE < UAVQP_MAX_ITER_REACHED: an infeasible axis decides the trajectory; 0 is no status)
#ifdef UAVQP_ROWS_REASON
            if (capped) it += 1000 * reason;
#endif
            if (a.iters) atomicMax(&a.iters[b], (int32_t)it);
            if (a.active) {
                unsigned long long* o = a.active + (size_t)g * (2 + 2 * K);
                const unsigned long long valid = M >= 2 ? ((1ull << M) - 2ull) : 0ull;
                o[0] = pin & ~eqmask & valid;
                o[1] = upper & o[0];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    o[2 + 2 * j] = ract[j] & ~req[j];
                    o[3 + 2 * j] = rup[j] & o[2 + 2 * j];
                }
            }
            g = -1;
        }
      }   // lane has a problem
    }     // trips
}

}  // namespace uavqp
