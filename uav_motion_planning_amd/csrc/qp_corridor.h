// qp_corridor.h -- corridor-constrained variant (north-star extension, SURVEY.md section 8-a'; BASELINE
// configs 3 and 5): the interior-waypoint equalities p_i(T_i) = w_{i+1} of the reference QP
// (minimum_control.cpp:34-42,118-124) become boxes lo <= p_i(T_i) <= hi -- the only inequality rows.
//
// In the Hermite variables the problem per axis is a strictly convex QP in x_k = (p_k, v_k, a_k[, j_k]) at
// the interior knots with an SPD block-tridiagonal Hessian (r x r blocks, function of the time
// allocation only) and simple bounds on the position components.  It is solved EXACTLY by a primal
// active-set method: pinned positions are eliminated symmetrically, every iteration is one block
// Thomas solve, a blocking bound is added on a partial step, the worst wrong-signed multiplier is
// released -- no ADMM tolerance, the result is the QP's minimiser to rounding (OSQP in the reference
// formulation converges to the same point, which is how the tests check it).
//
// One lane per (trajectory, axis): the active sets differ per axis, so the factorisation is not shared.
// Sweep state lives in an HBM workspace laid out [wave][knot][field][lane] (any segment count up to 63, ragged batches);
// uniform batches keep the records of their last knots in LDS.  Checked against exact-rational fixtures including the
// active sets (tests/test_corridor_golden.py), the OSQP-faithful port and a KKT certificate (tests/test_gpu_corridor.py).
#pragma once
#include "qp_device.h"

namespace uavqp {

struct CorridorArgs {
    int n_traj, uniform, max_segments, max_iter, pdas_rounds;
    const int32_t* seg_offsets;
    const double* waypoints;
    const double* times;
    const double* bc;
    const double* corr_lo;
    const double* corr_hi;
    double* coeff;
    int32_t* status;  // pre-filled with UAVQP_SOLVED; failing axes atomicMin their code in
    int32_t* iters;   // pre-filled with 0; atomicMax over axes (may be null)
    double* ws;
    unsigned long long* active;  // [n_traj][3][2] working set in/out (may be null)
    int warm;                    // read `active` as the initial working set
};

// r x r blocks of one segment including the position component (index 0):
//   B11 end/end = T^(a+b+1-2R) W[a][b],  B00 start/start = (-1)^(a+b) B11,  B01 start/end = -T^(a+b+1-2R) V[a][b]
template <int R>
struct FullBlocks {
    double B11[R][R];
    double B01[R][R];
    __device__ __forceinline__ void build(double T) {
        const double it = fast_rcp(T);
        double ip[2 * R];
        ip[0] = 1.0;
#pragma unroll
        for (int j = 1; j < 2 * R; ++j) ip[j] = ip[j - 1] * it;
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) {
                const double p = ip[2 * R - 1 - a - b];
                B11[a][b] = p * Tab<R>::W(a, b);
                B01[a][b] = -p * Tab<R>::V(a, b);
            }
    }
    __device__ __forceinline__ double B00(int i, int c) const { return ((i + c) & 1) ? -B11[i][c] : B11[i][c]; }
};

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// r = 3 is held to 2 waves per SIMD (256 registers, some scratch): the sweeps are chains of dependent
// operations, a second wave fills the gaps (measured on config 3: 4.30 ms at 1 wave, 3.25 ms at 2, 4.19 ms at 3).
//
// Every active-set iteration is exactly ONE forward and ONE backward pass over the knots; everything else is
// folded into them, and every HBM access of a pass is issued one knot ahead of its use:
//   * the update of the feasible iterate z decided by the previous iteration (block-pivot clip, partial step of
//     the ratio test, full step) is applied lazily in the forward sweep, one knot ahead of the elimination;
//   * the decisions of the iteration -- block-pivot sets, ratio test over the free positions, multipliers of the
//     active bounds (row 0 of the unmasked block row, finished one knot late when x_{k-1} appears) -- are
//     accumulated in the backward sweep.
// (The first version ran these as separate loops of dependent HBM round trips: 2-3x the time of the sweeps.)
template <int R>
struct SegRow0 {
    // what the multiplier of knot j needs from a segment s: e11[c] = B11_s[0][c], e01r[c] = B01_s[0][c], e01c[c] = B01_s[c][0]
    double e11[R], e01r[R], e01c[R];
};

template <int R, int LDS_KNOTS>
__global__ __launch_bounds__(64, R == 3 ? 2 : 1) void solve_corridor_kernel(CorridorArgs a) {
    constexpr int ND = R - 1, NC = 2 * R;
    // Prefetch distance is ONE knot.  Two was measured and is worse: the second record buffer pushes the r = 4 kernel
    // (256 VGPRs + AGPR spills already) over the edge -- config 5's corridor solve 2.83 ms at distance 1, 3.4 ms at 2.
    // sweep state per interior knot: LDL' factors of S_k (strict lower triangle + inverse pivots), x_k (first h_k,
    // overwritten by the solution in the backward sweep) and the current position iterate z_k.  E_k = S_k^-1 M_k is
    // NOT stored: it is re-derived from the factors where needed (the kernel is bound by this HBM traffic).
    constexpr int NL = R * (R - 1) / 2;
    constexpr int F_L = 0, F_DI = NL, F_X = NL + R, F_Z = NL + 2 * R, F = NL + 2 * R + 1;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_slots = gridDim.x * blockDim.x;
    // workspace: [wave][interior knot][field][lane] -- a wave's record of one knot is F consecutive 512-byte rows (one
    // page, not F pages a batch-stride apart: the sweeps are latency-bound, TLB and DRAM-row locality matter)
    const int kmax = (a.uniform > 0 ? a.uniform : a.max_segments) - 1;
    double* __restrict__ ws = a.ws + (size_t)(slot >> 6) * (size_t)(kmax > 1 ? kmax : 1) * F * 64 + (slot & 63);
    // The records of the last NT interior knots of a UNIFORM batch live in LDS instead (the forward sweep writes them
    // last, the backward sweep reads them first): NT of M-1 knots less HBM traffic for the bandwidth-bound large batch
    // (config 3: 4 of 15).  Ragged batches keep everything in HBM (NT = 0 instantiation): lanes of one wave would diverge on the test, and
    // the tests alone cost the latency-bound ragged case 30 % (config 5: 2.76 -> 3.61 ms when they were left in).
    constexpr int NT = LDS_KNOTS;
    __shared__ double s_rec[NT > 0 ? NT * F * 64 : 1];
    // knots k >= lds_from are in LDS; the ragged instantiation (NT = 0) compiles every test below away
    const int lds_from = NT > 0 ? a.uniform - NT : 0;
    const int lane = threadIdx.x & 63;
    auto G = [&](int k, int f) -> double& { return ws[((size_t)(k - 1) * F + f) * 64]; };        // interior knot k = 1..M-1
    auto S = [&](int k, int f) -> double& { return s_rec[((k - lds_from) * F + f) * 64 + lane]; };
    auto ld_xz = [&](int k, double& x0, double& z) {
        if (NT > 0 && k >= lds_from) { x0 = S(k, F_X); z = S(k, F_Z); } else { x0 = G(k, F_X); z = G(k, F_Z); }
    };
    auto st_z = [&](int k, double z) { if (NT > 0 && k >= lds_from) S(k, F_Z) = z; else G(k, F_Z) = z; };

    const long long total = (long long)a.n_traj * 3;
    for (long long g = slot; g < total; g += n_slots) {
        const int b = (int)(g / 3), ax = (int)(g - 3LL * b);
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        const size_t row0 = (size_t)(s0 + b);
        const double* __restrict__ wp = a.waypoints + 3 * row0 + ax;  // stride 3 per knot
        const double* __restrict__ lo = a.corr_lo + 3 * row0 + ax;
        const double* __restrict__ hi = a.corr_hi + 3 * row0 + ax;
        const double* __restrict__ T = a.times + s0;
        const double* __restrict__ bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
        double* __restrict__ out = a.coeff + (size_t)3 * NC * s0 + (size_t)ax * NC * M;

        bool ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;  // pin masks are 64-bit
        if (ok)
            for (int i = 0; i < M; ++i) ok = ok && (T[i] > 0.0) && (T[i] < INFINITY);
        if (ok)
            for (int k = 1; k < M; ++k) ok = ok && (lo[3 * k] <= hi[3 * k]);
        if (!ok) {
            atomicMin(&a.status[b], (int32_t)UAVQP_INVALID_INPUT);
            continue;
        }
        double x0[R], xM[R];
        x0[0] = wp[0];
        xM[0] = wp[3 * M];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            x0[d + 1] = bc[d * 3];
            xM[d + 1] = bc[(ND + d) * 3];
        }

        // ---- initial feasible point and permanent pins (lo == hi: a true equality row, as in the reference)
        // A warm start only supplies the first working set (bounds guessed active sit on their bound); wrong guesses
        // are repaired by the iterations below like any other intermediate working set.
        unsigned long long eqmask = 0ull, pin = 0ull, upper = 0ull;
        unsigned long long wpin = 0ull, wupper = 0ull;
        if (a.active && a.warm) {
            wpin = a.active[2 * g];
            wupper = a.active[2 * g + 1];
        }
        for (int k = 1; k < M; ++k) {
            const double l = lo[3 * k], h = hi[3 * k];
            double z = wp[3 * k];
            z = z < l ? l : (z > h ? h : z);
            if (l == h) {
                eqmask |= 1ull << k;
            } else if ((wpin >> k) & 1ull) {
                const bool up = (wupper >> k) & 1ull;
                z = up ? h : l;
                pin |= 1ull << k;
                if (up) upper |= 1ull << k;
            }
            st_z(k, z);
        }
        pin |= eqmask;

        int it = 0;
        int pdas_left = a.pdas_rounds;  // default 3 (measured on config 3: 3 rounds 13.8 mean iterations, 0 rounds 17.5, 10 rounds 15.7)
        bool converged = (M == 1);
        bool final_pass = false;  // max_iter hit: one last solve with every position pinned at the feasible iterate
        // pending update of z, applied by the next forward sweep (zmode 0: none; 1: block-pivot round; 2: partial step
        // of length zalpha blocked at knot zblock; 3: full step)
        int zmode = 0, zblock = -1;
        bool zblock_upper = false;
        unsigned long long zpin = 0ull;
        double zalpha = 1.0;
        auto znew = [&](int k, double x0old, double zold, double l, double h) -> double {
            const bool zp = (zpin >> k) & 1ull;
            const double clipped = x0old < l ? l : (x0old > h ? h : x0old);
            double z = zold;
            if (zmode == 1) {
                // every (newly) pinned position sits on its bound, the free ones keep a feasible iterate for the safe
                // phase: the clipped subspace minimiser
                z = zp ? (((eqmask >> k) & 1ull) ? zold : (((upper >> k) & 1ull) ? h : l)) : clipped;
            } else if (zmode == 2) {
                z = zp ? zold : (k == zblock ? (zblock_upper ? h : l) : zold + zalpha * (x0old - zold));
            } else if (zmode == 3) {
                z = zp ? zold : clipped;
            }
            return z;
        };

        while (!converged) {
            // ================= forward sweep: lazy z update + pinned block elimination =================
            {
                FullBlocks<R> sa;
                sa.build(T[0]);
                SmallLDL<R> lprev;
                double hprev[R];
                // raw fields of knot k+1 (old x[0], old z, bounds) are in flight while knot k is eliminated
                double zp_ = 0.0, zc, zn = 0.0;
                double nx0 = 0.0, nz = 0.0, nl = 0.0, nh = 0.0, Tn;  // knot k+1
                {
                    double fx, fz;
                    ld_xz(1, fx, fz);
                    zc = znew(1, fx, fz, lo[3], hi[3]);
                }
                if (M > 2) { ld_xz(2, nx0, nz); nl = lo[6]; nh = hi[6]; }
                Tn = T[1];
                for (int k = 1; k < M; ++k) {
                    FullBlocks<R> sb;
                    sb.build(Tn);
                    if (k + 1 < M) zn = znew(k + 1, nx0, nz, nl, nh);
                    if (k + 2 < M) { ld_xz(k + 2, nx0, nz); nl = lo[3 * (k + 2)]; nh = hi[3 * (k + 2)]; }
                    if (k + 1 < M) Tn = T[k + 1];
                    st_z(k, zc);
                    const bool pk = (pin >> k) & 1ull;
                    const bool pprev = (k > 1) && ((pin >> (k - 1)) & 1ull);
                    const bool pnext = (k < M - 1) && ((pin >> (k + 1)) & 1ull);
                    double D[R][R], rhs[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        rhs[i] = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) D[i][c] = sa.B11[i][c] + sb.B00(i, c);
                    }
                    if (k == 1) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int c = 0; c < R; ++c) rhs[i] -= sa.B01[c][i] * x0[c];
                    } else if (pprev) {
#pragma unroll
                        for (int i = 0; i < R; ++i) rhs[i] -= sa.B01[0][i] * zp_;
                    }
                    if (k == M - 1) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int c = 0; c < R; ++c) rhs[i] -= sb.B01[i][c] * xM[c];
                    } else if (pnext) {
#pragma unroll
                        for (int i = 0; i < R; ++i) rhs[i] -= sb.B01[i][0] * zn;
                    }
                    if (pk) {
#pragma unroll
                        for (int i = 1; i < R; ++i) {
                            rhs[i] -= D[i][0] * zc;
                            D[i][0] = 0.0;
                            D[0][i] = 0.0;
                        }
                        D[0][0] = 1.0;
                        rhs[0] = zc;
                    }
                    if (k > 1) {
                        // masked coupling block between (k-1, k) and E = S_{k-1}^-1 Mp from the previous factors
                        double Mp[R][R], Ep[R][R];
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int c = 0; c < R; ++c) Mp[i][c] = ((pprev && i == 0) || (pk && c == 0)) ? 0.0 : sa.B01[i][c];
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            double col[R];
#pragma unroll
                            for (int i = 0; i < R; ++i) col[i] = Mp[i][c];
                            lprev.solve(col);
#pragma unroll
                            for (int i = 0; i < R; ++i) Ep[i][c] = col[i];
                        }
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int q = 0; q < R; ++q) {
#pragma unroll
                                for (int c = 0; c <= i; ++c) D[i][c] -= Mp[q][i] * Ep[q][c];
                                rhs[i] -= Mp[q][i] * hprev[q];
                            }
                    }
                    SmallLDL<R> ldl;
                    ldl.factor(D);
                    ldl.solve(rhs);
                    auto put = [&](auto&& at) {   // factors + h of knot k, straight from the registers they were computed in
                        int f = 0;
#pragma unroll
                        for (int i = 1; i < R; ++i)
#pragma unroll
                            for (int c = 0; c < i; ++c) at(F_L + (f++)) = ldl.l[i][c];
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            at(F_DI + i) = ldl.dinv[i];
                            at(F_X + i) = rhs[i];
                        }
                    };
                    if (NT > 0 && k >= lds_from) put([&](int q) -> double& { return S(k, q); });
                    else put([&](int q) -> double& { return G(k, q); });
#pragma unroll
                    for (int i = 0; i < R; ++i) hprev[i] = rhs[i];
                    lprev = ldl;
                    sa = sb;
                    zp_ = zc;
                    zc = zn;
                }
            }
            zmode = 0;

            // ================= backward sweep: x_k = h_k - S_k^-1 (M_k x_{k+1}) + the decisions of this iteration ======
            // The record of knot k-1 (factors, h, z) and its bounds are fetched while knot k is processed.
            const bool pdas = pdas_left > 0;
            unsigned long long npin = eqmask, nupper = 0ull;  // block-pivot round
            double alpha = 1.0, worst = 0.0;                    // ratio test / worst wrong-signed multiplier
            int block = -1, rel = -1;
            bool block_upper = false;
            {
                double xn[R], nx[F], nl, nh, Tn;  // record of the knot processed next
                double lamA = 0.0, magA = 0.0;  // part of knot (k+1)'s multiplier known before x_k is
#pragma unroll
                for (int f = 0; f < F; ++f) nx[f] = (NT > 0 && M - 1 >= lds_from) ? S(M - 1, f) : G(M - 1, f);
                nl = lo[3 * (M - 1)];
                nh = hi[3 * (M - 1)];
                Tn = T[M - 1];
#pragma unroll
                for (int i = 0; i < R; ++i) xn[i] = xM[i];
                for (int k = M - 1; k >= 0; --k) {
                    // segment k: inverse powers of its duration, coupling block B01 and the row-0 pieces
                    const double itv = fast_rcp(Tn);
                    double ip[2 * R];
                    ip[0] = 1.0;
#pragma unroll
                    for (int j = 1; j < 2 * R; ++j) ip[j] = ip[j - 1] * itv;
                    SegRow0<R> s0r;
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const double p = ip[2 * R - 1 - c];
                        s0r.e11[c] = p * Tab<R>::W(0, c);
                        s0r.e01r[c] = -p * Tab<R>::V(0, c);
                        s0r.e01c[c] = -p * Tab<R>::V(c, 0);
                    }
                    double x[R];
                    double zk = 0.0, lk = 0.0, hk = 0.0;
                    if (k >= 1) {
                        double cur[F];
#pragma unroll
                        for (int f = 0; f < F; ++f) cur[f] = nx[f];
                        lk = nl;
                        hk = nh;
                        zk = cur[F_Z];
                        if (k >= 2) {
                            if (NT > 0 && k - 1 >= lds_from) {
#pragma unroll
                                for (int f = 0; f < F; ++f) nx[f] = S(k - 1, f);
                            } else {
#pragma unroll
                                for (int f = 0; f < F; ++f) nx[f] = G(k - 1, f);
                            }
                            nl = lo[3 * (k - 1)];
                            nh = hi[3 * (k - 1)];
                        }
                        Tn = T[k - 1];
#pragma unroll
                        for (int i = 0; i < R; ++i) x[i] = cur[F_X + i];
                        if (k < M - 1) {
                            const bool pk = (pin >> k) & 1ull;
                            const bool pnext = (pin >> (k + 1)) & 1ull;
                            double t[R];
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                double acc = 0.0;
#pragma unroll
                                for (int c = 0; c < R; ++c) {
                                    const double m = -ip[2 * R - 1 - i - c] * Tab<R>::V(i, c);  // B01 of segment k
                                    acc += (((pk && i == 0) || (pnext && c == 0)) ? 0.0 : m) * xn[c];
                                }
                                t[i] = acc;
                            }
                            SmallLDL<R> ldl;
                            {
                                int f = 0;
#pragma unroll
                                for (int i = 1; i < R; ++i)
#pragma unroll
                                    for (int c = 0; c < i; ++c) ldl.l[i][c] = cur[F_L + (f++)];
                            }
#pragma unroll
                            for (int i = 0; i < R; ++i) ldl.dinv[i] = cur[F_DI + i];
                            ldl.solve(t);
#pragma unroll
                            for (int i = 0; i < R; ++i) x[i] -= t[i];
                            if (NT > 0 && k >= lds_from) {
#pragma unroll
                                for (int i = 0; i < R; ++i) S(k, F_X + i) = x[i];
                            } else {
#pragma unroll
                                for (int i = 0; i < R; ++i) G(k, F_X + i) = x[i];
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < R; ++i) x[i] = x0[i];
                    }
                    // ---- multiplier of knot j = k+1, now that x_k is known: d(cost)/d p_j (up to the factor 2), row 0 of
                    // the unmasked block row;  lower bound active: need lam >= 0, upper: lam <= 0
                    if (k + 1 <= M - 1) {
                        const int j = k + 1;
                        double lam = lamA, mag = magA;
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const double t1 = s0r.e01c[c] * x[c], t2 = s0r.e11[c] * xn[c];
                            lam += t1 + t2;
                            mag += fabs(t1) + fabs(t2);
                        }
                        const bool pj = (pin >> j) & 1ull, ej = (eqmask >> j) & 1ull, uj = (upper >> j) & 1ull;
                        const double viol = uj ? lam : -lam;
                        const bool wrong = viol > 1e-13 * mag;  // rounding of lam is a few ulp of mag; 1e-11 let a 4e-4-relative wrong-signed multiplier pass on T^-7-scaled blocks (tools/soak.py, seed 11)
                        if (pj && !ej) {
                            if (!wrong) {  // multiplier has the right sign: stays active in a block-pivot round
                                npin |= 1ull << j;
                                if (uj) nupper |= 1ull << j;
                            } else if (viol > worst || (viol == worst && rel >= 0)) {  // ties: the lowest knot, as an ascending scan
                                worst = viol;
                                rel = j;
                            }
                        }
                    }
                    if (k >= 1) {
                        // ---- first half of knot k's multiplier (needs x_k and x_{k+1} only)
                        lamA = 0.0;
                        magA = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const double t2 = ((c & 1) ? -s0r.e11[c] : s0r.e11[c]) * x[c], t3 = s0r.e01r[c] * xn[c];
                            lamA += t2 + t3;
                            magA += fabs(t2) + fabs(t3);
                        }
                        // ---- free position: outside its box?
                        if (!((pin >> k) & 1ull)) {
                            const double ph = x[0];
                            const bool below = ph < lk - 1e-12 * (1.0 + fabs(lk));
                            const bool above = !below && (ph > hk + 1e-12 * (1.0 + fabs(hk)));
                            if (below || above) {
                                npin |= 1ull << k;
                                if (above) nupper |= 1ull << k;
                                const double al = ((above ? hk : lk) - zk) / (ph - zk);
                                if (al < alpha || (al == alpha && block >= 0)) { alpha = al; block = k; block_upper = above; }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < R; ++i) xn[i] = x[i];
                    }
                }
            }
            if (final_pass) break;

            // ================= block-pivoting warm-up (primal-dual active set) =================
            // The first iterations change the whole working set at once: every free position outside its box is
            // pinned at the violated bound, every pinned one whose multiplier has the wrong sign is released.
            // It usually identifies the active set in 2-3 solves (vs one change per solve) but is not monotone, so
            // after PDAS_ITERS rounds the safe single-pivot method below takes over from the clipped (feasible) point.
            if (pdas) {
                ++it;
                if (npin == pin && nupper == upper) {  // KKT point: free positions feasible, all multipliers right
                    converged = true;
                    continue;
                }
                --pdas_left;
                zmode = 1;
                zpin = npin;
                pin = npin;
                upper = nupper;
                if (it >= a.max_iter) { pin = ~0ull; final_pass = true; }
                continue;
            }

            // ================= safe phase: primal active set, one change per solve =================
            if (block >= 0) {
                // partial step to the first blocking bound, which joins the working set
                zmode = 2;
                zpin = pin;
                zblock = block;
                zblock_upper = block_upper;
                zalpha = alpha < 0.0 ? 0.0 : alpha;
                pin |= 1ull << block;
                if (block_upper) upper |= 1ull << block; else upper &= ~(1ull << block);
            } else {
                // full step: free positions move to the subspace minimiser; release the worst wrong-signed multiplier
                if (rel < 0) {
                    converged = true;
                } else {
                    zmode = 3;
                    zpin = pin;
                    pin &= ~(1ull << rel);
                }
            }
            ++it;
            if (!converged && it >= a.max_iter) {
                pin = ~0ull;  // freeze the feasible iterate, re-solve the derivatives only
                final_pass = true;
            }
        }

        // ================= emission (record of knot k-1 in flight while segment k is written) =================
        double xe[R];
#pragma unroll
        for (int i = 0; i < R; ++i) xe[i] = xM[i];
        bool finite = true;
        double nxs[R + 1], Tn = T[M - 1];
        if (M >= 2) {
#pragma unroll
            for (int i = 0; i < R; ++i) nxs[i] = (NT > 0 && M - 1 >= lds_from) ? S(M - 1, F_X + i) : G(M - 1, F_X + i);
            nxs[R] = (NT > 0 && M - 1 >= lds_from) ? S(M - 1, F_Z) : G(M - 1, F_Z);
        }
        for (int k = M - 1; k >= 0; --k) {
            double xs[R];
            const double Tk = Tn;
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < R; ++i) xs[i] = x0[i];
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) xs[i] = nxs[i];
                if (!final_pass && ((pin >> k) & 1ull)) xs[0] = nxs[R];  // pinned positions: exact bound value
                if (k >= 2) {
                    if (NT > 0 && k - 1 >= lds_from) {
#pragma unroll
                        for (int i = 0; i < R; ++i) nxs[i] = S(k - 1, F_X + i);
                        nxs[R] = S(k - 1, F_Z);
                    } else {
#pragma unroll
                        for (int i = 0; i < R; ++i) nxs[i] = G(k - 1, F_X + i);
                        nxs[R] = G(k - 1, F_Z);
                    }
                }
                Tn = T[k - 1];
            }
            double ys[ND], ye[ND], c[NC];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                ys[d] = xs[d + 1];
                ye[d] = xe[d + 1];
            }
            segment_coeffs<R>(xs[0], ys, xe[0], ye, Tk, fast_rcp(Tk), c);
            double* o = out + (size_t)k * NC;
#pragma unroll
            for (int j = 0; j < NC; ++j) o[j] = c[j];
            finite = finite && (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
#pragma unroll
            for (int i = 0; i < R; ++i) xe[i] = xs[i];
        }
        if (!finite) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
        else if (final_pass) atomicMin(&a.status[b], (int32_t)UAVQP_MAX_ITER_REACHED);
        if (a.iters) atomicMax(&a.iters[b], (int32_t)it);
        if (a.active) {
            const unsigned long long valid = M >= 2 ? ((1ull << M) - 2ull) : 0ull;  // bits 1..M-1
            const unsigned long long fin = final_pass ? 0ull : (pin & ~eqmask & valid);
            a.active[2 * g] = fin;
            a.active[2 * g + 1] = upper & fin;
        }
    }
}

}  // namespace uavqp
