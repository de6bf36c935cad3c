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
// Sweep state lives in a lane-interleaved HBM workspace (any segment count, ragged batches).
#pragma once
#include "qp_device.h"

namespace uavqp {

struct CorridorArgs {
    int n_traj, uniform, max_segments, max_iter;
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

// r = 3 is held to 2 waves per SIMD (256 registers, ~100 B/lane scratch): the sweeps are chains of dependent
// HBM round trips, a second wave hides them (measured on config 3: 4.30 ms at 1 wave, 3.25 ms at 2, 4.19 ms at 3).
template <int R>
__global__ __launch_bounds__(64, R == 3 ? 2 : 1) void solve_corridor_kernel(CorridorArgs a) {
    constexpr int ND = R - 1, NC = 2 * R;
    // sweep state per interior knot: LDL' factors of S_k (strict lower triangle + inverse pivots), x_k (first h_k,
    // overwritten by the solution in the backward sweep) and the current position iterate z_k.  E_k = S_k^-1 M_k is
    // NOT stored: it is re-derived from the factors where needed (the kernel is bound by this HBM traffic).
    constexpr int NL = R * (R - 1) / 2;
    constexpr int F_L = 0, F_DI = NL, F_X = NL + R, F_Z = NL + 2 * R, F = NL + 2 * R + 1, F_H = F_X;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_slots = gridDim.x * blockDim.x;
    double* __restrict__ ws = a.ws + slot;
    const size_t wst = (size_t)n_slots;
    auto W = [&](int k, int f) -> double& { return ws[((size_t)(k - 1) * F + f) * wst]; };  // interior knot k = 1..M-1

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
        unsigned long long eqmask = 0ull, pin = 0ull, upper = 0ull;
        for (int k = 1; k < M; ++k) {
            const double l = lo[3 * k], h = hi[3 * k];
            double z = wp[3 * k];
            z = z < l ? l : (z > h ? h : z);
            W(k, F_Z) = z;
            if (l == h) eqmask |= 1ull << k;
        }
        pin = eqmask;

        int it = 0;
        int pdas_left = 3;  // PDAS_ITERS (measured on config 3: 3 rounds 13.8 mean iterations, 0 rounds 17.5, 10 rounds 15.7)
        bool converged = (M == 1);
        bool final_pass = false;  // max_iter hit: one last solve with every position pinned at the feasible iterate
        while (!converged) {
            // ================= pinned block-Thomas solve =================
            {
                FullBlocks<R> sa;
                sa.build(T[0]);
                SmallLDL<R> lprev;
                double hprev[R];
                for (int k = 1; k < M; ++k) {
                    FullBlocks<R> sb;
                    sb.build(T[k]);
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
                        const double zp = W(k - 1, F_Z);
#pragma unroll
                        for (int i = 0; i < R; ++i) rhs[i] -= sa.B01[0][i] * zp;
                    }
                    if (k == M - 1) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int c = 0; c < R; ++c) rhs[i] -= sb.B01[i][c] * xM[c];
                    } else if (pnext) {
                        const double zn = W(k + 1, F_Z);
#pragma unroll
                        for (int i = 0; i < R; ++i) rhs[i] -= sb.B01[i][0] * zn;
                    }
                    if (pk) {
                        const double zk = W(k, F_Z);
#pragma unroll
                        for (int i = 1; i < R; ++i) {
                            rhs[i] -= D[i][0] * zk;
                            D[i][0] = 0.0;
                            D[0][i] = 0.0;
                        }
                        D[0][0] = 1.0;
                        rhs[0] = zk;
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
                    {
                        int f = 0;
#pragma unroll
                        for (int i = 1; i < R; ++i)
#pragma unroll
                            for (int c = 0; c < i; ++c) W(k, F_L + (f++)) = ldl.l[i][c];
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        W(k, F_DI + i) = ldl.dinv[i];
                        hprev[i] = rhs[i];
                        W(k, F_X + i) = rhs[i];
                    }
                    lprev = ldl;
                    sa = sb;
                }
                // backward sweep: x_k = h_k - S_k^-1 (M_k x_{k+1}); the state of knot k-1 is fetched while knot k is
                // processed (one lane per (trajectory, axis) has nothing else to hide an HBM round trip per knot behind)
                constexpr int NB = NL + 2 * R;  // factors + h
                double xn[R], nx[NB];
#pragma unroll
                for (int f = 0; f < NB; ++f) nx[f] = W(M - 1, f);
                for (int k = M - 1; k >= 1; --k) {
                    double cur[NB];
#pragma unroll
                    for (int f = 0; f < NB; ++f) cur[f] = nx[f];
                    if (k >= 2) {
#pragma unroll
                        for (int f = 0; f < NB; ++f) nx[f] = W(k - 1, f);
                    }
                    double x[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) x[i] = cur[F_X + i];
                    if (k < M - 1) {
                        FullBlocks<R> sb;
                        sb.build(T[k]);
                        const bool pk = (pin >> k) & 1ull;
                        const bool pnext = (pin >> (k + 1)) & 1ull;
                        double t[R];
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            double acc = 0.0;
#pragma unroll
                            for (int c = 0; c < R; ++c) acc += (((pk && i == 0) || (pnext && c == 0)) ? 0.0 : sb.B01[i][c]) * xn[c];
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
#pragma unroll
                        for (int i = 0; i < R; ++i) W(k, F_X + i) = x[i];
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i) xn[i] = x[i];
                }
            }
            if (final_pass) break;

            // ================= block-pivoting warm-up (primal-dual active set) =================
            // The first iterations change the whole working set at once: every free position outside its box is
            // pinned at the violated bound, every pinned one whose multiplier has the wrong sign is released.
            // It usually identifies the active set in 2-3 solves (vs one change per solve) but is not monotone, so
            // after PDAS_ITERS rounds the safe single-pivot method below takes over from the clipped (feasible) point.
            if (pdas_left > 0) {
                unsigned long long npin = eqmask, nupper = 0ull;
                {
                    FullBlocks<R> sa;
                    sa.build(T[0]);
                    for (int k = 1; k < M; ++k) {
                        FullBlocks<R> sb;
                        sb.build(T[k]);
                        const double l = lo[3 * k], h = hi[3 * k];
                        if ((eqmask >> k) & 1ull) {
                        } else if ((pin >> k) & 1ull) {
                            double lam = 0.0, mag = 0.0;
#pragma unroll
                            for (int c = 0; c < R; ++c) {
                                const double xp = (k == 1) ? x0[c] : W(k - 1, F_X + c);
                                const double xk = W(k, F_X + c);
                                const double xq = (k == M - 1) ? xM[c] : W(k + 1, F_X + c);
                                const double t1 = sa.B01[c][0] * xp, t2 = (sa.B11[0][c] + sb.B00(0, c)) * xk, t3 = sb.B01[0][c] * xq;
                                lam += t1 + t2 + t3;
                                mag += fabs(t1) + fabs(t2) + fabs(t3);
                            }
                            const bool up = (upper >> k) & 1ull;
                            const double viol = up ? lam : -lam;
                            if (!(viol > 1e-11 * mag)) {  // multiplier has the right sign: stays active
                                npin |= 1ull << k;
                                if (up) nupper |= 1ull << k;
                            }
                        } else {
                            const double ph = W(k, F_X);
                            if (ph < l - 1e-12 * (1.0 + fabs(l))) npin |= 1ull << k;
                            else if (ph > h + 1e-12 * (1.0 + fabs(h))) { npin |= 1ull << k; nupper |= 1ull << k; }
                        }
                        sa = sb;
                    }
                }
                ++it;
                if (npin == pin && nupper == upper) {
                    // KKT point: free positions feasible, all multipliers right
                    for (int k = 1; k < M; ++k)
                        if (!((pin >> k) & 1ull)) W(k, F_Z) = W(k, F_X);
                    converged = true;
                    continue;
                }
                --pdas_left;
                for (int k = 1; k < M; ++k) {
                    const double l = lo[3 * k], h = hi[3 * k];
                    if ((npin >> k) & 1ull) {
                        if (!((eqmask >> k) & 1ull)) W(k, F_Z) = ((nupper >> k) & 1ull) ? h : l;
                    } else {
                        // keep a feasible iterate for the safe phase: clip the subspace minimiser
                        double z = W(k, F_X);
                        W(k, F_Z) = z < l ? l : (z > h ? h : z);
                    }
                }
                pin = npin;
                upper = nupper;
                if (it >= a.max_iter) { pin = ~0ull; final_pass = true; }
                continue;
            }

            // ================= ratio test on the free positions =================
            double alpha = 1.0;
            int block = -1;
            bool block_upper = false;
            for (int k = 1; k < M; ++k) {
                if ((pin >> k) & 1ull) continue;
                const double ph = W(k, F_X), zc = W(k, F_Z), l = lo[3 * k], h = hi[3 * k];
                if (ph < l - 1e-12 * (1.0 + fabs(l))) {
                    const double al = (l - zc) / (ph - zc);
                    if (al < alpha) { alpha = al; block = k; block_upper = false; }
                } else if (ph > h + 1e-12 * (1.0 + fabs(h))) {
                    const double al = (h - zc) / (ph - zc);
                    if (al < alpha) { alpha = al; block = k; block_upper = true; }
                }
            }
            if (block >= 0) {
                if (alpha < 0.0) alpha = 0.0;
                for (int k = 1; k < M; ++k) {
                    if ((pin >> k) & 1ull) continue;
                    const double zc = W(k, F_Z);
                    W(k, F_Z) = zc + alpha * (W(k, F_X) - zc);
                }
                W(block, F_Z) = block_upper ? hi[3 * block] : lo[3 * block];
                pin |= 1ull << block;
                if (block_upper) upper |= 1ull << block; else upper &= ~(1ull << block);
            } else {
                // full step: free positions move to the subspace minimiser; check the multipliers of the active bounds
                for (int k = 1; k < M; ++k)
                    if (!((pin >> k) & 1ull)) {
                        const double l = lo[3 * k], h = hi[3 * k];
                        double z = W(k, F_X);
                        W(k, F_Z) = z < l ? l : (z > h ? h : z);
                    }
                double worst = 0.0;
                int rel = -1;
                if (pin != eqmask) {
                    FullBlocks<R> sa;
                    sa.build(T[0]);
                    for (int k = 1; k < M; ++k) {
                        FullBlocks<R> sb;
                        sb.build(T[k]);
                        if (((pin & ~eqmask) >> k) & 1ull) {
                            // d(cost)/d p_k (up to the factor 2): row 0 of the unmasked block row
                            double lam = 0.0, mag = 0.0;
#pragma unroll
                            for (int c = 0; c < R; ++c) {
                                const double xp = (k == 1) ? x0[c] : W(k - 1, F_X + c);
                                const double xk = W(k, F_X + c);
                                const double xq = (k == M - 1) ? xM[c] : W(k + 1, F_X + c);
                                const double t1 = sa.B01[c][0] * xp, t2 = (sa.B11[0][c] + sb.B00(0, c)) * xk, t3 = sb.B01[0][c] * xq;
                                lam += t1 + t2 + t3;
                                mag += fabs(t1) + fabs(t2) + fabs(t3);
                            }
                            const double viol = ((upper >> k) & 1ull) ? lam : -lam;  // lower: need lam >= 0, upper: lam <= 0
                            if (viol > 1e-11 * mag && viol > worst) { worst = viol; rel = k; }
                        }
                        sa = sb;
                    }
                }
                if (rel < 0) converged = true;
                else pin &= ~(1ull << rel);
            }
            ++it;
            if (!converged && it >= a.max_iter) {
                pin = ~0ull;  // freeze the feasible iterate, re-solve the derivatives only
                final_pass = true;
            }
        }

        // ================= emission =================
        double xe[R];
#pragma unroll
        for (int i = 0; i < R; ++i) xe[i] = xM[i];
        bool finite = true;
        for (int k = M - 1; k >= 0; --k) {
            double xs[R];
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < R; ++i) xs[i] = x0[i];
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) xs[i] = W(k, F_X + i);
                if (!final_pass && ((pin >> k) & 1ull)) xs[0] = W(k, F_Z);  // pinned positions: exact bound value
            }
            double ys[ND], ye[ND], c[NC];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                ys[d] = xs[d + 1];
                ye[d] = xe[d + 1];
            }
            const double Tk = T[k];
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
    }
}

}  // namespace uavqp
