// qp_twisted.h -- register-resident specialised kernel for uniform batches (compile-time R, M).
//
// Two lanes per trajectory ("twisted" / two-sided block elimination): lane L works in the original
// time direction on segments 0..mL-1, lane R on the time-reversed trajectory (segments M-1..mL), both
// eliminating interior knots towards the meeting knot c = mL with the SAME instruction stream -- the
// min-control problem is time-reversal symmetric (derivative d picks up (-1)^d).  At the meeting knot
// the two partial Schur complements are exchanged through DPP (lane ^ 1), both lanes solve it, then
// back-substitute their own half and emit the monomial coefficients of their own segments.
//
// All per-knot state (E_k, h_k) lives in VGPRs; HBM traffic is exactly the algorithmic bytes:
// inputs are read once with coalesced loads into LDS, coefficients leave through an LDS transpose
// as full 16-B-per-lane stores (each 2R-coefficient chunk is written by adjacent lanes).
#pragma once
#include "qp_device.h"

namespace uavqp {

template <int R, int M>
struct TwistedCfg {
    static constexpr int ND = R - 1, NC = 2 * R, NK = M + 1;
    static constexpr int mL = (M + 1) / 2, mR = M / 2;
    static constexpr int TILE = 32;                 // trajectories per wave (2 lanes each)
    static constexpr int WP_D = TILE * NK * 3;      // doubles
    static constexpr int T_D = TILE * M;
    static constexpr int BC_D = TILE * 2 * ND * 3;
    static constexpr int OUT_STRIDE = 10;           // doubles per staged chunk (80 B: conflict-free b128 writes)
    static constexpr int PQ = NC / 2;               // 16-byte pieces per chunk
};

template <int R, int M>
__global__ __launch_bounds__(64, 2) void solve_twisted_kernel(BatchArgs a) {
    using C = TwistedCfg<R, M>;
    constexpr int ND = C::ND, NC = C::NC, NK = C::NK, mL = C::mL, mR = C::mR, TILE = C::TILE;
    static_assert(M >= 2, "twisted kernel needs an interior knot");

    __shared__ __attribute__((aligned(16))) double s_wp[C::WP_D];
    __shared__ __attribute__((aligned(16))) double s_T[C::T_D];
    __shared__ __attribute__((aligned(16))) double s_bc[C::BC_D];
    __shared__ __attribute__((aligned(16))) double s_out[2][64 * C::OUT_STRIDE];
    __shared__ int s_ok[TILE];

    const int lane = threadIdx.x;
    const int isR = lane & 1;
    const int tl = lane >> 1;
    const int m = isR ? mR : mL;
    const int n_tiles = (a.n_traj + TILE - 1) / TILE;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = tile * TILE;
        const int nv = min(TILE, a.n_traj - base);

        // ---------------- coalesced tile load: HBM -> LDS ----------------
        {
            const double* __restrict__ g = a.waypoints + (size_t)base * NK * 3;
            const int n = nv * NK * 3;
#pragma unroll
            for (int i = lane; i < C::WP_D; i += 64) s_wp[i] = (i < n) ? g[i] : 0.0;
        }
        {
            const double* __restrict__ g = a.times + (size_t)base * M;
            const int n = nv * M;
#pragma unroll
            for (int i = lane; i < C::T_D; i += 64) s_T[i] = (i < n) ? g[i] : 1.0;
        }
        {
            const double* __restrict__ g = a.bc + (size_t)base * 2 * ND * 3;
            const int n = nv * 2 * ND * 3;
#pragma unroll
            for (int i = lane; i < C::BC_D; i += 64) s_bc[i] = (i < n) ? g[i] : 0.0;
        }
        __syncthreads();

        // ---------------- validate own half, sanitise so that the arithmetic stays finite ----------------
        bool ok = (tl < nv);
#pragma unroll
        for (int j = 0; j < mL; ++j)
            if (j < m) {
                const double t = s_T[tl * M + (isR ? M - 1 - j : j)];
                ok = ok && (t > 0.0) && (t < INFINITY);
            }
        {
            const int oki = ok ? 1 : 0;
            const int other = __builtin_amdgcn_mov_dpp(oki, 0xB1, 0xF, 0xF, true);
            ok = (oki & other) != 0;
        }
        if (!isR) s_ok[tl] = ok ? 1 : 0;

        auto Tof = [&](int j) -> double {
            const double t = s_T[tl * M + (isR ? M - 1 - j : j)];
            return ok ? t : 1.0;
        };
        auto pos = [&](int j, int ax) -> double { return s_wp[(tl * NK + (isR ? M - j : j)) * 3 + ax]; };

        // own-frame boundary derivatives: y'_0 = F y_M for the reversed lane, F = diag((-1)^d)
        double y0[ND][3];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double v = s_bc[((tl * 2 + isR) * ND + d) * 3 + ax];
                y0[d][ax] = (isR && ((d & 1) == 0)) ? -v : v;
            }

        // ---------------- elimination of own interior knots j = 1..m-1 ----------------
        double E[mL][ND][ND], h[mL][ND][3];  // index 0 = boundary knot: E_0 = 0, h_0 = y0
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int c = 0; c < ND; ++c) E[0][i][c] = 0.0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) h[0][i][ax] = y0[i][ax];
        }
        SegBlocks<R> sa;
        sa.build(Tof(0));
        double pb[3], dpa[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            pb[ax] = pos(1, ax);
            dpa[ax] = pb[ax] - pos(0, ax);
        }
#pragma unroll
        for (int j = 1; j < mL; ++j) {
            if (j < m) {
                SegBlocks<R> sb;
                sb.build(Tof(j));
                double dpb[3];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const double pc = pos(j + 1, ax);
                    dpb[ax] = pc - pb[ax];
                    pb[ax] = pc;
                }
                double S[ND][ND], z[ND][3];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
#pragma unroll
                    for (int c = 0; c < ND; ++c) S[i][c] = sa.A11[i][c] + sb.A00[i][c];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) z[i][ax] = sb.gv[i] * dpb[ax] - sa.gw[i] * dpa[ax];
                }
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int q = 0; q < ND; ++q) {
                        if (j > 1) {
#pragma unroll
                            for (int c = 0; c <= i; ++c) S[i][c] -= sa.A01[q][i] * E[j - 1][q][c];
                        }
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) z[i][ax] -= sa.A01[q][i] * h[j - 1][q][ax];
                    }
                SmallLDL<ND> ldl;
                ldl.factor(S);
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    double col[ND];
#pragma unroll
                    for (int i = 0; i < ND; ++i) col[i] = z[i][ax];
                    ldl.solve(col);
#pragma unroll
                    for (int i = 0; i < ND; ++i) h[j][i][ax] = col[i];
                }
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double col[ND];
#pragma unroll
                    for (int i = 0; i < ND; ++i) col[i] = sb.A01[i][c];
                    ldl.solve(col);
#pragma unroll
                    for (int i = 0; i < ND; ++i) E[j][i][c] = col[i];
                }
                sa = sb;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) dpa[ax] = dpb[ax];
            }
        }

        // ---------------- meeting knot: own partial Schur complement, exchange, solve ----------------
        // sa = blocks of the last own segment (m-1); E/h index m-1 is the last eliminated knot (or the boundary).
        double P[ND][ND], zp[ND][3];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int c = 0; c < ND; ++c) P[i][c] = sa.A11[i][c];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) zp[i][ax] = -sa.gw[i] * dpa[ax];
        }
        {
            // E/h of knot m-1, selected per lane when the halves differ in length (odd M)
            double El[ND][ND], hl[ND][3];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c < ND; ++c)
                    El[i][c] = (mL == mR || !isR) ? E[mL - 1][i][c] : E[(mR > 0 ? mR : 1) - 1][i][c];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
                    hl[i][ax] = (mL == mR || !isR) ? h[mL - 1][i][ax] : h[(mR > 0 ? mR : 1) - 1][i][ax];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int q = 0; q < ND; ++q) {
#pragma unroll
                    for (int c = 0; c <= i; ++c) P[i][c] -= sa.A01[q][i] * El[q][c];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) zp[i][ax] -= sa.A01[q][i] * hl[q][ax];
                }
        }
        double ym[ND][3];  // solution at the meeting knot, own frame
        {
            double S[ND][ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c <= i; ++c) {
                    const double o = swap_pair(P[i][c]);
                    S[i][c] = P[i][c] + (((i + c) & 1) ? -o : o);
                }
#pragma unroll
                for (int c = i + 1; c < ND; ++c) S[i][c] = 0.0;  // upper triangle is never read
            }
            SmallLDL<ND> ldl;
            ldl.factor(S);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double col[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const double o = swap_pair(zp[i][ax]);
                    col[i] = zp[i][ax] + ((i & 1) ? o : -o);  // F_ii = (-1)^(i+1) for derivative d = i+1
                }
                ldl.solve(col);
#pragma unroll
                for (int i = 0; i < ND; ++i) ym[i][ax] = col[i];
            }
        }

        // ---------------- back-substitution + emission of own segments j = m-1 .. 0 ----------------
        double ynext[ND][3];
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) ynext[i][ax] = ym[i][ax];
        bool finite = true;
        int buf = 0;
        double* __restrict__ out = a.coeff + (size_t)base * 3 * M * NC;

#pragma unroll
        for (int jj = mL - 1; jj >= 0; --jj) {
            // lanes whose half is shorter (R, odd M) run one index behind so that knot indices stay aligned
            const int j = (mL == mR || !isR) ? jj : jj - 1;
            const bool act = (j >= 0);
            double y[ND][3];
            if (act) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        // h/E of knot j: compile-time index jj for L (and for R when halves are equal), jj-1 otherwise
                        double hv = (mL == mR || !isR) ? h[jj][i][ax] : h[jj > 0 ? jj - 1 : 0][i][ax];
                        y[i][ax] = hv;
                    }
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = 0; c < ND; ++c) {
                        const double e = (mL == mR || !isR) ? E[jj][i][c] : E[jj > 0 ? jj - 1 : 0][i][c];
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) y[i][ax] -= e * ynext[c][ax];
                    }
            }
            const double Tj = act ? Tof(j) : 1.0;
            const double itj = fast_rcp(Tj);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double c8[NC];
                if (act) {
                    const double pj = pos(j, ax), pj1 = pos(j + 1, ax);
                    double ys[ND], ye[ND];
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        // original orientation: L: start = knot j, end = knot j+1;  R: start = F knot j+1, end = F knot j
                        const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                        ys[d] = isR ? fs * ynext[d][ax] : y[d][ax];
                        ye[d] = isR ? fs * y[d][ax] : ynext[d][ax];
                    }
                    segment_coeffs<R>(isR ? pj1 : pj, ys, isR ? pj : pj1, ye, Tj, itj, c8);
                    finite = finite && (fabs(c8[NC - 1]) < INFINITY) && (fabs(c8[R]) < INFINITY);
                } else {
#pragma unroll
                    for (int k = 0; k < NC; ++k) c8[k] = 0.0;
                }
                // stage: lane-contiguous chunk -> LDS -> 16-B pieces on adjacent lanes -> HBM
                double* so = &s_out[buf][lane * C::OUT_STRIDE];
#pragma unroll
                for (int k = 0; k < NC; k += 2) *reinterpret_cast<double2*>(so + k) = make_double2(c8[k], c8[k + 1]);
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int cid = it * 16 + (lane >> 2), q = lane & 3;
                    const int ctl = cid >> 1, cR = cid & 1;
                    const int cj = (mL == mR || !cR) ? jj : jj - 1;  // own-frame segment of the producing lane
                    const int seg = cR ? (M - 1 - cj) : cj;
                    if (q < C::PQ && cj >= 0 && s_ok[ctl]) {
                        const double2 v = *reinterpret_cast<const double2*>(&s_out[buf][cid * C::OUT_STRIDE + 2 * q]);
                        double* dst = out + (((size_t)ctl * 3 + ax) * M + seg) * NC + 2 * q;
                        *reinterpret_cast<double2*>(dst) = v;
                    }
                }
                buf ^= 1;
            }
            if (act) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) ynext[i][ax] = y[i][ax];
            }
        }
        {
            const int f = finite ? 1 : 0;
            const int other = __builtin_amdgcn_mov_dpp(f, 0xB1, 0xF, 0xF, true);
            if (!isR && tl < nv && a.status) a.status[base + tl] = ok ? ((f & other) ? UAVQP_SOLVED : UAVQP_NON_FINITE) : UAVQP_INVALID_INPUT;
        }
        __syncthreads();  // LDS tile is reused by the next iteration
    }
}

}  // namespace uavqp
