// qp_phased.h -- phase-split workgroup kernel for SMALL uniform batches (latency shape, compile-time R, M).
//
// At the benchmark batch (4096 trajectories) the solve is latency-bound: one wave per SIMD executes a
// trajectory tile's whole dependent instruction stream, and an I/O-only kernel of the same grid already
// takes 3.4 us (tools/ubench/io_floor.hip).  This kernel shortens the per-tile critical path by giving
// every phase the lane mapping that exposes its own parallelism, handing data over through LDS:
//
//   P1  (trajectory, segment)        lanes  T-dependent blocks A11 / A01 / gw of each segment          -> LDS
//   P2  (trajectory, half)           lanes  two-sided matrix elimination ONCE per trajectory
//                                           (S_j^-1 by cofactors, E_j) instead of once per axis         -> LDS
//   P3  (trajectory, axis, half)     lanes  right-hand sides: forward sweep, meeting knot, back-substitution -> LDS
//   P4  (trajectory, axis, segment)  lanes  monomial coefficients, one 2r-chunk per lane, stored directly
//
// STATUS: correct (same parity tests as the other shapes, variant 64) but NOT faster -- measured 8.1 us per
// 4096-trajectory launch against 5.9 us for the register-resident 8-lane shape.  s_memtime stamps of one
// workgroup: load 1.9 k | blocks 0.7 k | chain 3.2 k | right-hand sides 2.4 k | emission 3.1 k cycles: every
// phase pays LDS round trips and runs with one wave per SIMD, which costs more than the removed redundancy
// saves.  Kept selectable (uavqp_set_variant(ctx, 64)) as the starting point for a version with 128-bit LDS
// traffic and two tiles in flight per workgroup; never chosen automatically.
//
// 256 threads, 16 trajectories per workgroup (256 workgroups for 4096 trajectories: every CU, because one
// CU moves only ~10 B/clk).  Same arithmetic as qp_twisted.h (same blocks, same two-sided elimination with
// the time-reversed half and the DPP exchange at the meeting knot); halves must be equal (even M).
#pragma once
#include "qp_device.h"

namespace uavqp {

template <int R, int M>
struct PhasedCfg {
    static constexpr int ND = R - 1, NC = 2 * R, NK = M + 1, mH = M / 2;
    static constexpr int G = 16;                         // trajectories per workgroup
    static constexpr int NA11 = ND * (ND + 1) / 2, NA01 = ND * ND;
    static constexpr int BLK = NA11 + NA01 + ND;         // doubles per (trajectory, segment)
    static constexpr int CHN = NA11 + NA01;              // per eliminated knot: S^-1 (packed) + E
    // per-trajectory strides are kept ODD (in doubles): lanes of different trajectories then fall on different
    // LDS banks (an even stride of 144 doubles cost a 16-way conflict in P2: 3.4 k -> cycles for 3 knots)
    static constexpr int SB = (M * BLK) | 1;
    static constexpr int WP_D = G * NK * 3, T_D = G * M, BC_D = G * 2 * ND * 3;
    static_assert(M % 2 == 0 && M >= 2, "phased kernel: equal halves");
};

template <int R, int M>
__global__ __launch_bounds__(256) void solve_phased_kernel(BatchArgs a) {
    using C = PhasedCfg<R, M>;
    constexpr int ND = C::ND, NC = C::NC, NK = C::NK, mH = C::mH, G = C::G;
    constexpr int NA11 = C::NA11, NA01 = C::NA01, BLK = C::BLK, CHN = C::CHN;

    __shared__ __attribute__((aligned(16))) double s_wp[C::WP_D];
    __shared__ __attribute__((aligned(16))) double s_T[C::T_D];
    __shared__ __attribute__((aligned(16))) double s_bc[C::BC_D];
    __shared__ double s_blk[G * C::SB];
    __shared__ double s_chn[G * 2 * (mH > 1 ? mH - 1 : 1) * CHN];
    __shared__ double s_meet[G * NA11];                 // S_m^-1 at the meeting knot, L frame
    __shared__ double s_y[G * 3 * NK * ND];             // derivatives at every knot, original frame
    __shared__ int s_ok[G];
    __shared__ int s_fin[G];

    const int tid = threadIdx.x;
    auto pk = [](int i, int c) { return i >= c ? i * (i + 1) / 2 + c : c * (c + 1) / 2 + i; };
    const int n_tiles = (a.n_traj + G - 1) / G;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = tile * G;
        const int nv = min(G, a.n_traj - base);

        UAVQP_STAMP(0);
        // ---------------- P0: inputs -> LDS ----------------
        {
            const double* __restrict__ g = a.waypoints + (size_t)base * NK * 3;
            for (int i = tid; i < C::WP_D; i += 256) s_wp[i] = (i < nv * NK * 3) ? g[i] : 0.0;
            const double* __restrict__ gt = a.times + (size_t)base * M;
            for (int i = tid; i < C::T_D; i += 256) s_T[i] = (i < nv * M) ? gt[i] : 1.0;
            const double* __restrict__ gb = a.bc + (size_t)base * 2 * ND * 3;
            for (int i = tid; i < C::BC_D; i += 256) s_bc[i] = (i < nv * 2 * ND * 3) ? gb[i] : 0.0;
            if (tid < G) {
                s_ok[tid] = tid < nv ? 1 : 0;
                s_fin[tid] = 1;
            }
        }
        __syncthreads();

        UAVQP_STAMP(1);
        // ---------------- P1: blocks per (trajectory, segment) ----------------
        if (tid < G * M) {
            const int t = tid / M, seg = tid - t * M;
            double T = s_T[t * M + seg];
            const bool good = (T > 0.0) && (T < INFINITY);
            if (!good) {
                atomicAnd(&s_ok[t], 0);
                T = 1.0;
            }
            SegBlocks<R> sb;
            sb.build(T);
            double* o = &s_blk[t * C::SB + seg * BLK];
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int c = 0; c <= i; ++c) o[pk(i, c)] = sb.A11[i][c];
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int c = 0; c < ND; ++c) o[NA11 + i * ND + c] = sb.A01[i][c];
#pragma unroll
            for (int i = 0; i < ND; ++i) o[NA11 + NA01 + i] = sb.gw[i];
        }
        __syncthreads();

        UAVQP_STAMP(2);
        // ---------------- P2: matrix elimination per (trajectory, half) ----------------
        if (tid < 2 * G) {
            const int t = tid >> 1, isR = tid & 1;
            auto blk = [&](int j) -> const double* { return &s_blk[t * C::SB + (isR ? M - 1 - j : j) * BLK]; };
            double Eprev[ND][ND];
#pragma unroll
            for (int j = 1; j < mH; ++j) {
                const double* ba = blk(j - 1);
                const double* bb = blk(j);
                double S[ND][ND];
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c) S[i][c] = ba[pk(i, c)] + (((i + c) & 1) ? -bb[pk(i, c)] : bb[pk(i, c)]);
                if (j > 1) {
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int q = 0; q < ND; ++q)
#pragma unroll
                            for (int c = 0; c <= i; ++c) S[i][c] -= ba[NA11 + q * ND + i] * Eprev[q][c];
                }
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = i + 1; c < ND; ++c) S[i][c] = 0.0;
                SymInv<ND> inv;
                inv.factor(S);
                double* o = &s_chn[((t * 2 + isR) * (mH - 1) + (j - 1)) * CHN];
                double Si[ND][ND];
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double col[ND];
#pragma unroll
                    for (int i = 0; i < ND; ++i) col[i] = (i == c) ? 1.0 : 0.0;
                    inv.solve(col);
#pragma unroll
                    for (int i = 0; i < ND; ++i) Si[i][c] = col[i];
                }
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c) o[pk(i, c)] = Si[i][c];
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double col[ND];
#pragma unroll
                    for (int i = 0; i < ND; ++i) col[i] = bb[NA11 + i * ND + c];
                    inv.solve(col);
#pragma unroll
                    for (int i = 0; i < ND; ++i) {
                        Eprev[i][c] = col[i];
                        o[NA11 + i * ND + c] = col[i];
                    }
                }
            }
            // meeting knot: own partial Schur complement, exchange, inverse (L frame stored)
            const double* bl = blk(mH - 1);
            double P[ND][ND];
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int c = 0; c <= i; ++c) P[i][c] = bl[pk(i, c)];
            if (mH > 1) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int q = 0; q < ND; ++q)
#pragma unroll
                        for (int c = 0; c <= i; ++c) P[i][c] -= bl[NA11 + q * ND + i] * Eprev[q][c];
            }
            double S[ND][ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c <= i; ++c) {
                    const double o = swap_pair(P[i][c]);
                    S[i][c] = P[i][c] + (((i + c) & 1) ? -o : o);
                }
#pragma unroll
                for (int c = i + 1; c < ND; ++c) S[i][c] = 0.0;
            }
            SymInv<ND> inv;
            inv.factor(S);
            if (!isR) {
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double col[ND];
#pragma unroll
                    for (int i = 0; i < ND; ++i) col[i] = (i == c) ? 1.0 : 0.0;
                    inv.solve(col);
#pragma unroll
                    for (int i = c; i < ND; ++i) s_meet[t * NA11 + pk(i, c)] = col[i];
                }
            }
        }
        __syncthreads();

        UAVQP_STAMP(3);
        // ---------------- P3: right-hand sides per (trajectory, axis, half) ----------------
        if (tid < 6 * G) {
            const int t = tid / 6, rem = tid - t * 6, ax = rem >> 1, isR = rem & 1;
            auto blk = [&](int j) -> const double* { return &s_blk[t * C::SB + (isR ? M - 1 - j : j) * BLK]; };
            auto pos = [&](int j) -> double { return s_wp[(t * NK + (isR ? M - j : j)) * 3 + ax]; };
            double h[mH][ND];  // own frame; h[0] = boundary derivatives
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const double v = s_bc[((t * 2 + isR) * ND + i) * 3 + ax];
                h[0][i] = (isR && ((i & 1) == 0)) ? -v : v;
            }
            double pb = pos(1), dpa = pb - pos(0);
#pragma unroll
            for (int j = 1; j < mH; ++j) {
                const double* ba = blk(j - 1);
                const double* bb = blk(j);
                const double* ch = &s_chn[((t * 2 + isR) * (mH - 1) + (j - 1)) * CHN];
                const double pc = pos(j + 1);
                const double dpb = pc - pb;
                pb = pc;
                double z[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const double gvb = (i & 1) ? bb[NA11 + NA01 + i] : -bb[NA11 + NA01 + i];
                    double acc = gvb * dpb - ba[NA11 + NA01 + i] * dpa;
#pragma unroll
                    for (int q = 0; q < ND; ++q) acc -= ba[NA11 + q * ND + i] * h[j - 1][q];
                    z[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < ND; ++q) acc += ch[pk(i, q)] * z[q];
                    h[j][i] = acc;
                }
                dpa = dpb;
            }
            // meeting knot
            const double* bl = blk(mH - 1);
            double zp[ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                double acc = -bl[NA11 + NA01 + i] * dpa;
#pragma unroll
                for (int q = 0; q < ND; ++q) acc -= bl[NA11 + q * ND + i] * h[mH - 1][q];
                zp[i] = acc;
            }
            double ynext[ND];
            {
                double zm[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const double o = swap_pair(zp[i]);
                    zm[i] = zp[i] + ((i & 1) ? o : -o);  // z_own + F z_other
                }
                // S_m^-1 in the own frame: L as stored, R = F S^-1 F
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < ND; ++q) {
                        const double sv = s_meet[t * NA11 + pk(i, q)];
                        acc += ((isR && ((i + q) & 1)) ? -sv : sv) * zm[q];
                    }
                    ynext[i] = acc;
                }
            }
            double* yo = &s_y[(t * 3 + ax) * NK * ND];
            if (!isR) {
#pragma unroll
                for (int i = 0; i < ND; ++i) yo[mH * ND + i] = ynext[i];
            }
            // back-substitution, written in the original frame (R: knot M-j, derivative d flips by (-1)^d)
#pragma unroll
            for (int j = mH - 1; j >= 0; --j) {
                double y[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) y[i] = h[j][i];
                if (j > 0) {
                    const double* ch = &s_chn[((t * 2 + isR) * (mH - 1) + (j - 1)) * CHN];
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int q = 0; q < ND; ++q) y[i] -= ch[NA11 + i * ND + q] * ynext[q];
                }
                const int knot = isR ? M - j : j;
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    yo[knot * ND + i] = (isR && ((i & 1) == 0)) ? -y[i] : y[i];
                    ynext[i] = y[i];
                }
            }
        }
        __syncthreads();

        UAVQP_STAMP(4);
        // ---------------- P4: coefficients per (trajectory, axis, segment), one chunk per lane ----------------
        double* __restrict__ out = a.coeff + (size_t)base * 3 * M * NC;
        for (int u = tid; u < G * 3 * M; u += 256) {
            const int t = u / (3 * M), r2 = u - t * 3 * M, ax = r2 / M, seg = r2 - ax * M;
            const double* yo = &s_y[(t * 3 + ax) * NK * ND];
            double ys[ND], ye[ND], c8[NC];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                ys[d] = yo[seg * ND + d];
                ye[d] = yo[(seg + 1) * ND + d];
            }
            const double Ts = s_T[t * M + seg];
            const bool tok = s_ok[t] != 0;
            const double Tj = tok ? Ts : 1.0;
            segment_coeffs<R>(s_wp[(t * NK + seg) * 3 + ax], ys, s_wp[(t * NK + seg + 1) * 3 + ax], ye, Tj, fast_rcp(Tj), c8);
            if (!((fabs(c8[NC - 1]) < INFINITY) && (fabs(c8[R]) < INFINITY))) atomicAnd(&s_fin[t], 0);
            if (tok) {
                double* dst = out + (((size_t)t * 3 + ax) * M + seg) * NC;
#pragma unroll
                for (int k = 0; k < NC; k += 2) *reinterpret_cast<double2*>(dst + k) = make_double2(c8[k], c8[k + 1]);
            }
        }
        __syncthreads();
        UAVQP_STAMP(5);
        if (tid < nv && a.status) a.status[base + tid] = s_ok[tid] ? (s_fin[tid] ? UAVQP_SOLVED : UAVQP_NON_FINITE) : UAVQP_INVALID_INPUT;
        __syncthreads();  // LDS is reused by the next tile
    }
}

}  // namespace uavqp
