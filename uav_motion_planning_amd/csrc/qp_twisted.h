// qp_twisted.h -- register-resident specialised kernel for uniform batches (compile-time R, M).
//
// Two lanes per trajectory ("twisted" / two-sided block elimination): lane L works in the original
// time direction on segments 0..mL-1, lane R on the time-reversed trajectory (segments M-1..mL), both
// eliminating interior knots towards the meeting knot c = mL with the SAME instruction stream -- the
// min-control problem is time-reversal symmetric (derivative d picks up (-1)^d).  At the meeting knot
// the two partial Schur complements are exchanged through DPP (lane ^ 1), both lanes solve it, then
// back-substitute their own half and emit the monomial coefficients of their own segments.
//
// Memory plan (one wave = one workgroup = TILE trajectories per tile, persistent over tiles):
//   * inputs of tile n+1 are prefetched HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs)
//     while tile n is being eliminated: double-buffered 2 x 13.25 KiB (r=4, M=8);
//   * all per-knot state (E_k, h_k) lives in VGPRs -- nothing is spilled to HBM;
//   * coefficients of one segment (3 axes) are transposed through LDS so that every
//     global_store_dwordx4 writes whole 2r-coefficient chunks from adjacent lanes.
// HBM traffic is therefore exactly the algorithmic bytes.
#pragma once
#include "qp_device.h"

#ifdef UAVQP_PHASE_TIMING
#define UAVQP_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define UAVQP_STAMP(i) do {} while (0)
#endif

namespace uavqp {

template <int R, int M, int TILE, int LPT = 2>
struct TwistedCfg {
    static constexpr int ND = R - 1, NC = 2 * R, NK = M + 1;
    static constexpr int mL = (M + 1) / 2, mR = M / 2;
    static constexpr int WP_D = TILE * NK * 3;  // doubles per tile
    static constexpr int T_D = TILE * M;
    static constexpr int BC_D = TILE * 2 * ND * 3;
    static constexpr int T_OFF = WP_D, BC_OFF = WP_D + T_D;
    static constexpr int IN_D = WP_D + T_D + BC_D;
    // staged chunk row per producing lane: 3 axes x 2r doubles, padded so that the row stride in dwords
    // is 4 (mod 8): conflict-free ds_write_b128 within its 8-lane groups
    static constexpr int OUT_STRIDE = (3 * NC + 2) % 4 == 0 ? 3 * NC + 4 : 3 * NC + 2;
    static constexpr int PQ = NC / 2;  // 16-byte pieces per chunk
    // LPT = lanes per trajectory.  2: one (L, R) lane pair carries all three axes (throughput shape).
    // 8: one lane pair per axis (+ one idle pair) -- the matrix elimination is repeated by the three pairs,
    // right-hand sides, back-substitution and coefficient emission are split by axis (latency shape for
    // small batches: ~2x shorter critical path per wave, 8 trajectories per wave).
    static constexpr int NAX = LPT == 2 ? 3 : 1;
    // Pair mode: both halves have an even number of segments, so own segments are emitted two at a time and
    // the two 2r-coefficient chunks of a (trajectory, axis, segment pair) are stored by back-to-back
    // instructions -- for r = 4 that is one whole 128-B line, which L2 then writes out as a full line
    // (measured: 5.3 TB/s vs 3.5 TB/s when the halves of a line arrive a segment apart).
    static constexpr bool PAIRS = (mL == mR) && (mL % 2 == 0);
    static constexpr int PAIR_STRIDE = (2 * NC + 2) % 4 == 0 ? 2 * NC + 4 : 2 * NC + 2;  // one axis, two segments
    static constexpr int OUT_D = 64 * (PAIRS ? PAIR_STRIDE : OUT_STRIDE);
    static_assert(WP_D % 2 == 0 && T_D % 2 == 0 && BC_D % 2 == 0, "tile arrays must be whole 16-B pairs");
};

// Single-wave workgroups: LDS operations of one wave execute in issue order, so cross-lane hand-offs
// through LDS need no s_barrier and -- unlike __syncthreads() -- must NOT wait for outstanding global
// stores (vmcnt).  This only pins the compiler's ordering of LDS accesses.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// s_waitcnt vmcnt(0) through the builtin (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15) so that the
// compiler's own wait-count bookkeeping knows that pending LDS-DMA writes have landed.
__device__ __forceinline__ void wait_vmcnt0() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}

// 16-B coefficient store.  UAVQP_NT_STORES: non-temporal (streaming) stores -- the output is never
// re-read by this kernel, and lines that are not left dirty in L2 do not have to be written back at
// the kernel boundary.
__device__ __forceinline__ void store_pair(double* dst, double2 v) {
#ifdef UAVQP_NT_STORES
    typedef double nt_v2 __attribute__((ext_vector_type(2)));
    nt_v2 w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_v2*>(dst));  // global_store_dwordx4 ... nt
#else
    *reinterpret_cast<double2*>(dst) = v;
#endif
}

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// Asynchronous HBM -> LDS copy of ND_ doubles (full tile): one global_load_lds_dwordx4 per 1 KiB.
template <int ND_>
__device__ __forceinline__ void dma_tile(const double* __restrict__ g, double* s, int lane) {
    constexpr int NP = ND_ / 2, NL = (NP + 63) / 64;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int p = lane + 64 * k;
        if (p < NP) __builtin_amdgcn_global_load_lds((gas_ptr)(g + 2 * p), (las_ptr)(s + 128 * k), 16, 0, 0);
    }
}

// Guarded synchronous copy for the partial last tile; entries at or beyond n_valid become `fill`.
template <int ND_>
__device__ __forceinline__ void load_tile_guarded(const double* __restrict__ g, int n_valid, double* s, int lane, double fill) {
    for (int i = lane; i < ND_; i += 64) s[i] = (i < n_valid) ? g[i] : fill;
}

template <int R, int M, int TILE, int LPT = 2>
__global__ __launch_bounds__(64, 1) void solve_twisted_kernel(BatchArgs a) {
    using C = TwistedCfg<R, M, TILE, LPT>;
    constexpr int NAX = C::NAX;
    constexpr int ND = C::ND, NC = C::NC, NK = C::NK, mL = C::mL, mR = C::mR;
    static_assert(M >= 2, "twisted kernel needs an interior knot");
    static_assert((LPT == 2 && (TILE == 32 || TILE == 16)) || (LPT == 8 && TILE == 8), "tile shapes: 2 lanes x 32|16, 8 lanes x 8");

    __shared__ __attribute__((aligned(16))) double s_in[2][C::IN_D];
    __shared__ __attribute__((aligned(16))) double s_out[LPT == 2 ? C::OUT_D : 2];

    const int lane = threadIdx.x;
    const int isR = lane & 1;
    const int tl = lane / LPT;
    const int tlc = tl < TILE ? tl : TILE - 1;  // clamp LDS indexing of idle lanes (TILE == 16)
    const int axl = (lane % LPT) >> 1;           // LPT == 8: axis of this lane pair (3 = idle pair)
    const int ax0 = LPT == 2 ? 0 : (axl < 3 ? axl : 2);
    const int m = isR ? mR : mL;
    const int n_tiles = (a.n_traj + TILE - 1) / TILE;

    auto issue_tile = [&](int tile, int buf) {
        const int base = tile * TILE;
        const int nv = min(TILE, a.n_traj - base);
        double* s = s_in[buf];
        if (nv == TILE) {
            dma_tile<C::WP_D>(a.waypoints + (size_t)base * NK * 3, s, lane);
            dma_tile<C::T_D>(a.times + (size_t)base * M, s + C::T_OFF, lane);
            dma_tile<C::BC_D>(a.bc + (size_t)base * 2 * ND * 3, s + C::BC_OFF, lane);
        } else {
            load_tile_guarded<C::WP_D>(a.waypoints + (size_t)base * NK * 3, nv * NK * 3, s, lane, 0.0);
            load_tile_guarded<C::T_D>(a.times + (size_t)base * M, nv * M, s + C::T_OFF, lane, 1.0);
            load_tile_guarded<C::BC_D>(a.bc + (size_t)base * 2 * ND * 3, nv * 2 * ND * 3, s + C::BC_OFF, lane, 0.0);
        }
    };

    int buf = 0;
    if ((int)blockIdx.x < n_tiles) issue_tile(blockIdx.x, 0);
    wait_vmcnt0();
    wave_lds_sync();

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        const int base = tile * TILE;
        const int nv = min(TILE, a.n_traj - base);
        UAVQP_STAMP(0);
        // prefetch the next tile into the other buffer; it lands while this tile is eliminated
        if (tile + (int)gridDim.x < n_tiles) issue_tile(tile + gridDim.x, buf ^ 1);
        UAVQP_STAMP(1);

        const double* __restrict__ s_wp = s_in[buf];
        const double* __restrict__ s_T = s_in[buf] + C::T_OFF;
        const double* __restrict__ s_bc = s_in[buf] + C::BC_OFF;

        // ---------------- validate own half, sanitise so that the arithmetic stays finite ----------------
        bool ok = (tl < nv) && (LPT == 2 || axl < 3);
#pragma unroll
        for (int j = 0; j < mL; ++j)
            if (j < m) {
                const double t = s_T[tlc * M + (isR ? M - 1 - j : j)];
                ok = ok && (t > 0.0) && (t < INFINITY);
            }
        {
            const int oki = ok ? 1 : 0;
            const int other = __builtin_amdgcn_mov_dpp(oki, 0xB1, 0xF, 0xF, true);
            ok = (oki & other) != 0;
        }
        const unsigned long long okmask = __ballot(ok);  // bit LPT*t: trajectory t of this tile is valid

        auto Tof = [&](int j) -> double {
            const double t = s_T[tlc * M + (isR ? M - 1 - j : j)];
            return ok ? t : 1.0;
        };
        auto pos = [&](int j, int ax) -> double { return s_wp[(tlc * NK + (isR ? M - j : j)) * 3 + ax0 + ax]; };

        // ---------------- elimination of own interior knots j = 1..m-1 ----------------
        // index 0 = boundary knot: E_0 = 0, h_0 = y0 (own frame: y'_0 = F y_M for the reversed lane)
        double E[mL][ND][ND], h[mL][ND][NAX];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int c = 0; c < ND; ++c) E[0][i][c] = 0.0;
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                const double v = s_bc[((tlc * 2 + isR) * ND + i) * 3 + ax0 + ax];
                h[0][i][ax] = (isR && ((i & 1) == 0)) ? -v : v;
            }
        }
        SegBlocks<R> sa;
        sa.build(Tof(0));
        double pb[NAX], dpa[NAX];
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) {
            pb[ax] = pos(1, ax);
            dpa[ax] = pb[ax] - pos(0, ax);
        }
#pragma unroll
        for (int j = 1; j < mL; ++j) {
            if (j < m) {
                SegBlocks<R> sb;
                sb.build(Tof(j));
                double dpb[NAX];
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) {
                    const double pc = pos(j + 1, ax);
                    dpb[ax] = pc - pb[ax];
                    pb[ax] = pc;
                }
                double S[ND][ND], z[ND][NAX];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
#pragma unroll
                    for (int c = 0; c < ND; ++c) S[i][c] = sa.A11[i][c] + sb.A00(i, c);
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) z[i][ax] = sb.gv(i) * dpb[ax] - sa.gw[i] * dpa[ax];
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
                        for (int ax = 0; ax < NAX; ++ax) z[i][ax] -= sa.A01[q][i] * h[j - 1][q][ax];
                    }
                SmallLDL<ND> ldl;
                ldl.factor(S);
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) {
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
                for (int ax = 0; ax < NAX; ++ax) dpa[ax] = dpb[ax];
            }
        }
        UAVQP_STAMP(2);

        // ---------------- meeting knot: own partial Schur complement, exchange, solve ----------------
        // sa = blocks of the last own segment (m-1); E/h index m-1 is the last eliminated knot (or the boundary).
        double P[ND][ND], zp[ND][NAX];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int c = 0; c < ND; ++c) P[i][c] = sa.A11[i][c];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) zp[i][ax] = -sa.gw[i] * dpa[ax];
        }
        {
            // E/h of knot m-1, selected per lane when the halves differ in length (odd M)
            double El[ND][ND], hl[ND][NAX];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c < ND; ++c) El[i][c] = (mL == mR || !isR) ? E[mL - 1][i][c] : E[mR - 1][i][c];
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) hl[i][ax] = (mL == mR || !isR) ? h[mL - 1][i][ax] : h[mR - 1][i][ax];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int q = 0; q < ND; ++q) {
#pragma unroll
                    for (int c = 0; c <= i; ++c) P[i][c] -= sa.A01[q][i] * El[q][c];
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) zp[i][ax] -= sa.A01[q][i] * hl[q][ax];
                }
        }
        double ynext[ND][NAX];  // solution at the meeting knot, own frame
        {
            double S[ND][ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c <= i; ++c) {
                    const double o = swap_pair(P[i][c]);
                    S[i][c] = P[i][c] + (((i + c) & 1) ? -o : o);  // P_own + F P_other F
                }
#pragma unroll
                for (int c = i + 1; c < ND; ++c) S[i][c] = 0.0;  // upper triangle is never read
            }
            SmallLDL<ND> ldl;
            ldl.factor(S);
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                double col[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const double o = swap_pair(zp[i][ax]);
                    col[i] = zp[i][ax] + ((i & 1) ? o : -o);  // z_own + F z_other, F_ii = (-1)^(i+1)
                }
                ldl.solve(col);
#pragma unroll
                for (int i = 0; i < ND; ++i) ynext[i][ax] = col[i];
            }
        }
        UAVQP_STAMP(3);

        // The prefetched tile has had the whole elimination to land; this also retires the previous
        // tile's stores (issued before the prefetch) so that the waits below never see them.
        wait_vmcnt0();

        // ---------------- back-substitution + emission of own segments j = m-1 .. 0 ----------------
        bool finite = true;
        double* __restrict__ out = a.coeff + (size_t)base * 3 * M * NC;

        if constexpr (LPT == 8) {
            // one axis per lane pair: coefficients go straight from registers to HBM (4 x 16 B per segment);
            // at the batch sizes this shape is used for, store efficiency is irrelevant, latency is not.
#pragma unroll
            for (int jj = mL - 1; jj >= 0; --jj) {
                const int j = (mL == mR || !isR) ? jj : jj - 1;
                const bool act = (j >= 0);
                double y[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) y[i] = (mL == mR || !isR) ? h[jj][i][0] : h[jj > 0 ? jj - 1 : 0][i][0];
                if (jj > 0 || mL != mR) {
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int c = 0; c < ND; ++c) {
                            const double e = (mL == mR || !isR) ? E[jj][i][c] : E[jj > 0 ? jj - 1 : 0][i][c];
                            y[i] -= e * ynext[c][0];
                        }
                }
                const int jc = act ? j : 0;
                const double Tj = Tof(jc);
                const double itj = fast_rcp(Tj);
                const double pj = pos(jc, 0), pj1 = pos(jc + 1, 0);
                double ys[ND], ye[ND], c8[NC];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                    ys[d] = isR ? fs * ynext[d][0] : y[d];
                    ye[d] = isR ? fs * y[d] : ynext[d][0];
                }
                segment_coeffs<R>(isR ? pj1 : pj, ys, isR ? pj : pj1, ye, Tj, itj, c8);
                if (act) finite = finite && (fabs(c8[NC - 1]) < INFINITY) && (fabs(c8[R]) < INFINITY);
                const int seg = isR ? (M - 1 - jc) : jc;
                double* dst = (ok && act) ? out + (((size_t)tlc * 3 + ax0) * M + seg) * NC : a.dummy + 2 * lane;
                const int step = (ok && act) ? 2 : 0;  // the sink is one 16-B slot per lane
#pragma unroll
                for (int k = 0; k < NC; k += 2) store_pair(dst + (k / 2) * step, make_double2(c8[k], c8[k + 1]));
                if (act) {
#pragma unroll
                    for (int i = 0; i < ND; ++i) ynext[i][0] = y[i];
                }
            }
        } else if constexpr (C::PAIRS) {
#pragma unroll
            for (int pp = mL / 2 - 1; pp >= 0; --pp) {
                constexpr int NII = TILE / 8;
                const int j1 = 2 * pp + 1, j0 = 2 * pp;  // own segments of this pair (halves are equal: no lag)
                double y1[ND][NAX], y0[ND][NAX];
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) y1[i][ax] = h[j1][i][ax];
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = 0; c < ND; ++c)
#pragma unroll
                        for (int ax = 0; ax < NAX; ++ax) y1[i][ax] -= E[j1][i][c] * ynext[c][ax];
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) y0[i][ax] = h[j0][i][ax];
                if (j0 > 0) {
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int c = 0; c < ND; ++c)
#pragma unroll
                            for (int ax = 0; ax < NAX; ++ax) y0[i][ax] -= E[j0][i][c] * y1[c][ax];
                }
                const double T1 = Tof(j1), T0 = Tof(j0);
                const double it1 = fast_rcp(T1), it0 = fast_rcp(T0);
                // LDS row of this lane: [lower original segment | higher original segment]
                double* so = &s_out[lane * C::PAIR_STRIDE];
                double* so1 = so + (isR ? 0 : NC);  // own j1: the higher original segment for L, the lower for R
                double* so0 = so + (isR ? NC : 0);
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) {
                    const double p0 = pos(j0, ax), p1 = pos(j1, ax), p2 = pos(j1 + 1, ax);
                    double ys[ND], ye[ND], ca[NC], cb[NC];
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                        ys[d] = isR ? fs * ynext[d][ax] : y1[d][ax];
                        ye[d] = isR ? fs * y1[d][ax] : ynext[d][ax];
                    }
                    segment_coeffs<R>(isR ? p2 : p1, ys, isR ? p1 : p2, ye, T1, it1, ca);
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                        ys[d] = isR ? fs * y1[d][ax] : y0[d][ax];
                        ye[d] = isR ? fs * y0[d][ax] : y1[d][ax];
                    }
                    segment_coeffs<R>(isR ? p1 : p0, ys, isR ? p0 : p1, ye, T0, it0, cb);
                    finite = finite && (fabs(ca[NC - 1]) < INFINITY) && (fabs(ca[R]) < INFINITY) &&
                             (fabs(cb[NC - 1]) < INFINITY) && (fabs(cb[R]) < INFINITY);
#pragma unroll
                    for (int k = 0; k < NC; k += 2) {
                        *reinterpret_cast<double2*>(so1 + k) = make_double2(ca[k], ca[k + 1]);
                        *reinterpret_cast<double2*>(so0 + k) = make_double2(cb[k], cb[k + 1]);
                    }
                    wave_lds_sync();
                    double2 v[NII][2];
#pragma unroll
                    for (int ii = 0; ii < NII; ++ii)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int pl = ii * 16 + (lane >> 2), q = lane & 3;
                            v[ii][hh] = *reinterpret_cast<const double2*>(&s_out[pl * C::PAIR_STRIDE + hh * NC + 2 * (q < C::PQ ? q : 0)]);
                        }
#pragma unroll
                    for (int ii = 0; ii < NII; ++ii)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int pl = ii * 16 + (lane >> 2), q = lane & 3;
                            const int ctl = pl >> 1, cR = pl & 1;
                            const int seg = (cR ? (M - 1 - j1) : j0) + hh;
                            const bool keep = (q < C::PQ) && ((okmask >> (2 * ctl)) & 1ull);
                            double* dst = keep ? out + (((size_t)ctl * 3 + ax) * M + seg) * NC + 2 * q : a.dummy + 2 * lane;
                            store_pair(dst, v[ii][hh]);
                        }
                    wave_lds_sync();  // the row is rewritten by the next axis: keep the reads above it
                }
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) ynext[i][ax] = y0[i][ax];
            }
        } else {
#pragma unroll
        for (int jj = mL - 1; jj >= 0; --jj) {
            // the lane with the shorter half (R, odd M) runs one index behind
            const int j = (mL == mR || !isR) ? jj : jj - 1;
            const bool act = (j >= 0);
            double y[ND][NAX];
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) y[i][ax] = (mL == mR || !isR) ? h[jj][i][ax] : h[jj > 0 ? jj - 1 : 0][i][ax];
            if (jj > 0 || mL != mR) {  // E_0 = 0: nothing to subtract at the boundary knot
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = 0; c < ND; ++c) {
                        const double e = (mL == mR || !isR) ? E[jj][i][c] : E[jj > 0 ? jj - 1 : 0][i][c];
#pragma unroll
                        for (int ax = 0; ax < NAX; ++ax) y[i][ax] -= e * ynext[c][ax];
                    }
            }
            const int jc = act ? j : 0;
            const double Tj = Tof(jc);
            const double itj = fast_rcp(Tj);
            double* so = &s_out[lane * C::OUT_STRIDE];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                const double pj = pos(jc, ax), pj1 = pos(jc + 1, ax);
                double ys[ND], ye[ND], c8[NC];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    // original orientation: L: start = knot j, end = knot j+1;  R: start = F knot j+1, end = F knot j
                    const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                    ys[d] = isR ? fs * ynext[d][ax] : y[d][ax];
                    ye[d] = isR ? fs * y[d][ax] : ynext[d][ax];
                }
                segment_coeffs<R>(isR ? pj1 : pj, ys, isR ? pj : pj1, ye, Tj, itj, c8);
                if (act) finite = finite && (fabs(c8[NC - 1]) < INFINITY) && (fabs(c8[R]) < INFINITY);
#pragma unroll
                for (int k = 0; k < NC; k += 2) *reinterpret_cast<double2*>(so + ax * NC + k) = make_double2(c8[k], c8[k + 1]);
            }
            wave_lds_sync();
            // transpose out: store instruction (ax, ii): lane -> 16-B piece q of the chunk produced by lane pl
            constexpr int NII = TILE / 8;  // producing lanes / 16
            double2 v[3][NII];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax)
#pragma unroll
                for (int ii = 0; ii < NII; ++ii) {
                    const int pl = ii * 16 + (lane >> 2), q = lane & 3;
                    v[ax][ii] = *reinterpret_cast<const double2*>(&s_out[pl * C::OUT_STRIDE + ax * NC + 2 * (q < C::PQ ? q : 0)]);
                }
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax)
#pragma unroll
                for (int ii = 0; ii < NII; ++ii) {
                    const int pl = ii * 16 + (lane >> 2), q = lane & 3;
                    const int ctl = pl >> 1, cR = pl & 1;
                    const int cj = (mL == mR || !cR) ? jj : jj - 1;  // own-frame segment of the producing lane
                    const int seg = cR ? (M - 1 - cj) : cj;
                    // branch-free: pieces of invalid / padding trajectories go to a scratch line instead
                    const bool keep = (q < C::PQ) && (cj >= 0) && ((okmask >> (2 * ctl)) & 1ull);
                    double* dst = keep ? out + (((size_t)ctl * 3 + ax) * M + seg) * NC + 2 * q : a.dummy + 2 * lane;
                    store_pair(dst, v[ax][ii]);
                }
            if (act) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) ynext[i][ax] = y[i][ax];
            }
        }
        }
        {
            const int f = finite ? 1 : 0;
            const int other = __builtin_amdgcn_mov_dpp(f, 0xB1, 0xF, 0xF, true);
            bool fin = (f & other) != 0;
            if constexpr (LPT == 8) {  // all three axis pairs of the trajectory must be finite
                const unsigned long long fm = __ballot(fin || axl == 3);
                fin = ((fm >> (8 * tlc)) & 0xFFull) == 0xFFull;
            }
            bool okt = ok;
            if constexpr (LPT == 8) okt = (okmask >> (8 * tlc)) & 1ull;  // lane 0 of the group is a real axis pair
            if ((lane % LPT) == 0 && tl < nv && a.status) a.status[base + tl] = okt ? (fin ? UAVQP_SOLVED : UAVQP_NON_FINITE) : UAVQP_INVALID_INPUT;
        }
        UAVQP_STAMP(4);
        wave_lds_sync();
    }
}

}  // namespace uavqp
