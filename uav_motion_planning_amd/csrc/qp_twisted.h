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
#include <type_traits>

#include "qp_core_kernels.h"
#include "qp_wave_utils.h"

#ifndef UAVQP_STAMP   // (probe builds under tools/ubench define their own: per-wave timelines)
#ifdef UAVQP_PHASE_TIMING
#define UAVQP_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define UAVQP_STAMP(i) do {} while (0)
#endif
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
    static constexpr int PQ = NC / 2;  // 16-byte pieces per chunk
    // LPT = lanes per trajectory.  2: one (L, R) lane pair carries all three axes (throughput shape).
    // 8: one lane pair per axis (+ one idle pair) -- the matrix elimination is repeated by the three pairs,
    // right-hand sides, back-substitution and coefficient emission are split by axis (latency shape for
    // small batches: ~2x shorter critical path per wave, 8 trajectories per wave).
    static constexpr int NAX = LPT == 2 ? 3 : 1;
    // Emission chunk: CH own segments of one axis per unit (r = 4: 2 segments = one 128-B line; r = 3: 4
    // segments = 192 B).  Staging = 64 lane rows of CH*NC doubles, row stride 4 (mod 8) dwords for
    // conflict-free ds_write_b128.  If double-buffered input + staging exceed the LDS budget of 4 waves
    // per CU (40448 B each), the staging rows alias the current input buffer instead.
    static constexpr int CH = (NC * 8) % 64 == 0 ? 2 : 4;
    static constexpr int STAGE_STRIDE0 = CH * NC + 2;
    static constexpr int STAGE_STRIDE = (STAGE_STRIDE0 * 2) % 8 == 4 ? STAGE_STRIDE0 : STAGE_STRIDE0 + 2;
    static constexpr int STAGE_D = 64 * STAGE_STRIDE;
    // (measured, 1 M batch: aliasing wins for r = 3 -- (3,16) 3.85 vs 3.71, (3,12) 3.97 vs 3.43 TB/s -- but the
    // register cost of the pre-loaded positions spills for r = 4, M = 12: 3.46 vs 3.92 TB/s with 3 waves/CU)
    static constexpr bool ALIAS = LPT == 2 && R == 3 && (2 * IN_D + STAGE_D) * 8 > 40448;
    // LPT >= 8 (latency shapes): every emission step leaves one NC-double chunk per working lane (6 * LPT / 8 per trajectory:
    // TILE * 6 * LPT / 8 = 48 rows for both shapes) in an LDS row of stride 10 doubles (20 dwords: the b128 writes of 16 rows tile
    // the 64 banks), read back linearly by the wave for the software-pipelined copy-out; rows 48.. take the writes of the idle pairs.
    static constexpr int ROWS8 = LPT >= 8 ? TILE * 6 * (LPT / 8) : 0, RS8 = 10;
    static_assert(LPT < 8 || ROWS8 == 48, "the copy-out maps 3 axes x 16 rows");
    static constexpr int OUT_D = LPT >= 8 ? (ROWS8 + 16) * RS8 + 16 : ((LPT == 2 && !ALIAS) ? STAGE_D : 2);
    static_assert(!ALIAS || STAGE_D <= IN_D, "aliased staging rows must fit the input buffer");
    static_assert(WP_D % 2 == 0 && T_D % 2 == 0 && BC_D % 2 == 0, "tile arrays must be whole 16-B pairs");
};

template <int R, int M, int TILE, int LPT = 2>
__global__ __launch_bounds__(64, 1) void solve_twisted_kernel(BatchArgs a) {
    using C = TwistedCfg<R, M, TILE, LPT>;
    constexpr int NAX = C::NAX;
    constexpr int ND = C::ND, NC = C::NC, NK = C::NK, mL = C::mL, mR = C::mR;
    static_assert(M >= 2, "twisted kernel needs an interior knot");
    static_assert((LPT == 2 && (TILE == 32 || TILE == 16)) || (LPT == 8 && TILE == 8) || (LPT == 16 && TILE == 4), "tile shapes: 2 lanes x 32|16, 8 lanes x 8, 16 lanes x 4");

    __shared__ __attribute__((aligned(16))) double s_in[2][C::IN_D];
    __shared__ __attribute__((aligned(16))) double s_out[C::OUT_D];
    (void)s_out;

    const int lane = threadIdx.x;
    const int isR = lane & 1;
    const int tl = lane / LPT;
    const int tlc = tl < TILE ? tl : TILE - 1;  // clamp LDS indexing of idle lanes (TILE == 16)
    const int axl = ((lane % LPT) >> 1) & 3;     // LPT >= 8: axis of this lane pair (3 = idle pair)
    const int sub = LPT == 16 ? (lane >> 3) & 1 : 0;   // LPT == 16: which half of the own segments this lane pair emits
    const int ax0 = LPT == 2 ? 0 : (axl < 3 ? axl : 2);
    const int m = isR ? mR : mL;
    const int n_tiles = (a.n_traj + TILE - 1) / TILE;

    // A partial last tile is SHIFTED back so that it ends at the batch end (n_traj >= TILE): it stays on the LDS-DMA path and
    // re-solves a few trajectories of its neighbour -- identical values written twice -- instead of sending one wave through the
    // guarded element-wise loads (measured on 4093 trajectories: 6.43 -> 5.03 us: that one wave was the kernel's critical path).
    // Batches smaller than one tile keep the guarded path.
    auto tile_base = [&](int tile) {
        const int b_ = tile * TILE;
        return (a.n_traj >= TILE && b_ + TILE > a.n_traj) ? a.n_traj - TILE : b_;
    };
    constexpr int DMA_AUX = LPT >= 8 ? 2 : 0;
    auto issue_tile = [&](int tile, int buf) {
        const int base = tile_base(tile);
        const int nv = min(TILE, a.n_traj - base);
        double* s = s_in[buf];
        if (nv == TILE) {
            dma_tile<C::WP_D, DMA_AUX>(a.waypoints + (size_t)base * NK * 3, s, lane);
            dma_tile<C::T_D, DMA_AUX>(a.times + (size_t)base * M, s + C::T_OFF, lane);
            dma_tile<C::BC_D, DMA_AUX>(a.bc + (size_t)base * 2 * ND * 3, s + C::BC_OFF, lane);
        } else {
            load_tile_guarded<C::WP_D>(a.waypoints + (size_t)base * NK * 3, nv * NK * 3, s, lane, 0.0);
            load_tile_guarded<C::T_D>(a.times + (size_t)base * M, nv * M, s + C::T_OFF, lane, 1.0);
            load_tile_guarded<C::BC_D>(a.bc + (size_t)base * 2 * ND * 3, nv * 2 * ND * 3, s + C::BC_OFF, lane, 0.0);
        }
    };

    int buf = 0;
    UAVQP_STAMP(5);
    if ((int)blockIdx.x < n_tiles) issue_tile(blockIdx.x, 0);
    UAVQP_STAMP(6);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        // The wait for the FIRST tile sits inside the loop: everything loop-invariant (lane indices, table constants, addresses --
        // ~130 instructions) lands in the loop pre-header, i.e. between the issue of the first loads and this wait, instead of
        // behind it (measured: load wait 2150 -> 1230 cycles per wave, 5.49 -> 5.05 us on the 4096 batch).
        if (tile == (int)blockIdx.x) {
            wait_vmcnt0();
            wave_lds_sync();
        }
        const int base = tile_base(tile);
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
        // (LPT == 2: the own durations stay in registers for the emission -- the duration tile has a row stride of M doubles, a
        // 4-way bank conflict per read for M = 8)
        // (LPT == 8: also their inverse powers T^-R..T^-(2R-1), which SegBlocks::build computes anyway and the emission needs again)
        constexpr bool KEEP_T = LPT == 2 || LPT == 8;
        double Town[KEEP_T ? mL : 1], Tip[LPT == 8 ? mL : 1][R];
#pragma unroll
        for (int j = 0; j < (KEEP_T ? mL : 1); ++j) Town[j] = 1.0;
#pragma unroll
        for (int j = 0; j < (LPT == 8 ? mL : 1); ++j)
#pragma unroll
            for (int k = 0; k < R; ++k) Tip[j][k] = 1.0;
        SegBlocks<R> sa;
        {
            const double t0 = Tof(0);
            Town[0] = t0;
            double ipw[NC];
            sa.build(t0, ipw);
            if constexpr (LPT == 8) {
#pragma unroll
                for (int k = 0; k < R; ++k) Tip[0][k] = ipw[R + k];
            }
        }
        double pb[NAX], dpa[NAX];
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) {
            pb[ax] = pos(1, ax);
            dpa[ax] = pb[ax] - pos(0, ax);
        }
#pragma unroll
        for (int j = 1; j < mL; ++j) {
#if defined(UAVQP_TW_SKIP_LAST_ELIM) && !defined(UAVQP_TIMING_BUILD)
#error "UAVQP_TW_SKIP_LAST_ELIM produces WRONG results: timing harness only (tools/ubench/tw, built with -DUAVQP_TIMING_BUILD), never libuavqp.so"
#endif
#ifdef UAVQP_TW_SKIP_LAST_ELIM   // timing-only build (tools/ubench/tw: h_16s): the last elimination step takes its inputs from TWO knots back, so it no
            // longer waits for the step before it -- the same work on a dependent chain one level shorter, i.e. what a cyclic reduction could buy
            // AT MOST (its exchanges not counted); the results are wrong
            constexpr bool tw_short = true;
#else
            constexpr bool tw_short = false;
#endif
            const int jp = (tw_short && j == mL - 1 && j >= 2) ? j - 2 : j - 1;
            if (j < m) {
                SegBlocks<R> sb;
                {
                    const double tj = Tof(j);
                    if constexpr (KEEP_T) Town[j] = tj;
                    double ipw[NC];
                    sb.build(tj, ipw);
                    if constexpr (LPT == 8) {
#pragma unroll
                        for (int k = 0; k < R; ++k) Tip[j][k] = ipw[R + k];
                    }
                }
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
                            for (int c = 0; c <= i; ++c) S[i][c] -= sa.A01[q][i] * E[jp][q][c];
                        }
#pragma unroll
                        for (int ax = 0; ax < NAX; ++ax) z[i][ax] -= sa.A01[q][i] * h[jp][q][ax];
                    }
                typename std::conditional<LPT >= 8, SymInv<ND>, SmallLDL<ND>>::type ldl;
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
            typename std::conditional<LPT >= 8, SymInv<ND>, SmallLDL<ND>>::type ldl;
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

        if constexpr (LPT >= 8) {
            // one axis per lane pair.  Back-substitution first (in place, h[j] <- y_j), so that the mL segment evaluations below are
            // independent of each other and the scheduler can overlap their dependent FP64 chains.
#pragma unroll
            for (int j = mL - 1; j >= 1; --j) {
                if (j < m) {
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int c = 0; c < ND; ++c) {
                            const double src = (j == mL - 1 || j == m - 1) ? ynext[c][0] : h[j + 1 < mL ? j + 1 : j][c][0];
                            h[j][i][0] -= E[j][i][c] * src;
                        }
                }
            }
            // LPT == 16: two lane pairs per (trajectory, axis) repeat elimination and back-substitution and split the EMISSION --
            // pair `sub` evaluates the own segments sub * H .. sub * H + H - 1 (selects between the two compile-time candidates, one
            // instruction stream): 4 trajectories per wave, a 4096-trajectory batch fills all 1024 SIMDs.
            //
            // Copy-out, software-pipelined (what the per-wave timelines of tools/ubench/tw asked for: a vector-memory store costs the
            // wave ~100 cycles of issue, sixteen scattered quarter-chunk stores per lane were 1100 of the emission's 3400 cycles and left
            // 6.3 MB dirty in L2 for the kernel boundary to write back):
            //   * the NC-double chunk a lane produces in step s goes to its own LDS row at the end of the step;
            //   * at the START of step s + 1 the wave reads the 48 rows back (16 bytes per lane: whole chunks from adjacent lanes,
            //     one axis per instruction) and its three write-through stores are interleaved with that step's arithmetic (sched_group_barrier): only the
            //     last step's stores are exposed;
            //   * buffer stores through a per-tile descriptor: pieces of an invalid trajectory, of a lane without a segment in that
            //     step (odd M) and -- for batches smaller than a tile -- of trajectories past the end get an out-of-range offset and
            //     are dropped by the hardware: no branch, no sink traffic.
            // 4096 x (M = 8, r = 4), us per launch over rotating buffers: 6.45 (direct stores) -> 5.80 (two-step chunks through LDS,
            // write-through) -> 5.52 (this) for 8 lanes per trajectory, 5.30 for 16.
            constexpr int NSUB = LPT / 8, H = (mL + NSUB - 1) / NSUB;
            // LDS rows: axis-major, 16 rows per axis (one per (trajectory, sub, side) = `rest`), so that copy instruction `it` moves
            // axis `it` and everything else about a piece comes from lane bits: rest = lane >> 2, 16-byte column = lane & 3 (a row
            // is padded to four pieces; r = 3 uses three).  Row stride 10 doubles + 4 doubles per axis: the 12 working lanes of any
            // 16 consecutive lanes start their b128 writes in 12 different bank quads.
            constexpr int RS = C::RS8, NIT = 3;
            auto row_addr = [&](int cax, int rest) -> int { return (cax * 16 + rest) * RS + cax * 4; };
            const unsigned tile_bytes = (unsigned)nv * 3u * M * NC * 8u;
            const unsigned long long obase = (unsigned long long)out;
            const unsigned ob_lo = __builtin_amdgcn_readfirstlane((unsigned)obase), ob_hi = __builtin_amdgcn_readfirstlane((unsigned)(obase >> 32));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)ob_hi << 32) | ob_lo), 0,
                                                                __builtin_amdgcn_readfirstlane(tile_bytes), 0x00020000);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            // the piece this lane copies (the same row slot for the three axes)
            const int c_rest = lane >> 2, c_col = lane & 3;
            const int c_R = c_rest & 1, c_sub = NSUB == 2 ? (c_rest >> 1) & 1 : 0, c_tl = c_rest >> NSUB;
            const int c_m = c_R ? mR : mL;
            const bool c_keep = (c_col < NC / 2) && ((okmask >> (LPT * c_tl)) & 1ull);
            const unsigned c_off = (unsigned)(((c_tl * 3 * M) * NC + 2 * c_col) * 8);   // + axis * M * NC * 8 + segment * NC * 8
            const int c_lds = c_rest * RS + 2 * c_col;
            const int my_rest = NSUB == 2 ? tlc * 4 + sub * 2 + isR : tlc * 2 + isR;
            const int my_lds = axl < 3 ? row_addr(axl, my_rest) : 48 * RS + 16 + (lane & 15) * RS;
            auto flush = [&](int sp) {   // copy-out of step sp: NIT LDS reads, NIT stores
                double2 v[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const double2*>(s_out + row_addr(it, 0) + c_lds);
                const int jv = c_sub * H + sp;                                    // own segment of the producing lane
                const int cseg = c_R ? (M - 1 - jv) : jv;
                const unsigned off0 = (c_keep && jv < c_m) ? c_off + (unsigned)(cseg * NC * 8) : 0xFFFFFFF0u;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    u32x4 w;
                    w.x = (unsigned)__double2loint(v[it].x); w.y = (unsigned)__double2hiint(v[it].x);
                    w.z = (unsigned)__double2loint(v[it].y); w.w = (unsigned)__double2hiint(v[it].y);
                    // (an out-of-range base stays out of range: 0xFFFFFFF0 + it * M * NC * 8 wraps only for absurd M)
                    const unsigned off = off0 == 0xFFFFFFF0u ? off0 : off0 + (unsigned)(it * M * NC * 8);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, off, 0, 17 /* sc0 sc1: write-through, nothing left dirty for the kernel boundary */);
                }
            };
#pragma unroll
            for (int s = 0; s < H; ++s) {
                if (s > 0) flush(s - 1);
                const int j = s + sub * H;
                const bool act = (j < m);
                const int jc = act ? j : (m > 0 ? m - 1 : 0);
                const double pj = pos(jc, 0), pj1 = pos(jc + 1, 0);
                double ys[ND], ye[ND], c8[NC];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                    auto Y = [&](int q) -> double { return (q >= mL || q == m) ? ynext[d][0] : h[q < mL ? q : mL - 1][d][0]; };
                    double yj = h[s][d][0], yj1 = Y(s + 1);
                    if constexpr (NSUB == 2) {
                        const int jb = (H + s < mL) ? H + s : mL - 1;
                        // (the candidates are pinned as VALUES first: left alone, the optimiser turns a select between two elements of h
                        // into ONE load through a selected pointer before the loops are unrolled -- and h, indexed at run time, goes to
                        // scratch: 32-48 bytes per lane in the M = 3, 4, 5 shapes)
                        double alt = h[jb][d][0], alt1 = Y(H + s + 1);
                        asm("" : "+v"(alt));
                        asm("" : "+v"(yj));
                        asm("" : "+v"(alt1));
                        asm("" : "+v"(yj1));
                        yj = sub ? alt : yj;
                        yj1 = sub ? alt1 : yj1;
                    }
                    ys[d] = isR ? fs * yj1 : yj;
                    ye[d] = isR ? fs * yj : yj1;
                }
                if constexpr (LPT == 8) {
                    // the duration and its inverse powers were kept from the elimination (SegBlocks::build): no reciprocal, no power chain
                    double ipj[NC];
#pragma unroll
                    for (int k = 0; k < R; ++k) { ipj[k] = 1.0; ipj[R + k] = Tip[s][k]; }
                    segment_coeffs_ip<R>(isR ? pj1 : pj, ys, isR ? pj : pj1, ye, Town[s], ipj, c8);
                } else {
                    const double Tj = Tof(jc);
                    segment_coeffs<R>(isR ? pj1 : pj, ys, isR ? pj : pj1, ye, Tj, fast_rcp(Tj), c8);
                }
                if (act) finite = finite && (fabs(c8[NC - 1]) < INFINITY) && (fabs(c8[R]) < INFINITY);
                if (s > 0) {
                    // the previous step's copy-out between this step's arithmetic: LDS reads first, then one store per ~24 VALU instructions
                    __builtin_amdgcn_sched_group_barrier(0x100, NIT, 0);
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        __builtin_amdgcn_sched_group_barrier(0x2, 24, 0);
                        __builtin_amdgcn_sched_group_barrier(0x40, 1, 0);
                    }
                }
                {
                    // (idle pairs and lanes without a segment in this step write too -- rows / chunks that are never copied out)
                    double* so = s_out + my_lds;
#pragma unroll
                    for (int k = 0; k < NC; k += 2) *reinterpret_cast<double2*>(so + k) = make_double2(c8[k], c8[k + 1]);
                }
                // single-wave workgroup: LDS operations execute in issue order; this only pins the compiler's order of the accesses
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (s == H - 1) flush(s);
            }
        } else {
            // ---------------- chunk mode (LPT == 2): CH own segments x one axis per emission unit ----------------
            // Every lane writes the 2r*CH coefficients of its chunk, in ORIGINAL segment order, to its own LDS
            // row; the wave then copies the 64 rows out linearly, 16 B per lane: a lane row is one contiguous
            // run in HBM (r = 4, CH = 2: exactly one 128-B line; r = 3, CH = 4: 192 B, and when CH covers the
            // whole half the L and R rows of a trajectory are adjacent, i.e. whole (trajectory, axis) rows).
            // Units are axis-major so that the fragments of a line shared by two chunks are written close
            // together.  Measured on the store pattern alone (tools/ubench/write_patterns.hip): 5.3-5.5 TB/s
            // for line-complete runs vs 3.0-3.5 TB/s when half lines / 48-B chunks arrive a segment apart.
            constexpr int CH = C::CH, NQ = (mL + CH - 1) / CH, RS = C::STAGE_STRIDE, PPL = CH * NC / 2;
            // in-place back-substitution: h[j] <- y_j for the own interior knots j = m-1 .. 1
#pragma unroll
            for (int j = mL - 1; j >= 1; --j) {
                if (j < m) {
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int c = 0; c < ND; ++c)
#pragma unroll
                            for (int ax = 0; ax < NAX; ++ax) {
                                const double src = (j == mL - 1 || j == m - 1) ? ynext[c][ax] : h[j + 1 < mL ? j + 1 : j][c][ax];
                                h[j][i][ax] -= E[j][i][c] * src;
                            }
                }
            }
            double* stage = C::ALIAS ? s_in[buf] : s_out;
            // When the staging rows alias the current input buffer, everything emission still needs from it
            // is pulled into registers first.
            double P_[C::ALIAS ? mL + 1 : 1][3], T_[C::ALIAS ? mL : 1];
            if constexpr (C::ALIAS) {
#pragma unroll
                for (int j = 0; j <= mL; ++j)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) P_[j][ax] = pos(j <= m ? j : m, ax);
#pragma unroll
                for (int j = 0; j < mL; ++j) T_[j] = Tof(j < m ? j : m - 1);
                wave_lds_sync();
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
                for (int q = NQ - 1; q >= 0; --q) {
                    const int cnt = max(0, min((q + 1) * CH, m) - q * CH);  // own segments in this chunk (lane-dependent for odd M)
                    double* row = stage + lane * RS;
#pragma unroll
                    for (int s = 0; s < CH; ++s) {
                        const int j = q * CH + s;  // own segment
                        if (j < mL) {
                            const int jc = j < m ? j : (m > 0 ? m - 1 : 0);
                            double ys[ND], ye[ND], c8[NC];
#pragma unroll
                            for (int d = 0; d < ND; ++d) {
                                const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                                const double yj = h[j][d][ax];
                                const double yj1 = (j + 1 >= mL || j + 1 == m) ? ynext[d][ax] : h[j + 1 < mL ? j + 1 : j][d][ax];
                                ys[d] = isR ? fs * yj1 : yj;
                                ye[d] = isR ? fs * yj : yj1;
                            }
                            double pa, pb, Tj;
                            if constexpr (C::ALIAS) { pa = P_[j][ax]; pb = P_[j + 1][ax]; Tj = T_[j]; }
                            else { pa = pos(jc, ax); pb = pos(jc + 1, ax); Tj = Town[j]; }
                            segment_coeffs<R>(isR ? pb : pa, ys, isR ? pa : pb, ye, Tj, fast_rcp(Tj), c8);
                            if (j < m) finite = finite && (fabs(c8[NC - 1]) < INFINITY) && (fabs(c8[R]) < INFINITY);
                            // slot in original order: L ascending, R descending within the chunk
                            const int slot = isR ? (cnt - 1 - s) : s;
                            if (s < cnt) {
                                double* so = row + slot * NC;
#pragma unroll
                                for (int k = 0; k < NC; k += 2) *reinterpret_cast<double2*>(so + k) = make_double2(c8[k], c8[k + 1]);
                            }
                        }
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int it = 0; it < PPL; ++it) {
                        // which 16-byte piece a lane copies.  Plain order (lane it*64 + l -> piece l of 8 consecutive rows)
                        // is 2-way conflicted for 128-byte rows at the write-friendly stride of 36 dwords: ds_read_b128
                        // serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) over 64 banks, and the four
                        // half rows a group reads must tile them -- which at 9 bank-quads per row only both halves of
                        // rows rho and rho + 8 do.  So instruction `it` takes rows it + 8 k (k = 0..7), quad q of a
                        // 32-lane half reads half (q >> 1) & 1 of row it + 8 * {0, 2, 2, 0, 3, 1, 1, 3}[q]: every row is
                        // still written out as one complete 128-byte line (SQ_LDS_BANK_CONFLICT 34 % -> see profiles/).
                        int pl, col;
                        if constexpr (PPL == 8 && TILE == 32) {
                            const int kq = (lane >> 2) & 7;
                            pl = it + 8 * ((0xD728 >> (2 * kq)) & 3) + (lane & 32);
                            col = ((kq >> 1) & 1) * 4 + (lane & 3);
                        } else if constexpr (PPL == 12 && TILE == 32) {
                            // 192-byte rows at 52 dwords (13 bank-quads): a read group tiles the banks with the same third
                            // of four rows 4 apart -- instruction `it` copies third it % 3 of rows gamma + 16 (it / 3) + 4 i,
                            // gamma = the lane group, i = the quad's place in it
                            const int kq = lane >> 2, k7 = kq & 7;
                            const int gam = (kq >> 3) * 2 + ((0x96 >> k7) & 1), i4 = k7 >> 1;
                            pl = gam + 16 * (it / 3) + 4 * i4;
                            col = 4 * (it % 3) + (lane & 3);
                        } else {
                            const int g = it * 64 + lane;
                            pl = g / PPL;
                            col = g - pl * PPL;
                        }
                        const int ctl = pl >> 1, cR = pl & 1;
                        const int cm = cR ? mR : mL;
                        const int ccnt = max(0, min((q + 1) * CH, cm) - q * CH);
                        const int first = cR ? (M - q * CH - ccnt) : q * CH;
                        const double2 v = *reinterpret_cast<const double2*>(stage + pl * RS + 2 * col);
                        const bool keep = (2 * col < ccnt * NC) && (ctl < TILE) && ((okmask >> (2 * (ctl < TILE ? ctl : 0))) & 1ull);
                        // (the explicit vmcnt(0) above is a builtin, so the compiler knows no LDS-DMA is pending here
                        //  and a predicated store does not make it re-insert waits before the LDS reads)
                        if (keep) store_pair_wt(out + (((size_t)ctl * 3 + ax) * M + first) * NC + 2 * col, v);
                    }
                    wave_lds_sync();  // rows are rewritten by the next unit: keep the reads above it
                }
            }
        }
        {
            const int f = finite ? 1 : 0;
            const int other = __builtin_amdgcn_mov_dpp(f, 0xB1, 0xF, 0xF, true);
            bool fin = (f & other) != 0;
            if constexpr (LPT >= 8) {  // all axis pairs of the trajectory must be finite
                const unsigned long long fm = __ballot(fin || axl == 3);
                constexpr unsigned long long grp = LPT == 8 ? 0xFFull : 0xFFFFull;
                fin = ((fm >> (LPT * tlc)) & grp) == grp;
            }
            bool okt = ok;
            if constexpr (LPT >= 8) okt = (okmask >> (LPT * tlc)) & 1ull;  // lane 0 of the group is a real axis pair
            if ((lane % LPT) == 0 && tl < nv && a.status) a.status[base + tl] = okt ? (fin ? UAVQP_SOLVED : UAVQP_NON_FINITE) : UAVQP_INVALID_INPUT;
        }
        UAVQP_STAMP(4);
        wave_lds_sync();
    }
}

}  // namespace uavqp
