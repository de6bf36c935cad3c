// qp_corridor_lane.h -- the corridor prelude (qp_corridor_dual.h: the working set a dual active-set method in position space ends with) with ONE
// LANE PER TRAJECTORY (round 5).
//
// corridor_dual_kernel runs eight lanes per trajectory: the block LDL' chain that gives G = [H^-1]_pp is repeated in every lane of a
// group (1/8 useful), and a wave advances at the pace of the slowest of its eight groups (11.5 trips per axis against a mean of 6.9 on
// BASELINE config 3) -- per trip ~375 instructions of selection, ratio test, column hand-over through LDS and butterflies around a sweep of
// 30.  Here a lane owns a whole trajectory:
//   * the chain runs once per trajectory, in its lane, with nothing replicated;
//   * the tableau of a problem -- n <= 15 knot positions, symmetric: the 105 entries below the diagonal -- lives in LDS columns that only
//     its lane touches ([pair of entries][lane][2]: every access is a conflict-free ds_read/write_b128 at 16 bytes per lane; 53 KiB per
//     wave, three waves per CU), the diagonal and the per-knot vectors (position / multiplier, bounds) in registers with compile-time
//     indices;
//   * a pivot is a per-lane run-time index: its column is 14 LDS reads at computed addresses, the rank-one update of the triangle 106 FMAs
//     on compile-time pairs -- no lane ever waits for another lane's pivot, there is no group to be in lockstep with; a lane takes its
//     three axes one after the other, each from a copy of G kept in a per-wave HBM scratch (same [pair][lane][2] layout: coalesced);
//   * it is also the preparation kernel (CorridorArgs::prep_in_dual), as corridor_dual_kernel is: validation, status / iteration reset,
//     problem descriptors, one-segment trajectories.
// The tableau is kept in SINGLE precision (positions, multipliers, bounds, step lengths and the chain stay in float64): 27 KiB of LDS per
// wave instead of 54 -- six waves per CU instead of three, which is what this kernel needs: a wave's trip is a chain of dependent
// selections and LDS round trips (first version, float64 tableau, three waves per CU: 629 us on config 3 against the group kernel's
// 338).  A rounding error of the tableau can only make the set a near-degenerate problem ends with differ from the exact one by a
// bound that is active with a multiplier of ~1e-6 relative -- and then the verifying solve makes one more iteration.
// As before NOTHING HERE DECIDES A RESULT: the set goes to corridor_solve_kernel as its cold-start guess and is verified there by an
// exact block solve.  Taken for batches whose trajectories have at most 16 segments and no G cache (BASELINE config 3; the first solve of
// a ragged config-5 batch and its re-solves on the cached G stay with the group / wave kernels).
#pragma once
#include <utility>

#include "qp_corridor_dual.h"

namespace uavqp {

constexpr int LANE_NV = 15;                                   // variables = interior knots per trajectory, at most
constexpr int LANE_NT = LANE_NV * (LANE_NV - 1) / 2;          // entries below the diagonal: (r, c), r > c, at r (r - 1) / 2 + c
constexpr int LANE_NP = (LANE_NT + 1) / 2;                    // pairs of them
// per-wave scratch in doubles: G below the diagonal (as pairs), its diagonal, the unconstrained minimiser of the three axes, the chain
// records (S_k^-1, E_k per knot); every field [index][lane]: all 64 lanes of an access hit 512 contiguous bytes
constexpr int lane_rec_doubles(int R) { return R * (R + 1) / 2 + R * R; }
constexpr int lane_scratch_doubles(int R) { return (LANE_NP + LANE_NV + 9 * LANE_NV + LANE_NV * lane_rec_doubles(R)) * 64; }     // (G: float pairs; y0, lo, hi: 3 axes each)

constexpr int lane_tri_row(int e) { int r = 1; while (r * (r + 1) / 2 <= e) ++r; return r; }      // entry e of the strict lower triangle sits in row r, column e - r (r - 1) / 2
// the rank-one update of the two entries of pair P (rows and columns are compile-time constants: tc / sc stay in registers)
typedef float2 __attribute__((may_alias)) float2_a;
template <int P>
__device__ __forceinline__ void lane_sweep_pair(float2_a* sT, const float (&tc)[LANE_NV], const float (&sc)[LANE_NV]) {
    constexpr int e0 = 2 * P, r0 = lane_tri_row(e0), c0 = e0 - r0 * (r0 - 1) / 2, e1 = 2 * P + 1;
    float2 v = sT[P * 64];
    v.x = fmaf(-tc[r0], sc[c0], v.x);
    if constexpr (e1 < LANE_NT) {
        constexpr int r1 = lane_tri_row(e1), c1 = e1 - r1 * (r1 - 1) / 2;
        v.y = fmaf(-tc[r1], sc[c1], v.y);
    }
    sT[P * 64] = v;
}
// A whole batch of pairs: every load first, then the FMAs, then the stores.  (Pair by pair the compiler keeps each LDS load behind the
// store of the pair before it -- it cannot see that the addresses differ -- and a trip paid 27 LDS round trips: 7.9 k cycles.)
template <int P0, int... P>
__device__ __forceinline__ void lane_sweep_batch(float2_a* sT, const float (&tc)[LANE_NV], const float (&sc)[LANE_NV], std::integer_sequence<int, P...>) {
    float2 v[sizeof...(P)];
    ((v[P] = sT[(P0 + P) * 64]), ...);
    lds_publish();      // (every load is issued before the first store: the compiler would interleave them again, two or three in flight)
    auto upd = [&](auto pc) {
        constexpr int Pq = P0 + decltype(pc)::value;
        constexpr int e0 = 2 * Pq, r0 = lane_tri_row(e0), c0 = e0 - r0 * (r0 - 1) / 2, e1 = 2 * Pq + 1;
        v[decltype(pc)::value].x = fmaf(-tc[r0], sc[c0], v[decltype(pc)::value].x);
        if constexpr (e1 < LANE_NT) {
            constexpr int r1 = lane_tri_row(e1), c1 = e1 - r1 * (r1 - 1) / 2;
            v[decltype(pc)::value].y = fmaf(-tc[r1], sc[c1], v[decltype(pc)::value].y);
        }
    };
    (upd(std::integral_constant<int, P>{}), ...);
    lds_publish();
    ((sT[(P0 + P) * 64] = v[P]), ...);
}

#ifdef UAVQP_LANE_TIMING   // probe build (tools/lane_sections.py): cycles of block 0 per section -> queue block, bytes 128..
#define LN_T_DECL long long ln_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ln_t = __builtin_readcyclecounter();
#define LN_T(k) do { const long long n_ = __builtin_readcyclecounter(); ln_acc[k] += n_ - ln_t; ln_t = n_; } while (0)
#define LN_T_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) reinterpret_cast<long long*>(a.queue)[16 + k_] = ln_acc[k_]; } while (0)
#else
#define LN_T_DECL
#define LN_T(k) do {} while (0)
#define LN_T_FLUSH do {} while (0)
#endif

template <int R>
__global__ __launch_bounds__(64, 1) void corridor_dual_lane_kernel(CorridorArgs a, double* __restrict__ scratch, int max_trips_extra) {
    constexpr int NV = LANE_NV, NT = LANE_NT, NP = LANE_NP, ND = R - 1, NE = R * (R + 1) / 2;
    (void)NE; (void)NT;
    using Inv = SmallLDL<R>;
    __shared__ __attribute__((aligned(16))) float s_T[NP * 64 * 2];
    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && lane == 0) *a.queue = 0u;      // the work counter of the solve kernel that follows
    double* const gw = scratch + (size_t)blockIdx.x * lane_scratch_doubles(R);
    float2_a* const gT = reinterpret_cast<float2_a*>(gw) + lane;                      // pair p at gT[p * 64]
    double* const gD = gw + NP * 64 + lane;                                           // diagonal entry j at gD[j * 64]
    double* const gY = gw + (NP + NV) * 64 + lane;                                    // y0 of (axis, j) at gY[(axis * NV + j) * 64]
    double* const gLo = gw + (NP + NV + 3 * NV) * 64 + lane;                          // box of (axis, j), same indexing
    double* const gHi = gw + (NP + NV + 6 * NV) * 64 + lane;
    float2_a* const sT = reinterpret_cast<float2_a*>(s_T) + lane;                     // pair p at sT[p * 64]
    float* const sE = s_T + 2 * lane;                                                 // entry e at sE[(e >> 1) * 128 + (e & 1)]

    const int n_eff = a.n_active ? *a.n_active : a.n_traj;
    const long long n_chunks = ((long long)n_eff + 63) / 64;
    LN_T_DECL
    for (long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        LN_T(7);
        // ================= this lane's trajectory: validation, reset, descriptors (the duties of corridor_prep_kernel) =================
        const long long bq = ch * 64 + lane;
        const bool have = bq < n_eff;
        const int b = have ? (a.order ? a.order[bq] : (int)bq) : 0;
        int s0 = 0, M = 2;
        if (have) { if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; } }
        const bool shape_ok = have && (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;
        const bool fits = shape_ok && M >= 2 && M - 1 <= NV;
        const double* const TT = a.times + s0;
        // (fixed trip counts with guards: the loads of a lane are in flight together instead of one round trip per segment)
        bool t_ok = shape_ok;
        double Tr[NV + 1];
#pragma unroll
        for (int i = 0; i <= NV; ++i) Tr[i] = (shape_ok && i < M) ? TT[i] : 1.0;
#pragma unroll
        for (int i = 0; i <= NV; ++i) t_ok = t_ok && (Tr[i] > 0.0) && (Tr[i] < INFINITY);
        if (shape_ok && M > NV + 1)
            for (int i = NV + 1; i < M; ++i) { const double t = TT[i]; t_ok = t_ok && (t > 0.0) && (t < INFINITY); }      // (longer than any tableau: invalid for this kernel's batches anyway)
        bool solve_any = fits && t_ok;
        unsigned long long dsc[3] = {0ull, 0ull, 0ull};
        // ---- the boxes of the chunk: 64 lanes x 15 knots x 3 axes, 24-byte pieces 400 bytes apart from lane to lane -- read directly, every load
        // instruction touches 64 cache lines (the first version spent a quarter of its time on them: here and again at every axis start).  Without a
        // dealing order the chunk's waypoint rows are ONE contiguous run: it goes through LDS (the tableau is not in use yet) in coalesced
        // 512-byte loads, every lane picks its own values up there and leaves them in the per-wave scratch as [axis][knot][lane].
        {
            const bool contiguous = a.order == nullptr;
            const long long first = __builtin_amdgcn_readfirstlane((int)(ch * 64));      // (dealing position = trajectory index)
            long long row0 = 0, row1 = 0;
            if (contiguous) {
                const int bl = (int)min((long long)n_eff - 1, first + 63);
                if (a.uniform > 0) { row0 = first * (a.uniform + 1); row1 = ((long long)bl + 1) * (a.uniform + 1); }
                else { row0 = (long long)a.seg_offsets[first] + first; row1 = (long long)a.seg_offsets[bl + 1] + bl + 1; }
            }
            const long long nd = 3 * (row1 - row0);                                      // doubles of the run
            const bool staged = contiguous && nd > 0 && nd <= (long long)(NP * 64);      // (fits the tableau's LDS: 64 x 17 x 3 doubles and a little more)
            double* const sD = reinterpret_cast<double*>(s_T);
            const long long myrow = (long long)s0 + b - row0;                            // first waypoint row of this lane's trajectory inside the run
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                const double* const src = which == 0 ? a.corr_lo : a.corr_hi;
                double* const dst = which == 0 ? gLo : gHi;
                if (staged) {
                    lds_publish();
#pragma unroll 1
                    for (int bt = 0; bt < NP; bt += 9) {      // (nine loads in flight, then their LDS stores: bounded register use)
                        double tmp[9];
#pragma unroll
                        for (int i9 = 0; i9 < 9; ++i9) { const long long i = (long long)(bt + i9) * 64 + lane; tmp[i9] = (bt + i9 < NP && i < nd) ? src[3 * row0 + i] : 0.0; }
#pragma unroll
                        for (int i9 = 0; i9 < 9; ++i9) if (bt + i9 < NP) sD[(bt + i9) * 64 + lane] = tmp[i9];
                    }
                    lds_publish();
                }
#pragma unroll 1
                for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
                    for (int k = 1; k <= NV; ++k) {
                        double v = 0.0;
                        if (fits && k < M) v = staged ? sD[3 * (myrow + k) + ax] : src[3LL * ((long long)s0 + b + k) + ax];
                        dst[(ax * NV + (k - 1)) * 64] = v;
                    }
                }
                lds_publish();
            }
        }
        if (solve_any) {
            // boxes: lo <= hi at every interior knot of every axis or the trajectory is invalid as a whole; lo == hi: an equality row
            bool bad = false;
#pragma unroll 1
            for (int ax = 0; ax < 3; ++ax) {
                unsigned long long eq_ = 0ull;
#pragma unroll
                for (int k = 1; k <= NV; ++k) {
                    if (k < M) {
                        const double l_ = gLo[(ax * NV + (k - 1)) * 64], h_ = gHi[(ax * NV + (k - 1)) * 64];
                        bad = bad || !(l_ <= h_);
                        if (l_ == h_) eq_ |= 1ull << k;
                    }
                }
                if (ax == 0) dsc[0] = eq_; else if (ax == 1) dsc[1] = eq_; else dsc[2] = eq_;
            }
            solve_any = !bad;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dsc[ax] = solve_any ? (dsc[ax] | 1ull) : 0ull;
            a.status[b] = solve_any ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
            if (a.iters) a.iters[b] = 0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) a.desc[3LL * b + ax] = dsc[ax];
        } else if (have) {
            // nothing to solve: an invalid trajectory (left untouched), or a single segment (its polynomial follows from the boundary data)
            const bool valid1 = shape_ok && t_ok && M == 1;
            a.status[b] = (shape_ok && t_ok && M - 1 <= NV) ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
            if (a.iters) a.iters[b] = 0;
#pragma unroll 1
            for (int ax = 0; ax < 3; ++ax) {
                a.desc[3LL * b + ax] = 0ull;
                if (valid1 && a.active) { a.active[2 * (3LL * b + ax)] = 0ull; a.active[2 * (3LL * b + ax) + 1] = 0ull; }
                if (valid1) {
                    constexpr int NC = 2 * R;
                    const long long base3 = 3LL * ((long long)s0 + b) + ax;
                    const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                    double ys[ND], ye[ND], c1[NC];
#pragma unroll
                    for (int d = 0; d < ND; ++d) { ys[d] = bc[d * 3]; ye[d] = bc[(ND + d) * 3]; }
                    const double Tk = TT[0];
                    segment_coeffs_det<R>(a.waypoints[base3], ys, a.waypoints[base3 + 3], ye, Tk, fast_rcp(Tk), c1);
                    if (!((fabs(c1[NC - 1]) < INFINITY) && (fabs(c1[R]) < INFINITY))) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                    double* o = a.coeff + ((size_t)3 * s0 + ax) * NC;
#pragma unroll
                    for (int j = 0; j < NC; ++j) o[j] = c1[j];
                }
            }
        }
        if (__ballot(solve_any) == 0ull) continue;
        if (!solve_any) { M = 2; s0 = 0; }            // (keeps every index in range; nothing is written for this lane)
        const int n = M - 1;
        // durations through this lane's LDS column (free until G is written; one 8-byte slot each): the chain reads them with a run-time index
        double* const sTd = reinterpret_cast<double*>(s_T) + lane;
#pragma unroll
        for (int i = 0; i <= NV; ++i) sTd[i * 64] = solve_any ? Tr[i] : 1.0;
        lds_publish();
        auto ldT = [&](int i) -> double { return sTd[(i < M ? i : M - 1) * 64]; };
        int kmax = n;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
        kmax = __builtin_amdgcn_readfirstlane(kmax);
        LN_T(0);

        // ================= forward: the block LDL' chain; its records (S_k^-1, E_k) through the scratch (225 / 390 doubles per lane: more than
        // the idle tableau column holds; [field][lane]: coalesced, and the per-wave scratch is re-used chunk after chunk: cache-resident) =======
        constexpr int RS = lane_rec_doubles(R);
        double* const rec0 = gw + (NP + NV + 9 * NV) * 64 + lane;
        auto REC = [&](int k, int f) -> double& { return rec0[(size_t)((k - 1) * RS + f) * 64]; };
        {
            FullBlocks<R> sa;
            sa.build(ldT(0));
            Inv lprev;
            LDLPack<R>::zero(lprev);
#pragma unroll 1
            for (int k = 1; k <= kmax; ++k) {
                const bool vk = k <= n;
                FullBlocks<R> sb;
                sb.build(ldT(min(k, M - 1)));
                const double cpl = (vk && k >= 2) ? 1.0 : 0.0;
                double D[R][R], Mp[R][R], Yp[R][R], Zp[R][R];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const double dv = sa.B11[i][c] + sb.B00(i, c);
                        D[i][c] = vk ? dv : (i == c ? 1.0 : 0.0);
                        Mp[i][c] = sa.B01[i][c] * cpl;
                    }
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    double col[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) col[i] = Mp[i][c];
                    lprev.forward(col);
#pragma unroll
                    for (int i = 0; i < R; ++i) { Yp[i][c] = col[i]; Zp[i][c] = col[i] * lprev.dinv[i]; }
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q)
#pragma unroll
                        for (int c = 0; c <= i; ++c) D[i][c] -= Yp[q][i] * Zp[q][c];
                if (k >= 2) {      // E_{k-1} = S_{k-1}^-1 X_{k-1}
                    double E[R][R];
#pragma unroll
                    for (int c = 0; c < R; ++c) {
#pragma unroll
                        for (int i = R - 1; i >= 0; --i) {
                            double v = Zp[i][c];
#pragma unroll
                            for (int q = i + 1; q < R; ++q) v -= lprev.l[q][i] * E[q][c];
                            E[i][c] = v;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) REC(k - 1, NE + i * R + c) = E[i][c];
                }
                Inv ldl;
                ldl.factor(D);
                {
                    double Si[R][R];
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        double col[R];
#pragma unroll
                        for (int i = 0; i < R; ++i) col[i] = (i == c) ? 1.0 : 0.0;
                        ldl.solve(col);
#pragma unroll
                        for (int i = 0; i < R; ++i) Si[i][c] = col[i];
                    }
                    int f = 0;
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c <= i; ++c) REC(k, f++) = Si[i][c];
                }
                lprev = ldl;
                sa = sb;
            }
#pragma unroll
            for (int i = 0; i < R * R; ++i) REC(kmax, NE + i) = 0.0;      // E of the last chain knot: nothing behind it
        }
        lds_publish();
        LN_T(1);

        // ================= backward: diagonal blocks of H^-1, the last block column, G and the unconstrained minimisers =================
        // cvj[j] = Z_{k, j+1} e_0 while k descends (the entry of G is its component 0), wvj[j] = e_0' Z_{j+1, n}
        {
            double Zk1[R][R], Zkn[R][R], cvj[NV][R];
            // boundary data of the three axes as right-hand sides at knot 1 (r1) and knot n (rn): y0_j = cvj_j(at k = 1)' r1 + (e_0' Z_{j+1,n}) rn;
            // the second part is known when k passes j + 1 and goes to the scratch at once (no 45 doubles of registers for it)
            double r1[3][R], rn[3][R];
            {
                FullBlocks<R> seg0, segl;
                seg0.build(ldT(0));
                segl.build(ldT(M - 1));
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const long long base3 = 3LL * ((long long)s0 + b) + ax;
                    const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                    double x0[R], xM[R];
                    x0[0] = solve_any ? a.waypoints[base3] : 0.0;
                    xM[0] = solve_any ? a.waypoints[base3 + 3LL * M] : 0.0;
#pragma unroll
                    for (int d = 0; d < ND; ++d) { x0[d + 1] = solve_any ? bc[d * 3] : 0.0; xM[d + 1] = solve_any ? bc[(ND + d) * 3] : 0.0; }
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        double v1 = 0.0, vn = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) { v1 -= seg0.B01[c][i] * x0[c]; vn -= segl.B01[i][c] * xM[c]; }
                        r1[ax][i] = v1;
                        rn[ax][i] = vn;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int c = 0; c < R; ++c) { Zk1[i][c] = 0.0; Zkn[i][c] = 0.0; }
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < R; ++i) cvj[j][i] = 0.0;
            // G is kept in the LDS tableau while it is built (entries below the diagonal) and copied to the scratch afterwards, so that the
            // chain records in the scratch are never overwritten before they are read
            double gdiag[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) gdiag[j] = 1.0;
            double rnx[RS];      // the record of the knot the next step takes: requested one step ahead (a global round trip per knot otherwise)
#pragma unroll
            for (int f = 0; f < RS; ++f) rnx[f] = REC(kmax, f);
#pragma unroll 1
            for (int k = kmax; k >= 1; --k) {
                double Si[R][R], E[R][R];
                {
                    int f = 0;
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c <= i; ++c) { Si[i][c] = rnx[f]; Si[c][i] = Si[i][c]; ++f; }
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) E[i][c] = rnx[NE + i * R + c];
                    const int kn = k >= 2 ? k - 1 : 1;
#pragma unroll
                    for (int f2 = 0; f2 < RS; ++f2) rnx[f2] = REC(kn, f2);
                }
                double P[R][R], Zkk[R][R];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q < R; ++q) v += E[i][q] * Zk1[q][c];
                        P[i][c] = v;
                    }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c) {
                        double v = Si[i][c];
#pragma unroll
                        for (int q = 0; q < R; ++q) v += P[i][q] * E[c][q];
                        Zkk[i][c] = v;
                        Zkk[c][i] = v;
                    }
                const double dn = (k == n) ? 1.0 : 0.0;
                {
                    double Zn[R][R];
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            double v = dn * Zkk[i][c];
#pragma unroll
                            for (int q = 0; q < R; ++q) v -= E[i][q] * Zkn[q][c];
                            Zn[i][c] = v;
                        }
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) Zkn[i][c] = Zn[i][c];
                }
                // columns j + 1 >= k (variable index j >= k - 1); the entry (j, k - 1) of G is component 0 of the new vector
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int kj = j + 1;
                    const double dj = (k == kj) ? 1.0 : 0.0;
                    double nv[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        double v = dj * Zkk[i][0];
#pragma unroll
                        for (int q = 0; q < R; ++q) v -= E[i][q] * cvj[j][q];
                        nv[i] = v;
                    }
                    const bool on = k <= kj && kj <= n;
#pragma unroll
                    for (int i = 0; i < R; ++i) cvj[j][i] = nv[i];
                    if (k == kj) {      // e_0' Z_{j+1,n} is row 0 of the last block column as it stands now
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            double v = 0.0;
#pragma unroll
                            for (int c = 0; c < R; ++c) v += Zkn[0][c] * rn[ax][c];
                            gY[(ax * NV + j) * 64] = v;
                        }
                    }
                    if (on) {
                        if (k == kj) gdiag[j] = nv[0];
                        else { const int e = j * (j - 1) / 2 + (k - 1); sE[(e >> 1) * 128 + (e & 1)] = (float)nv[0]; }
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c < R; ++c) Zk1[i][c] = Zkk[i][c];
            }
            // rows and columns beyond n: neutral in the sweeps
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (j >= n) {
#pragma unroll
                    for (int c = 0; c < NV; ++c)
                        if (c < j) { const int e = j * (j - 1) / 2 + c; sE[(e >> 1) * 128 + (e & 1)] = 0.0f; }
                }
            if ((NT & 1) != 0) sE[(NT >> 1) * 128 + 1] = 0.0f;     // the unused half of the last pair
            lds_publish();
            // ---- G -> scratch (the axes start from copies of it); unconstrained minimisers of the three axes
#pragma unroll
            for (int p = 0; p < NP; ++p) gT[p * 64] = sT[p * 64];
#pragma unroll
            for (int j = 0; j < NV; ++j) gD[j * 64] = gdiag[j];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    double v = (j + 1 <= kmax) ? gY[(ax * NV + j) * 64] : 0.0;       // (columns beyond the wave's longest chain were never visited)
#pragma unroll
                    for (int c = 0; c < R; ++c) v += cvj[j][c] * r1[ax][c];
                    gY[(ax * NV + j) * 64] = (solve_any && j < n) ? v : 0.0;
                }
        }
        lds_publish();
        LN_T(2);

        // ================= the three axes of this lane's trajectory, one after the other; every lane at its own pace =================
        double y[NV], lo[NV], hi[NV], dg[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { y[j] = 0.0; lo[j] = 0.0; hi[j] = 0.0; dg[j] = 1.0; }
        unsigned inW = 0u, neg = 0u, eqm = 0u;            // bit j: in the working set / at its upper bound / an equality row
        const unsigned validm = solve_any ? ((1u << n) - 1u) : 0u;
        int axis = -1, q = -1, trips = 0;
        double sdir = 0.0, muq = 0.0, tolA = 1e-12;
        bool running = false, finished = !solve_any;
        const int max_trips = 4 * n + 16 + max_trips_extra;
        for (;;) {
            // ---- a lane without an axis takes its next one: tableau <- G, positions <- the unconstrained minimiser, boxes of the axis
            const bool want = !finished && !running;
            const unsigned long long wantm = __ballot(want), runm = __ballot(running);
            if (wantm != 0ull && (runm == 0ull || __popcll(wantm) >= 16)) {
                if (want) {
                    ++axis;
#pragma unroll
                    for (int s_ = 0; s_ < 2; ++s_)
                        if (axis < 3 && !((axis == 0 ? dsc[0] : (axis == 1 ? dsc[1] : dsc[2])) & 1ull)) ++axis;
                    if (axis >= 3) finished = true;
                }
                const bool init = want && !finished;
                if (init) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) sT[p * 64] = gT[p * 64];
                    const unsigned long long dcur = axis == 0 ? dsc[0] : (axis == 1 ? dsc[1] : dsc[2]);
                    eqm = (unsigned)((dcur >> 1) & 0x7FFFull) & validm;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        dg[j] = gD[j * 64];
                        y[j] = gY[(axis * NV + j) * 64];
                        lo[j] = gLo[(axis * NV + j) * 64];
                        hi[j] = gHi[(axis * NV + j) * 64];
                    }
                    tolA = 1e-12 * (1.0 + raw_min(fabs(lo[0]), fabs(hi[0])));      // (one tolerance per axis: the selection is a heuristic)
                    inW = 0u; neg = 0u; q = -1; trips = 0; running = true;
                }
                lds_publish();
                LN_T(3);
            }
            if (__ballot(running) == 0ull) {
                if (__ballot(!finished) == 0ull) break;
                continue;
            }
            const bool go = running;
            // ---- entering constraint: steepest dual ascent, violation^2 / T_qq (an equality row first)
            bool done = false;
            if (go && q < 0) {
                // (keys in single precision -- the choice among the violated constraints is a heuristic -- and a tree, not a chain, of compares)
                float key[16];
                int code[16];
                const unsigned candm = validm & ~inW;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const double below = lo[j] - y[j], above = y[j] - hi[j];
                    const double v = raw_max(below, above);
                    const bool cand = ((candm >> j) & 1u) && v > tolA && dg[j] > 0.0;
                    const float vf = (float)v;
                    const float kv = ((eqm >> j) & 1u) ? 3.0e38f : fminf(vf * vf * __builtin_amdgcn_rcpf((float)dg[j]), 1.0e38f);
                    key[j] = cand ? kv : -1.0f;
                    code[j] = j | (below > above ? 16 : 0);
                }
                key[15] = -1.0f; code[15] = 0;
#pragma unroll
                for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                    for (int j = 0; j < w; ++j) {
                        const bool tk_ = key[j + w] > key[j];
                        key[j] = tk_ ? key[j + w] : key[j];
                        code[j] = tk_ ? code[j + w] : code[j];
                    }
                if (key[0] > 0.0f && trips < max_trips) { q = code[0] & 15; sdir = (code[0] & 16) ? 1.0 : -1.0; muq = 0.0; }
                else done = true;
            }
            LN_T(4);
            if (go && !done) {
                const int qq = q;
                // ---- direction: column q of the tableau (this lane's LDS column: nothing to hand over), full step length
                float u[NV];
                double dq = 1.0, yq = 0.0, bq_ = 0.0;
                const int q2 = qq * (qq - 1) / 2;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int e = i > qq ? i * (i - 1) / 2 + qq : q2 + i;       // (i == qq: a valid address, the value is replaced below)
                    const int ec = i == qq ? 0 : e;
                    u[i] = sE[(ec >> 1) * 128 + (ec & 1)];
                    dq = i == qq ? dg[i] : dq;
                    yq = i == qq ? y[i] : yq;
                    bq_ = i == qq ? (sdir > 0.0 ? lo[i] : hi[i]) : bq_;
                }
                const double pvq = rcp1(dq);
                const double t1 = (bq_ - yq) * sdir * pvq;
                // ---- first multiplier of the working set to reach zero along d = sdir * column q
                double rbest = 1e300;
                int kb = -1;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const double dj = sdir * (j == qq ? dq : (double)u[j]);
                    const bool inw = (inW & ~eqm) >> j & 1u;
                    const double swj = ((neg >> j) & 1u) ? -1.0 : 1.0;
                    const bool blocks = inw && swj * dj > 0.0;
                    const double ratio = raw_min(raw_max(-y[j] * rcp1(dj), 0.0), 1e299);
                    if (blocks && ratio < rbest) { rbest = ratio; kb = j; }
                }
                const bool partial = rbest < t1;
                const double t = partial ? rbest : t1;
                const int kp = partial ? kb : qq;
#pragma unroll
                for (int j = 0; j < NV; ++j) y[j] = fma(t, sdir * (j == qq ? dq : (double)u[j]), y[j]);
                muq = fma(sdir, t, muq);
                LN_T(5);
                // ---- sweep on the pivot kp: the constraint q enters (full step) or the blocking one leaves (partial step)
                double tk = dq;
                if (partial) {
                    const int k2 = kp * (kp - 1) / 2;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int e = i > kp ? i * (i - 1) / 2 + kp : k2 + i;
                        const int ec = i == kp ? 0 : e;
                        u[i] = sE[(ec >> 1) * 128 + (ec & 1)];
                        tk = i == kp ? dg[i] : tk;
                    }
                }
                const double piv = partial ? rcp1(tk) : pvq;
                const double muq_now = muq;
                float tc[NV], sc[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const bool pc = j == kp;
                    const double tcd = pc ? tk - (partial ? -1.0 : 1.0) : (double)u[j];     // (entering: T_qq > 0; leaving: -[G_WW^-1]_kk < 0)
                    const double scd = tcd * piv;
                    tc[j] = (float)tcd;
                    sc[j] = (float)scd;
                    const double dnew = fma(-tcd, scd, dg[j]);
                    dg[j] = pc ? -piv : dnew;
                    const double yb = ((neg >> j) & 1u) ? hi[j] : lo[j];
                    y[j] = pc ? (partial ? yb : -muq_now) : y[j];
                }
                {
                    const unsigned bit = 1u << kp;
                    inW = partial ? (inW & ~bit) : (inW | bit);
                    neg = partial ? neg : (sdir < 0.0 ? (neg | bit) : (neg & ~bit));
                }
                constexpr int NPA = NP / 2;
                lane_sweep_batch<0>(sT, tc, sc, std::make_integer_sequence<int, NPA>{});
                lane_sweep_batch<NPA>(sT, tc, sc, std::make_integer_sequence<int, NP - NPA>{});
                if (!partial) q = -1;
                ++trips;
            }
            if (go && done) {
                // ---- hand the working set of this axis over (bit k = interior knot k, as the solve kernel reads it)
                const unsigned act = inW & ~eqm & validm, upm = act & neg;
                a.guess[2 * (3LL * b + axis)] = (unsigned long long)act << 1;
                a.guess[2 * (3LL * b + axis) + 1] = (unsigned long long)upm << 1;
                running = false;
            }
            LN_T(6);
#ifdef UAVQP_LANE_TIMING
            ln_acc[7] += 1 << 20;      // trips of the wave in the high bits
#endif
        }
    }
    LN_T_FLUSH;
}

}  // namespace uavqp
