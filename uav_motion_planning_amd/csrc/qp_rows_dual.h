// qp_rows_dual.h -- the starting working set of the general-rows solve (qp_rows2.h) from the dual active-set method of
// qp_corridor_dual.h, extended from knot boxes to rows lo <= p_i^(d)(tau T_i) <= hi (round 4).
//
// A row at time tau T_i of derivative order d < R is a bound on COMPONENT d of the Hermite state of a knot inserted at tau T_i:
// inserting a free knot does not change the optimum (the optimal polynomial of the segment, cut in two, is feasible for the refined
// problem and costs the same), so the QP with rows is the knot-box QP of qp_corridor_dual.h on a refined time grid whose constraints
// are (knot, component) pairs: the boxes (original interior knots, component 0) and the rows (inserted knots, component d).  The dense
// inverse Hessian restricted to those pairs, G_ij = e_ai' [H^-1]_{ki,kj} e_aj, comes from the same backward recursions (columns started
// with Z_kk e_a instead of Z_kk e_0), the dual method on its swept tableau is unchanged.  BASELINE config 3 with K = 2 rows per segment
// (position sample and velocity limit at mid-segment): 31 interior knots, 47 constraints per axis, 7.0 active at the solution, 8.6
// exchanges mean (CPU replay) -- against 25 block solves of size r + K over all knots for the dual method of qp_rows2.h from the box set.
// As in qp_corridor_dual.h nothing here decides a result: the set goes to rows_pair_kernel as its starting working set, which verifies it
// with its own exact solve and goes on from there if it has to.
//
// One group of 32 lanes per trajectory (two per wave), lane l owns tableau columns l and l + 32 (48 rows); lane l also prepares segment l
// (its inserted knots, their rows).  Handled: rows with 0 < tau < 1, two rows of a segment not identical in (tau, d), at most 48
// constraints and 47 refined interior knots (16 segments with K = 2, 24 with K = 1); anything else is left to the box phase
// (need_phase1) and starts the rows solve from the box set as before.
#pragma once
#include "qp_corridor_dual.h"
#include "qp_rows2.h"

namespace uavqp {

struct RowsDualArgs {
    RowsArgs r;
    const int32_t* order;            // dealing order of the trajectories (may be null)
    unsigned long long* warm_box;    // [problem][2]: (active, upper) of the knot boxes, bit k = interior knot k -- written for handled trajectories
    unsigned long long* warm_rows;   // [problem][2 K]: (active, upper) per row slot, bit s = segment s -- zeroed by the host, written for handled ones
    unsigned char* need_phase1;      // [n_traj]: 1 = not handled here
#ifdef UAVQP_DUAL_DEBUG
    double* dbg;
#endif
};

__device__ __forceinline__ double group_max32(double v) {
    v = __builtin_fmax(v, dpp_f64<0xB1>(v));
    v = __builtin_fmax(v, dpp_f64<0x4E>(v));
    v = __builtin_fmax(v, dpp_f64<0x141>(v));
    v = __builtin_fmax(v, dpp_f64<0x140>(v));
    return __builtin_fmax(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ double group_min32(double v) {
    v = __builtin_fmin(v, dpp_f64<0xB1>(v));
    v = __builtin_fmin(v, dpp_f64<0x4E>(v));
    v = __builtin_fmin(v, dpp_f64<0x141>(v));
    v = __builtin_fmin(v, dpp_f64<0x140>(v));
    return __builtin_fmin(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ double pack_code7(double v, int code) {
    return __longlong_as_double((__double_as_longlong(v) & ~127ll) | (long long)code);
}
__device__ __forceinline__ int code7_of(double v) { return (int)(__double_as_longlong(v) & 127ll); }

constexpr int rows_dual_lds_doubles() { return 48 * 48 + 2 * 50 + 8 + 8 + 56; }   // G / chain records, column buffer (+ durations before the dual phase), scalars, masks, int tables

template <int R, int K>
__global__ __launch_bounds__(64, 1) void rows_dual_kernel(RowsDualArgs aa, int max_trips_extra) {
    const RowsArgs& a = aa.r;
    constexpr int ND = R - 1, L = 32, NG = 2, NRW = 48, NE = R * (R + 1) / 2, RS = 48, CBS = 50;
    constexpr int O_CB = NRW * RS, O_SC = O_CB + 2 * CBS, O_MK = O_SC + 8, O_IT = O_MK + 8, GRP = rows_dual_lds_doubles();
    static_assert(NE + R * R <= RS, "a chain record fits a slot");
    static_assert(GRP % 2 == 0 && O_CB % 2 == 0 && O_IT % 2 == 0, "16-byte aligned rows");
    __shared__ __attribute__((aligned(16))) double s_all[NG * GRP];
    using Inv = SmallLDL<R>;
    const int lane = threadIdx.x, l = lane & 31, grp = lane >> 5;
    double* const sg = s_all + grp * GRP;
    double* const CB = sg + O_CB;      // [2][50]: the two columns of the pivot's owner; element 48 is a constant zero.  Before the dual phase: durations of the refined segments
    double* const SC = sg + O_SC;
    unsigned long long* const MK = reinterpret_cast<unsigned long long*>(sg + O_MK);   // working-set masks of the axis being handed over
    int* const KT = reinterpret_cast<int*>(sg + O_IT);       // [50] per refined knot: first constraint | count << 8 | comp0 << 12 | comp1 << 16
    int* const CD = KT + 50;                                   // [48] per constraint: knot | comp << 8 | kind << 12 (0 box, 1 + slot) | segment << 16
    int* const CNT = reinterpret_cast<int*>(CB);              // [3][32] per segment: knots, constraints, refined segments (before the durations are written)
    auto ES = [&](int k) -> double* { return sg + (k - 1) * RS; };
    auto GR = [&](int i) -> double* { return sg + i * RS; };
    const int cidx[2] = {l, l + L};
    const int crd[2] = {cidx[0] < NRW ? cidx[0] : NRW, cidx[1] < NRW ? cidx[1] : NRW};
    const int crow[2] = {min(cidx[0], NRW - 1), min(cidx[1], NRW - 1)};
    const unsigned long long gmask = 0xFFFFFFFFull << (32 * grp);

    const long long n_batches = ((long long)a.n_traj + NG - 1) / NG;
    for (long long bt = blockIdx.x; bt < n_batches; bt += gridDim.x) {
        const long long bq = bt * NG + grp;
        const bool have = bq < a.n_traj;
        const int b = have ? (aa.order ? aa.order[bq] : (int)bq) : 0;
        int s0 = 0, M = 0;
        if (have) { if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; } }
        const bool shape_ok = have && M >= 2 && M <= 32 && (a.uniform > 0 || M <= a.max_segments);
        lds_publish();
        // ---------------- lane l prepares segment l: its inserted knots, their rows, the durations of its pieces ----------------
        const bool myseg = shape_ok && l < M;
        bool segok = true;
        int nins = 0, nrow = 0, slotA = 0, slotB = 1, dA = 0, dB = 0;
        double Tseg = 1.0, cutA = 0.5, cutB = 0.5;
        if (myseg) {
            Tseg = a.times[s0 + l];
            segok = Tseg > 0.0 && Tseg < INFINITY;
            int d[K];
            double tau[K];
            bool used[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                d[j] = a.row_deriv[(size_t)(s0 + l) * K + j];
                tau[j] = a.row_tau[(size_t)(s0 + l) * K + j];
                used[j] = d[j] >= 0;
                if (used[j]) segok = segok && d[j] < R && tau[j] > 0.0 && tau[j] < 1.0;
            }
            if (K == 1) {
                if (used[0]) { nins = 1; nrow = 1; slotA = 0; dA = d[0]; cutA = tau[0]; }
            } else {
                if (used[0] && used[K - 1]) {
                    nrow = 2;
                    if (tau[0] == tau[K - 1]) {
                        segok = segok && d[0] != d[K - 1];
                        nins = 1; slotA = 0; slotB = 1; dA = d[0]; dB = d[K - 1]; cutA = tau[0];
                    } else {
                        nins = 2;
                        const bool sw_ = tau[K - 1] < tau[0];
                        slotA = sw_ ? 1 : 0; slotB = sw_ ? 0 : 1;
                        dA = sw_ ? d[K - 1] : d[0]; dB = sw_ ? d[0] : d[K - 1];
                        cutA = sw_ ? tau[K - 1] : tau[0]; cutB = sw_ ? tau[0] : tau[K - 1];
                    }
                } else if (used[0]) { nins = 1; nrow = 1; slotA = 0; dA = d[0]; cutA = tau[0]; }
                else if (used[K - 1]) { nins = 1; nrow = 1; slotA = 1; dA = d[K - 1]; cutA = tau[K - 1]; }
            }
        }
        const int hasbox = (myseg && l < M - 1) ? 1 : 0;     // the original knot that closes this segment is an interior knot
        CNT[l] = myseg ? nins + hasbox : 0;
        CNT[32 + l] = myseg ? nrow + hasbox : 0;
        CNT[64 + l] = myseg ? nins + 1 : 0;
        lds_publish();
        int ko = 1, co = 0, so = 0, nref = 0, NC = 0, Mr = 0;    // this segment's first knot / constraint / refined segment; totals
        for (int i = 0; i < 32; ++i) {
            const int a0 = CNT[i], a1 = CNT[32 + i], a2 = CNT[64 + i];
            if (i < l) { ko += a0; co += a1; so += a2; }
            nref += a0; NC += a1; Mr += a2;
        }
        const bool allok = (__ballot(segok || !myseg) & gmask) == gmask;
        const bool handled = shape_ok && allok && NC <= NRW && nref <= NRW - 1 && nref >= 1;
        lds_publish();
        if (handled && myseg) {
            // durations of the pieces
            if (nins == 0) CB[so] = Tseg;
            else if (nins == 1) { CB[so] = cutA * Tseg; CB[so + 1] = (1.0 - cutA) * Tseg; }
            else { CB[so] = cutA * Tseg; CB[so + 1] = (cutB - cutA) * Tseg; CB[so + 2] = (1.0 - cutB) * Tseg; }
            // knots and constraints
            const int kindA = 1 + slotA, kindB = 1 + slotB;
            if (nins == 1 && nrow == 1) {
                KT[ko] = co | (1 << 8) | (dA << 12);
                CD[co] = ko | (dA << 8) | (kindA << 12) | (l << 16);
            } else if (nins == 1 && nrow == 2) {
                KT[ko] = co | (2 << 8) | (dA << 12) | (dB << 16);
                CD[co] = ko | (dA << 8) | (kindA << 12) | (l << 16);
                CD[co + 1] = ko | (dB << 8) | (kindB << 12) | (l << 16);
            } else if (nins == 2) {
                KT[ko] = co | (1 << 8) | (dA << 12);
                KT[ko + 1] = (co + 1) | (1 << 8) | (dB << 12);
                CD[co] = ko | (dA << 8) | (kindA << 12) | (l << 16);
                CD[co + 1] = (ko + 1) | (dB << 8) | (kindB << 12) | (l << 16);
            }
            if (hasbox) {
                KT[ko + nins] = (co + nrow) | (1 << 8);
                CD[co + nrow] = (ko + nins) | (0 << 8) | (0 << 12) | (l << 16);
            }
        }
        if (have && l == 0) aa.need_phase1[b] = handled ? 0 : 1;
        if ((__ballot(handled) & ~0ull) == 0ull) continue;
        const int n = handled ? nref : 1;                      // refined interior knots
        const int Mref = handled ? Mr : 2;
        if (!handled && l < 2) CB[l] = 1.0;
        lds_publish();
        int kmax = n;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
        kmax = __builtin_amdgcn_readfirstlane(kmax);
        auto ldT = [&](int i) -> double { return CB[i]; };

        // ---------------- forward: block LDL' chain over the refined knots, replicated in the lanes of the group ----------------
        FullBlocks<R> sa;
        sa.build(ldT(0));
        FullBlocks<R> seg0 = sa, segl;
        segl.build(ldT(Mref - 1));
        Inv lprev;
        LDLPack<R>::zero(lprev);
#pragma unroll 1
        for (int k = 1; k <= kmax; ++k) {
            const bool vk = k <= n;
            FullBlocks<R> sb;
            sb.build(ldT(min(k, Mref - 1)));
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
            if (k >= 2) {
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
                if (l == 0) {
                    double* const rec = ES(k - 1);
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) rec[NE + i * R + c] = E[i][c];
                }
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
                if (l == 0) {
                    double* const rec = ES(k);
                    int f = 0;
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c <= i; ++c) rec[f++] = Si[i][c];
                }
            }
            lprev = ldl;
            sa = sb;
        }
        if (l == 0) {
            double* const rec = ES(kmax);
#pragma unroll
            for (int i = 0; i < R * R; ++i) rec[NE + i] = 0.0;
        }
        lds_publish();

        // ---------------- backward: diagonal blocks of H^-1, the last block column, the owned columns of G ----------------
        int kc[2], ac[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const bool vc = handled && cidx[sl] < NC;
            const int cd = vc ? CD[crow[sl]] : 0;
            kc[sl] = vc ? (cd & 255) : 0;          // 0: no such knot -- the column never starts
            ac[sl] = (cd >> 8) & 15;
        }
        double Zk1[R][R], Zkn[R][R], cv[2][R], wv[2][R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            cv[0][i] = 0.0; cv[1][i] = 0.0; wv[0][i] = 0.0; wv[1][i] = 0.0;
#pragma unroll
            for (int c = 0; c < R; ++c) { Zk1[i][c] = 0.0; Zkn[i][c] = 0.0; }
        }
#pragma unroll 1
        for (int k = kmax; k >= 1; --k) {
            double Si[R][R], E[R][R];
            {
                const double* const rec = ES(k);
                int f = 0;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c) { Si[i][c] = rec[f]; Si[c][i] = Si[i][c]; ++f; }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c < R; ++c) E[i][c] = rec[NE + i * R + c];
            }
            const int kt = (handled && k <= n) ? KT[k] : 0;     // constraints that sit at this knot
            lds_publish();
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
            const int cf = kt & 255, cnt = (kt >> 8) & 15;
            const int cp[2] = {(kt >> 12) & 15, (kt >> 16) & 15};
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                // column (kc, ac): starts at its own knot with Z_kk e_ac, then c <- -E_k c; w = e_ac' Z_{kc, n} is picked up at the same knot
                double dja[R];
#pragma unroll
                for (int q = 0; q < R; ++q) dja[q] = (k == kc[sl] && q == ac[sl]) ? 1.0 : 0.0;
                double nv[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) v += dja[q] * Zkk[i][q] - E[i][q] * cv[sl][q];
                    nv[i] = v;
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    cv[sl][i] = nv[i];
                    double w = wv[sl][i];
#pragma unroll
                    for (int q = 0; q < R; ++q) w = fma(dja[q], Zkn[q][i], w);
                    wv[sl][i] = w;
                }
                if (k <= kc[sl]) {     // entries (i, c) and (c, i) of every constraint i that sits at knot k
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if (t < cnt) {
                            double val = nv[0];
#pragma unroll
                            for (int q = 1; q < R; ++q) val = (cp[t] == q) ? nv[q] : val;
                            GR(cf + t)[cidx[sl]] = val;
                            GR(cidx[sl])[cf + t] = val;
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int c = 0; c < R; ++c) Zk1[i][c] = Zkk[i][c];
        }
        // rows and columns beyond the constraints hold what the chain records left there: they must be neutral in the sweeps
        const int ncv = handled ? NC : 0;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            if (cidx[sl] < NRW) {
                double* const row = GR(cidx[sl]);
#pragma unroll 1
                for (int i = cidx[sl] >= ncv ? 0 : ncv; i < NRW; ++i) row[i] = 0.0;
            }
        lds_publish();
        int nrows_w = ncv;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) nrows_w = max(nrows_w, __shfl_xor(nrows_w, o, 64));
        const int nrows = __builtin_amdgcn_readfirstlane(min(NRW, (nrows_w + 1) & ~1));

        // ---------------- per axis: unconstrained values and bounds of the owned columns ----------------
        double y0[3][2], lo3[3][2], hi3[3][2];
        {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const long long base3 = 3LL * ((long long)s0 + b) + ax;
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                double x0[R], xM[R], r1[R], rn[R];
                x0[0] = handled ? a.waypoints[base3] : 0.0;
                xM[0] = handled ? a.waypoints[base3 + 3LL * M] : 0.0;
#pragma unroll
                for (int d = 0; d < ND; ++d) { x0[d + 1] = handled ? bc[d * 3] : 0.0; xM[d + 1] = handled ? bc[(ND + d) * 3] : 0.0; }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    double v1 = 0.0, vn = 0.0;
#pragma unroll
                    for (int c = 0; c < R; ++c) { v1 -= seg0.B01[c][i] * x0[c]; vn -= segl.B01[i][c] * xM[c]; }
                    r1[i] = v1;
                    rn[i] = vn;
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const bool vc = handled && cidx[sl] < NC;
                    const int cd = vc ? CD[crow[sl]] : 0;
                    const int kind = (cd >> 12) & 15, seg = (cd >> 16) & 255;
                    double lo_ = 0.0, hi_ = 0.0;
                    if (vc) {
                        if (kind == 0) {
                            const long long at = base3 + 3LL * (seg + 1);
                            lo_ = a.corr_lo ? a.corr_lo[at] : a.waypoints[at];
                            hi_ = a.corr_hi ? a.corr_hi[at] : a.waypoints[at];
                        } else {
                            const size_t at = ((size_t)(s0 + seg) * K + (kind - 1)) * 3 + ax;
                            lo_ = a.row_lo[at];
                            hi_ = a.row_hi[at];
                        }
                    }
                    lo3[ax][sl] = lo_;
                    hi3[ax][sl] = hi_;
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < R; ++c) v += cv[sl][c] * r1[c] + wv[sl][c] * rn[c];
                    y0[ax][sl] = vc ? v : 0.0;
                }
            }
        }
        lds_publish();
        if (l == 0) { CB[NRW] = 0.0; CB[NRW + 1] = 0.0; CB[CBS + NRW] = 0.0; CB[CBS + NRW + 1] = 0.0; }
        lds_publish();
#ifdef UAVQP_DUAL_DEBUG
        // per trajectory (dealing position bq < 16): [0, 2304) G row-major [48][48]; [2304 + 192 ax + 48 what + col]: what 0 = y0, 1 = trips, 2 = y at the end; [2304 + 576 + col] = constraint descriptors
        double* const dbg = (aa.dbg && handled && bq < 16) ? aa.dbg + bq * 4096 : nullptr;
        if (dbg) {
            for (int sl = 0; sl < 2; ++sl)
                if (cidx[sl] < NC) {
                    for (int i = 0; i < NC; ++i) dbg[i * 48 + cidx[sl]] = GR(cidx[sl])[i];
                    dbg[2304 + 576 + cidx[sl]] = (double)CD[cidx[sl]];
                }
            if (l == 0) { dbg[2304 + 640] = NC; dbg[2304 + 641] = n; }
        }
#endif

        // ---------------- the three axes, one after the other for the whole wave (as qp_corridor_dual.h) ----------------
        double A[2][NRW];
        const int max_trips = 4 * NC + 16 + max_trips_extra;
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
        int q = -1, trips = 0;
        double sdir = 0.0, muq = 0.0;
        double dg[2], y[2], lo[2], hi[2], tol[2], sw[2] = {0.0, 0.0}, eqb[2];
        bool valid[2], inW[2] = {false, false};
        lds_publish();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            valid[sl] = handled && cidx[sl] < NC;
            lo[sl] = axis == 0 ? lo3[0][sl] : (axis == 1 ? lo3[1][sl] : lo3[2][sl]);
            hi[sl] = axis == 0 ? hi3[0][sl] : (axis == 1 ? hi3[1][sl] : hi3[2][sl]);
            y[sl] = axis == 0 ? y0[0][sl] : (axis == 1 ? y0[1][sl] : y0[2][sl]);
            tol[sl] = 1e-12 * (1.0 + fmin(fabs(lo[sl]), fabs(hi[sl])));
            eqb[sl] = (valid[sl] && lo[sl] == hi[sl]) ? 1e300 : 0.0;
            dg[sl] = valid[sl] ? GR(crow[sl])[crow[sl]] : 1.0;
            const double* const row = GR(crow[sl]);
#pragma unroll
            for (int i = 0; i < NRW; i += 2) {
                const double2 tt = *reinterpret_cast<const double2_a*>(row + i);
                A[sl][i] = tt.x;
                A[sl][i + 1] = tt.y;
            }
        }
#ifdef UAVQP_DUAL_DEBUG
        if (dbg)
            for (int sl = 0; sl < 2; ++sl) if (valid[sl]) dbg[2304 + 192 * axis + cidx[sl]] = y[sl];
#endif
        bool done = !handled;
        for (;;) {
            lds_publish();
            if (__ballot(!done && q < 0) != 0ull) {
                double key = 0.0;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const double below = lo[sl] - y[sl], above = y[sl] - hi[sl];
                    const double v = fmax(below, above);
                    const bool cand = valid[sl] && !inW[sl] && v > tol[sl] && dg[sl] > 0.0;
                    const double kv = fmax(fmin(v * v * __builtin_amdgcn_rcp(dg[sl]), 1e299), eqb[sl]);
                    const double pk = pack_code7(kv, (below > above ? 64 : 0) | cidx[sl]);
                    key = fmax(key, cand ? pk : 0.0);
                }
                key = group_max32(key);
                if (!done && q < 0) {
                    if (key > 1e-300 && trips < max_trips) { const int cd = code7_of(key); q = cd & 63; sdir = (cd & 64) ? 1.0 : -1.0; muq = 0.0; }
                    else done = true;
                }
            }
            if (__ballot(!done) == 0ull) break;
            const bool go = !done;
            const int qq = go ? q : 0;
            const int lq = qq & (L - 1), slq = qq >> 5;
            const bool own_q = go && (l == lq);
            if (own_q) {
#pragma unroll
                for (int i = 0; i < NRW; i += 2) {
                    *reinterpret_cast<double2_a*>(CB + i) = make_double2(A[0][i], A[0][i + 1]);
                    *reinterpret_cast<double2_a*>(CB + CBS + i) = make_double2(A[1][i], A[1][i + 1]);
                }
                const double dq = slq ? dg[1] : dg[0], pq = slq ? y[1] : y[0];
                const double bq_ = sdir > 0.0 ? (slq ? lo[1] : lo[0]) : (slq ? hi[1] : hi[0]);
                const double pv = rcp1(dq);
                CB[slq * CBS + qq] = dq;
                SC[0] = (bq_ - pq) * sdir * pv;
                SC[1] = pv;
            }
            lds_publish();
            double d[2];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) d[sl] = sdir * CB[slq * CBS + crd[sl]];
            const double t1 = SC[0];
            double rmin = 1e300;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const bool blocks = sw[sl] * d[sl] > 0.0;
                const double ratio = fmin(fmax(-y[sl] * rcp1(d[sl]), 0.0), 1e299);
                rmin = fmin(rmin, blocks ? pack_code7(ratio, cidx[sl]) : 1e300);
            }
            rmin = group_min32(rmin);
            const bool partial = go && rmin < t1;
            const double t = go ? (partial ? rmin : t1) : 0.0;
            const int kp = partial ? code7_of(rmin) & 63 : qq;
            const int lk = kp & (L - 1), slk = kp >> 5;
            const bool own_k = go && (l == lk);
            y[0] = fma(t, d[0], y[0]);
            y[1] = fma(t, d[1], y[1]);
            muq = fma(sdir, t, muq);
            if (__ballot(partial) != 0ull) {
                if (own_k && partial) {
#pragma unroll
                    for (int i = 0; i < NRW; i += 2) {
                        *reinterpret_cast<double2_a*>(CB + i) = make_double2(A[0][i], A[0][i + 1]);
                        *reinterpret_cast<double2_a*>(CB + CBS + i) = make_double2(A[1][i], A[1][i + 1]);
                    }
                    SC[1] = rcp1(slk ? dg[1] : dg[0]);
                }
            }
            if (own_k) {
                const double tk = slk ? dg[1] : dg[0];
                CB[slk * CBS + kp] = tk - (partial ? -1.0 : 1.0);
            }
            lds_publish();
            {
                const double piv = go ? SC[1] : 0.0;
                double s[2];
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const double tc = CB[slk * CBS + crd[sl]];
                    const bool pc = own_k && sl == slk;
                    s[sl] = tc * piv;
                    const double dn = fma(-tc, s[sl], dg[sl]);
                    dg[sl] = pc ? -piv : dn;
                    const double yb = sw[sl] < 0.0 ? hi[sl] : lo[sl];
                    y[sl] = pc ? (partial ? yb : -muq) : y[sl];
                    sw[sl] = pc ? ((partial || eqb[sl] != 0.0) ? 0.0 : sdir) : sw[sl];
                    inW[sl] = pc ? !partial : inW[sl];
                }
#pragma unroll
                for (int i = 0; i < NRW; i += 2) {
                    if (i < nrows) {
                        const double2 u = *reinterpret_cast<const double2_a*>(CB + slk * CBS + i);
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            A[sl][i] = fma(-u.x, s[sl], A[sl][i]);
                            A[sl][i + 1] = fma(-u.y, s[sl], A[sl][i + 1]);
                        }
                    }
                }
            }
            if (go) {
                if (!partial) q = -1;
                ++trips;
            }
        }
        // ---- hand the working set of this axis over in the rows kernel's layout: boxes by interior knot, rows by slot and segment
        lds_publish();
        if (l < 2 + 2 * K) MK[l] = 0ull;
        lds_publish();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (valid[sl] && inW[sl] && sw[sl] != 0.0) {
                const int cd = CD[crow[sl]];
                const int kind = (cd >> 12) & 15, seg = (cd >> 16) & 255;
                const int word = kind == 0 ? 0 : 2 * kind;
                const unsigned long long bit = 1ull << (kind == 0 ? seg + 1 : seg);
                atomicOr(&MK[word], bit);
                if (sw[sl] < 0.0) atomicOr(&MK[word + 1], bit);
            }
        }
        lds_publish();
#ifdef UAVQP_DUAL_DEBUG
        if (dbg)
            for (int sl = 0; sl < 2; ++sl) if (valid[sl]) { dbg[2304 + 192 * axis + 96 + cidx[sl]] = y[sl]; dbg[2304 + 192 * axis + 48 + cidx[sl]] = (double)trips; }
#endif
#ifdef UAVQP_DUAL_DEBUG
        if (dbg && l == 0) for (int j = 0; j < 2 + 2 * K; ++j) dbg[2304 + 700 + 8 * axis + j] = (double)MK[j];
#endif
        if (handled && l == 0) {
            const size_t prob = 3 * (size_t)b + axis;
            aa.warm_box[2 * prob] = MK[0];
            aa.warm_box[2 * prob + 1] = MK[1];
#pragma unroll
            for (int j = 0; j < 2 * K; ++j) aa.warm_rows[2 * K * prob + j] = MK[2 + j];
        }
        }
    }
}

}  // namespace uavqp
