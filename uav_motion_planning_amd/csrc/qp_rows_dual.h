// qp_rows_dual.h -- the starting working set of the general-rows solve (qp_rows2.h) from the dual active-set method of
// qp_corridor_dual.h, extended from knot boxes to rows lo <= p_i^(d)(tau T_i) <= hi (round 4).
//
// A row is a linear functional of the Hermite states of the two end knots of its segment, c' x = g_l' x_i + g_r' x_{i+1}
// (row_functional, qp_rows.h); a knot box is the functional e_0' x_k.  The dual method only needs the dense matrix
// G_ij = c_i' H^-1 c_j over all constraints and their unconstrained values c_j' H^-1 rhs.  Both come from the chain of
// qp_corridor_dual.h: with z_j = H^-1 c_j, the components of z_j at and above its own knots follow the back-substitution
//     z^(k) = dR Z_kk (g_r - E_{k-1}' g_l) + dL S_k^-1 g_l - E_k z^(k+1)        (dR: k = i + 1, dL: k = i; zero below the start)
// and G_ij = g_l,i' z_j^(k_i) + g_r,i' z_j^(k_i + 1) is taken whenever constraint i does not sit behind constraint j (the other
// half by symmetry).  This is the QP of rows_pair_kernel itself, not a relaxation: the set the dual method ends with is that
// QP's working set.  (The first version refined the time grid by a knot at every row and bounded components of the inserted knots --
// a relaxation, an inserted knot with an active row may break the higher derivatives: 3.65 verifying solves mean instead of 1.)
// BASELINE config 3 with K = 2 rows per segment: 15 knots, 47 constraints per axis, ~7 active at the solution, ~9 exchanges.
// As in qp_corridor_dual.h nothing here decides a result: the set goes to rows_pair_kernel as its starting working set.
//
// One trajectory per wave: lane c owns tableau column c (48 rows) in registers, lane s also prepares segment s (functionals of its
// rows); every branch of the dual loop is wave-uniform.  Handled: at most 48 constraints ((M - 1) + used rows), M <= 32; anything else
// is left to the box phase (need_phase1) and starts the rows solve from the box set as before.
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
    const double* gfun;              // [segment][K][2 R]: g_l, g_r of every row (rows_gfun_kernel, launched before this kernel)
#ifdef UAVQP_DUAL_DEBUG
    double* dbg;
#endif
};

__device__ __forceinline__ double group_max32(double v) {
    v = raw_max(v, dpp_f64<0xB1>(v));
    v = raw_max(v, dpp_f64<0x4E>(v));
    v = raw_max(v, dpp_f64<0x141>(v));
    v = raw_max(v, dpp_f64<0x140>(v));
    return raw_max(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ double group_min32(double v) {
    v = raw_min(v, dpp_f64<0xB1>(v));
    v = raw_min(v, dpp_f64<0x4E>(v));
    v = raw_min(v, dpp_f64<0x141>(v));
    v = raw_min(v, dpp_f64<0x140>(v));
    return raw_min(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ double pack_code7(double v, int code) {
    return __longlong_as_double((__double_as_longlong(v) & ~127ll) | (long long)code);
}
__device__ __forceinline__ int code7_of(double v) { return (int)(__double_as_longlong(v) & 127ll); }

constexpr int rows_dual_lds_doubles(int R) { return 48 * 48 + 48 * 2 * R + 50 + 34 + 8 + 8 + 64; }   // G / chain records, functionals, column buffer, durations, scalars, masks, int tables

__device__ __forceinline__ double wave_max64(double v) {
    v = raw_max(v, dpp_f64<0xB1>(v));
    v = raw_max(v, dpp_f64<0x4E>(v));
    v = raw_max(v, dpp_f64<0x141>(v));
    v = raw_max(v, dpp_f64<0x140>(v));
    v = raw_max(v, __shfl_xor(v, 16, 64));
    return raw_max(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ double wave_min64(double v) {
    v = raw_min(v, dpp_f64<0xB1>(v));
    v = raw_min(v, dpp_f64<0x4E>(v));
    v = raw_min(v, dpp_f64<0x141>(v));
    v = raw_min(v, dpp_f64<0x140>(v));
    v = raw_min(v, __shfl_xor(v, 16, 64));
    return raw_min(v, __shfl_xor(v, 32, 64));
}

template <int R, int K>
__global__ __launch_bounds__(64, 2) void rows_dual_kernel(RowsDualArgs aa, int max_trips_extra) {
    const RowsArgs& a = aa.r;
    constexpr int ND = R - 1, NRW = 48, NE = R * (R + 1) / 2, RS = 48;
    constexpr int O_GF = NRW * RS, O_CB = O_GF + NRW * 2 * R, O_TB = O_CB + 50, O_SC = O_TB + 34, O_MK = O_SC + 8, O_IT = O_MK + 8;
    static_assert(NE + R * R <= RS, "a chain record fits a slot");
    static_assert(O_GF % 2 == 0 && O_CB % 2 == 0 && O_TB % 2 == 0 && O_IT % 2 == 0, "16-byte aligned rows");
    __shared__ __attribute__((aligned(16))) double sg[rows_dual_lds_doubles(R)];
    using Inv = SmallLDL<R>;
    const int lane = threadIdx.x, c = lane;
    double* const GF = sg + O_GF;      // [48][2 R]: g_l, g_r of every constraint (a box: e_0, 0)
    double* const CB = sg + O_CB;      // [50]: the pivot's column; element 48 is a constant zero
    double* const TB = sg + O_TB;      // [33]: durations
    double* const SC = sg + O_SC;
    unsigned long long* const MK = reinterpret_cast<unsigned long long*>(sg + O_MK);
    int* const KT = reinterpret_cast<int*>(sg + O_IT);       // [34] per knot: first constraint that sits there | count << 8
    int* const CD = KT + 34;                                   // [48] per constraint: left knot | kind << 8 (0 box, 1 + slot) | segment << 12
    int* const CNT = CD + 48;                                  // [33] rows per segment
    auto ES = [&](int k) -> double* { return sg + (k - 1) * RS; };
    auto GR = [&](int i) -> double* { return sg + i * RS; };
    const int crd = c < NRW ? c : NRW, crow = min(c, NRW - 1);

    for (long long bq = blockIdx.x; bq < a.n_traj; bq += gridDim.x) {
        const int b = aa.order ? aa.order[bq] : (int)bq;
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        s0 = __builtin_amdgcn_readfirstlane(s0);
        M = __builtin_amdgcn_readfirstlane(M);
        const bool shape_ok = M >= 2 && M <= 32 && (a.uniform > 0 || M <= a.max_segments);
        if (!shape_ok) { if (lane == 0) aa.need_phase1[b] = 1; continue; }
        const int n = M - 1;
        lds_publish();
        // ---------------- lane s prepares segment s: the functionals of its rows ----------------
        const bool myseg = lane < M;
        bool segok = true;
        int nrow = 0, slot_of[K], d_of[K];
        double Tseg = 1.0, gls[K][R], grs[K][R];
#pragma unroll
        for (int j = 0; j < K; ++j) { slot_of[j] = 0; d_of[j] = 0; }
        if (myseg) {
            Tseg = a.times[s0 + lane];
            segok = Tseg > 0.0 && Tseg < INFINITY;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int d = a.row_deriv[(size_t)(s0 + lane) * K + j];
                const double tau = a.row_tau[(size_t)(s0 + lane) * K + j];
                if (d >= 0) {
                    segok = segok && d < R && tau >= 0.0 && tau < 1.0 && !(tau == 0.0 && d == 0);
                    double gl[R], gr[R];
                    const double* const gf = aa.gfun + ((size_t)(s0 + lane) * K + j) * 2 * R;     // (zeros for an invalid row: the trajectory is not handled then)
#pragma unroll
                    for (int q = 0; q < R; ++q) { gl[q] = gf[q]; gr[q] = gf[R + q]; }
                    // (compile-time slot index: the first used row goes to position 0)
                    if (nrow == 0) {
#pragma unroll
                        for (int q = 0; q < R; ++q) { gls[0][q] = gl[q]; grs[0][q] = gr[q]; }
                        slot_of[0] = j; d_of[0] = d;
                    } else {
#pragma unroll
                        for (int q = 0; q < R; ++q) { gls[K - 1][q] = gl[q]; grs[K - 1][q] = gr[q]; }
                        slot_of[K - 1] = j; d_of[K - 1] = d;
                    }
                    ++nrow;
                }
            }
            TB[lane] = Tseg;
        }
        if (lane <= 32) CNT[lane] = (myseg && lane < 33) ? nrow : 0;
        lds_publish();
        int roff = lane, NC = n;           // constraints: rows_0, box_1, rows_1, box_2, ..., box_n, rows_n
        for (int i = 0; i < M; ++i) {
            const int ci = CNT[i];
            if (i < lane) roff += ci;
            NC += ci;
        }
        const bool handled = (__ballot(segok) == ~0ull) && NC <= NRW;
        if (!handled) { if (lane == 0) aa.need_phase1[b] = 1; continue; }
        if (lane == 0) aa.need_phase1[b] = 0;
        if (myseg) {
#pragma unroll
            for (int jj = 0; jj < K; ++jj)
                if (jj < nrow) {
                    const int ci = roff + jj;
#pragma unroll
                    for (int q = 0; q < R; ++q) { GF[ci * 2 * R + q] = gls[jj][q]; GF[ci * 2 * R + R + q] = grs[jj][q]; }
                    CD[ci] = lane | ((1 + slot_of[jj]) << 8) | (lane << 12);
                }
            if (lane < n) {    // the box of the knot that closes this segment
                const int ci = roff + nrow;
#pragma unroll
                for (int q = 0; q < R; ++q) { GF[ci * 2 * R + q] = q == 0 ? 1.0 : 0.0; GF[ci * 2 * R + R + q] = 0.0; }
                CD[ci] = (lane + 1) | (0 << 8) | (lane << 12);
                const int nxt = CNT[lane + 1];
                KT[lane + 1] = (lane == 0 ? 0 : ci) | (((lane == 0 ? nrow : 0) + 1 + nxt) << 8);
            }
        }
        lds_publish();

        // ---------------- forward: block LDL' chain (one trajectory per wave: every lane computes it, lane 0 stores the records) ----------------
        FullBlocks<R> sa, seg0, segl;
        sa.build(TB[0]);
        seg0 = sa;
        segl.build(TB[M - 1]);
        Inv lprev;
        LDLPack<R>::zero(lprev);
#pragma unroll 1
        for (int k = 1; k <= n; ++k) {
            FullBlocks<R> sb;
            sb.build(TB[k]);
            double D[R][R], Yp[R][R], Zp[R][R];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int q = 0; q < R; ++q) D[i][q] = sa.B11[i][q] + sb.B00(i, q);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                double col[R];
#pragma unroll
                for (int i = 0; i < R; ++i) col[i] = sa.B01[i][q];
                lprev.forward(col);
#pragma unroll
                for (int i = 0; i < R; ++i) { Yp[i][q] = col[i]; Zp[i][q] = col[i] * lprev.dinv[i]; }
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int q = 0; q < R; ++q)
#pragma unroll
                    for (int cc = 0; cc <= i; ++cc) D[i][cc] -= Yp[q][i] * Zp[q][cc];
            if (k >= 2) {
                double E[R][R];
#pragma unroll
                for (int cc = 0; cc < R; ++cc) {
#pragma unroll
                    for (int i = R - 1; i >= 0; --i) {
                        double v = Zp[i][cc];
#pragma unroll
                        for (int q = i + 1; q < R; ++q) v -= lprev.l[q][i] * E[q][cc];
                        E[i][cc] = v;
                    }
                }
                if (lane == 0) {
                    double* const rec = ES(k - 1);
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int cc = 0; cc < R; ++cc) rec[NE + i * R + cc] = E[i][cc];
                }
            }
            Inv ldl;
            ldl.factor(D);
            {
                double Si[R][R];
#pragma unroll
                for (int cc = 0; cc < R; ++cc) {
                    double col[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) col[i] = (i == cc) ? 1.0 : 0.0;
                    ldl.solve(col);
#pragma unroll
                    for (int i = 0; i < R; ++i) Si[i][cc] = col[i];
                }
                if (lane == 0) {
                    double* const rec = ES(k);
                    int f = 0;
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int cc = 0; cc <= i; ++cc) rec[f++] = Si[i][cc];
                }
            }
            lprev = ldl;
            sa = sb;
        }
        lds_publish();

        // ---------------- backward: z_c = H^-1 c_c at and above its knots, the entries of G that do not sit behind it ----------------
        const bool vc = c < NC;
        const int cdc = vc ? CD[crow] : 0;
        const int kLc = vc ? (cdc & 255) : -100;       // left knot of this column's functional (0: the boundary knot)
        double gLc[R], gRc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { gLc[q] = vc ? GF[crow * 2 * R + q] : 0.0; gRc[q] = vc ? GF[crow * 2 * R + R + q] : 0.0; }
        double Zk1[R][R], Zkn[R][R], Ek[R][R], v[R], wv[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i] = 0.0; wv[i] = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) { Zk1[i][q] = 0.0; Zkn[i][q] = 0.0; Ek[i][q] = 0.0; }
        }
#pragma unroll 1
        for (int k = n; k >= 1; --k) {
            double Si[R][R], Em[R][R];      // S_k^-1, E_{k-1}
            {
                const double* const rec = ES(k);
                int f = 0;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q <= i; ++q) { Si[i][q] = rec[f]; Si[q][i] = Si[i][q]; ++f; }
                const double* const recm = ES(k >= 2 ? k - 1 : 1);
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) Em[i][q] = k >= 2 ? recm[NE + i * R + q] : 0.0;
            }
            const int kt = KT[k];
            lds_publish();
            double P[R][R], Zkk[R][R];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    double t = 0.0;
#pragma unroll
                    for (int p = 0; p < R; ++p) t += Ek[i][p] * Zk1[p][q];
                    P[i][q] = t;
                }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int q = 0; q <= i; ++q) {
                    double t = Si[i][q];
#pragma unroll
                    for (int p = 0; p < R; ++p) t += P[i][p] * Ek[q][p];
                    Zkk[i][q] = t;
                    Zkk[q][i] = t;
                }
            {
                const double dn = (k == n) ? 1.0 : 0.0;
                double Zn[R][R];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        double t = dn * Zkk[i][q];
#pragma unroll
                        for (int p = 0; p < R; ++p) t -= Ek[i][p] * Zkn[p][q];
                        Zn[i][q] = t;
                    }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) Zkn[i][q] = Zn[i][q];
            }
            // this lane's column
            const double dR = (k == kLc + 1) ? 1.0 : 0.0, dL = (k == kLc) ? 1.0 : 0.0;
            double inj1[R], inj2[R], wg[R], vn[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double t = gRc[i];
#pragma unroll
                for (int p = 0; p < R; ++p) t -= Em[p][i] * gLc[p];      // g_r - E_{k-1}' g_l   (E_0 = 0: a row of segment 0 has no variable on its left)
                inj1[i] = dR * t;
                inj2[i] = dL * gLc[i];
                wg[i] = dR * gRc[i] + dL * gLc[i];
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double t = 0.0;
#pragma unroll
                for (int p = 0; p < R; ++p) t += Zkk[i][p] * inj1[p] + Si[i][p] * inj2[p] - Ek[i][p] * v[p];
                vn[i] = t;
                double w = wv[i];
#pragma unroll
                for (int p = 0; p < R; ++p) w = fma(Zkn[p][i], wg[p], w);
                wv[i] = w;
            }
            // entries of G: constraints that sit at knot k against every column that does not sit in front of them
            const int cf = kt & 255, cnt = (kt >> 8) & 255;
            for (int t = 0; t < cnt; ++t) {
                const int i = cf + t;
                const int kLi = CD[i] & 255;
                double val = 0.0;
#pragma unroll
                for (int p = 0; p < R; ++p) {
                    const double ga = GF[i * 2 * R + p], gb = GF[i * 2 * R + R + p];
                    val += kLi == k ? (ga * vn[p] + gb * v[p]) : gb * vn[p];      // (kLi = 0 at k = 1: only its right knot is a variable)
                }
                if (vc && kLi <= kLc) {
                    GR(i)[c] = val;
                    GR(c)[i] = val;
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                v[i] = vn[i];
#pragma unroll
                for (int q = 0; q < R; ++q) { Zk1[i][q] = Zkk[i][q]; Ek[i][q] = Em[i][q]; }
            }
        }
        // rows and columns beyond the constraints hold what the chain records left there: they must be neutral in the sweeps
        if (c < NRW) {
            double* const row = GR(c);
#pragma unroll 1
            for (int i = c >= NC ? 0 : NC; i < NRW; ++i) row[i] = 0.0;
        }
        if (lane == 0) { CB[NRW] = 0.0; CB[NRW + 1] = 0.0; }
        lds_publish();

        // ---------------- per axis: unconstrained value and bounds of this lane's constraint ----------------
        double y0[3], lo3[3], hi3[3];
        {
            const int kind = (cdc >> 8) & 15, seg = (cdc >> 12) & 255;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const long long base3 = 3LL * ((long long)s0 + b) + ax;
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                double x0[R], xM[R], r1[R], rn[R];
                x0[0] = a.waypoints[base3];
                xM[0] = a.waypoints[base3 + 3LL * M];
#pragma unroll
                for (int d = 0; d < ND; ++d) { x0[d + 1] = bc[d * 3]; xM[d + 1] = bc[(ND + d) * 3]; }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    double v1 = 0.0, vn_ = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) { v1 -= seg0.B01[q][i] * x0[q]; vn_ -= segl.B01[i][q] * xM[q]; }
                    r1[i] = v1;
                    rn[i] = vn_;
                }
                double lo_ = 0.0, hi_ = 0.0, cst = 0.0, yv = 0.0;
                if (vc) {
                    if (kind == 0) {
                        const long long at = base3 + 3LL * kLc;
                        lo_ = a.corr_lo ? a.corr_lo[at] : a.waypoints[at];
                        hi_ = a.corr_hi ? a.corr_hi[at] : a.waypoints[at];
                    } else {
                        const size_t at = ((size_t)(s0 + seg) * K + (kind - 1)) * 3 + ax;
                        lo_ = a.row_lo[at];
                        hi_ = a.row_hi[at];
                        // the part of the functional that sits on a boundary knot is a constant: it moves the bounds
#pragma unroll
                        for (int q = 0; q < R; ++q) cst += (kLc == 0 ? gLc[q] * x0[q] : 0.0) + (kLc == n ? gRc[q] * xM[q] : 0.0);
                    }
#pragma unroll
                    for (int q = 0; q < R; ++q) yv += v[q] * r1[q] + wv[q] * rn[q];
                }
                lo3[ax] = lo_ - cst;
                hi3[ax] = hi_ - cst;
                y0[ax] = yv;
            }
        }
#ifdef UAVQP_DUAL_DEBUG
        // per trajectory (dealing position bq < 16): [0, 2304) G row-major [48][48]; [2304 + 192 ax + 48 what + col]: what 0 = y0, 1 = trips, 2 = y at the end;
        // [2304 + 576 + col] = constraint descriptors, [2304 + 640] = NC, [2304 + 700 + 8 ax + j] = the masks handed over
        double* const dbg = (aa.dbg && bq < 16) ? aa.dbg + bq * 4096 : nullptr;
        if (dbg) {
            if (vc) {
                for (int i = 0; i < NC; ++i) dbg[i * 48 + c] = GR(c)[i];
                dbg[2304 + 576 + c] = (double)cdc;
            }
            if (lane == 0) { dbg[2304 + 640] = NC; dbg[2304 + 641] = n; }
        }
#endif

        // ---------------- the three axes: the dual method, every branch wave-uniform ----------------
        double A[NRW];
        const int max_trips = 4 * NC + 16 + max_trips_extra;
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = axis == 0 ? lo3[0] : (axis == 1 ? lo3[1] : lo3[2]);
            const double hi = axis == 0 ? hi3[0] : (axis == 1 ? hi3[1] : hi3[2]);
            double y = axis == 0 ? y0[0] : (axis == 1 ? y0[1] : y0[2]);
            const double tol = 1e-12 * (1.0 + fmin(fabs(lo), fabs(hi)));
            const double eqb = (vc && lo == hi) ? 1e300 : 0.0;
            double dg = vc ? GR(crow)[crow] : 1.0, sw = 0.0;
            bool inW = false;
            lds_publish();
            {
                const double* const row = GR(crow);
#pragma unroll
                for (int i = 0; i < NRW; i += 2) {
                    const double2 tt = *reinterpret_cast<const double2_a*>(row + i);
                    A[i] = tt.x;
                    A[i + 1] = tt.y;
                }
            }
#ifdef UAVQP_DUAL_DEBUG
            if (dbg && vc) dbg[2304 + 192 * axis + c] = y;
#endif
            int trips = 0;
            for (;;) {
                lds_publish();
                // entering constraint: steepest dual ascent, violation^2 / T_qq
                const double below = lo - y, above = y - hi;
                const double viol = raw_max(below, above);
                const bool cand = vc && !inW && viol > tol && dg > 0.0;
                const double kv = raw_max(raw_min(viol * viol * __builtin_amdgcn_rcp(dg), 1e299), eqb);
                const double key = wave_max64(cand ? pack_code7(kv, (below > above ? 64 : 0) | c) : 0.0);
                if (!(key > 1e-300) || trips >= max_trips) break;
                const int cd = code7_of(key);
                const int q = __builtin_amdgcn_readfirstlane(cd & 63);
                const double sdir = (cd & 64) ? 1.0 : -1.0;
                double muq = 0.0;
                // the constraint moves towards its bound until it reaches it (it enters) -- each time a multiplier of the working set would
                // change sign first, that constraint leaves and the move goes on
                for (;;) {
                    lds_publish();
                    if (c == q) {
#pragma unroll
                        for (int i = 0; i < NRW; i += 2) *reinterpret_cast<double2_a*>(CB + i) = make_double2(A[i], A[i + 1]);
                        const double pv = rcp1(dg);
                        CB[q] = dg;
                        SC[0] = ((sdir > 0.0 ? lo : hi) - y) * sdir * pv;
                        SC[1] = pv;
                    }
                    lds_publish();
                    const double d = sdir * CB[crd];
                    const double t1 = SC[0];
                    const bool blocks = sw * d > 0.0;
                    const double ratio = raw_min(raw_max(-y * rcp1(d), 0.0), 1e299);
                    const double rmin = wave_min64(blocks ? pack_code7(ratio, c) : 1e300);
                    const bool partial = rmin < t1;
                    const double t = partial ? rmin : t1;
                    const int kp = __builtin_amdgcn_readfirstlane(partial ? (code7_of(rmin) & 63) : q);
                    y = fma(t, d, y);
                    muq = fma(sdir, t, muq);
                    if (partial) {
                        lds_publish();
                        if (c == kp) {
#pragma unroll
                            for (int i = 0; i < NRW; i += 2) *reinterpret_cast<double2_a*>(CB + i) = make_double2(A[i], A[i + 1]);
                            SC[1] = rcp1(dg);
                        }
                    }
                    if (c == kp) CB[kp] = dg - (partial ? -1.0 : 1.0);
                    lds_publish();
                    {
                        const double piv = SC[1];
                        const double tc = CB[crd];
                        const bool pc = c == kp;
                        const double s = tc * piv;
                        const double dn = fma(-tc, s, dg);
                        dg = pc ? -piv : dn;
                        const double yb = sw < 0.0 ? hi : lo;
                        y = pc ? (partial ? yb : -muq) : y;
                        sw = pc ? ((partial || eqb != 0.0) ? 0.0 : sdir) : sw;
                        inW = pc ? !partial : inW;
#pragma unroll
                        for (int i = 0; i < NRW; i += 2) {   // (all 48 rows: rows beyond the constraints are zero in every column -- a guard on the row count compiled to 96 selects)
                            const double2 u = *reinterpret_cast<const double2_a*>(CB + i);
                            A[i] = fma(-u.x, s, A[i]);
                            A[i + 1] = fma(-u.y, s, A[i + 1]);
                        }
                    }
                    ++trips;
                    if (!partial || trips >= max_trips) break;
                }
            }
            // ---- hand the working set of this axis over in the rows kernel's layout: boxes by interior knot, rows by slot and segment
            lds_publish();
            if (lane < 2 + 2 * K) MK[lane] = 0ull;
            lds_publish();
            if (vc && inW && sw != 0.0) {
                const int kind = (cdc >> 8) & 15, seg = (cdc >> 12) & 255;
                const int word = kind == 0 ? 0 : 2 * kind;
                const unsigned long long bit = 1ull << (kind == 0 ? kLc : seg);
                atomicOr(&MK[word], bit);
                if (sw < 0.0) atomicOr(&MK[word + 1], bit);
            }
            lds_publish();
#ifdef UAVQP_DUAL_DEBUG
            if (dbg && vc) { dbg[2304 + 192 * axis + 96 + c] = y; dbg[2304 + 192 * axis + 48 + c] = (double)trips; }
            if (dbg && lane == 0) for (int j = 0; j < 2 + 2 * K; ++j) dbg[2304 + 700 + 8 * axis + j] = (double)MK[j];
#endif
            if (lane == 0) {
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
