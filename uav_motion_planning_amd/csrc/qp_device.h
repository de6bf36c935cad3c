// qp_device.h -- device-side maths of the reduced min-jerk / min-snap solve (gfx950, float64).
//
// The reference QP (minimum_control.cpp:5-125) is posed in 2r monomial coefficients per segment with
// equality rows only.  Each segment polynomial is fixed by its endpoint derivatives 0..r-1 (Hermite
// data); positions at every knot and all derivatives at the two ends are given, so the only free
// quantities are the r-1 derivatives y_k = (v_k, a_k[, j_k]) at the M-1 interior knots.  In
// normalised time tau = t/T the segment cost is
//     J = T^(1-2r) e' W e,   e = s1 - C s0,   s*_d = T^d p^(d)(end)
// (W, V = C'W, U = C'WC, K: exact-rational constants, hermite_tables.h / tools/derive_tables.py).
// Stationarity in y_k gives an SPD block-tridiagonal system with (r-1)x(r-1) blocks that depends
// only on the time allocation -- the three axes are three right-hand sides of one factorisation
// (the reference re-runs OSQP setup per axis on identical P, A: test_minimum_jerk.cpp:75,100,125).
#pragma once
#include <hip/hip_runtime.h>

#include "hermite_tables.h"

namespace uavqp {

template <int R> struct Tab;
template <> struct Tab<3> {
    static __device__ __forceinline__ constexpr double K(int i, int j) { return HK3[i][j]; }
    static __device__ __forceinline__ constexpr double W(int i, int j) { return HW3[i][j]; }
    static __device__ __forceinline__ constexpr double V(int i, int j) { return HV3[i][j]; }
    static __device__ __forceinline__ constexpr double U(int i, int j) { return HU3[i][j]; }
};
template <> struct Tab<4> {
    static __device__ __forceinline__ constexpr double K(int i, int j) { return HK4[i][j]; }
    static __device__ __forceinline__ constexpr double W(int i, int j) { return HW4[i][j]; }
    static __device__ __forceinline__ constexpr double V(int i, int j) { return HV4[i][j]; }
    static __device__ __forceinline__ constexpr double U(int i, int j) { return HU4[i][j]; }
};

// 1/x: v_rcp_f64 seed + two Newton steps (full float64 accuracy for normal x; no denormal/inf fix-up,
// inputs are validated time allocations and SPD pivots).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// exchange a double with the neighbouring lane (lane ^ 1) through DPP quad_perm:[1,0,3,2]
__device__ __forceinline__ double swap_pair(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ constexpr double inv_fact(int k) {
    double f = 1.0;
    for (int j = 2; j <= k; ++j) f *= (double)j;
    return 1.0 / f;
}

// T-dependent blocks of one segment (ND = R-1 derivative unknowns per knot).
//   A11: end/end block      T^(a+b+1-2R) W[a][b]          (symmetric, lower triangle stored)
//   A00: start/start block  = (-1)^(a+b) A11[a][b]        (U = (-1)^(a+b) W: time-reversal symmetry)
//   A01: start/end block   -T^(a+b+1-2R) V[a][b]
//   gw : position coupling  T^(a+1-2R) W[a][0];   gv[a] = T^(a+1-2R) V[a][0] = (-1)^a gw[a]
// Only A11 / A01 / gw are materialised; A00 and gv are sign patterns applied at the use sites.
template <int R>
struct SegBlocks {
    static constexpr int ND = R - 1;
    double A11[ND][ND];
    double A01[ND][ND];
    double gw[ND];
    __device__ __forceinline__ void build(double T) {
        double ip[2 * R];
        build(T, ip);
    }
    // same, handing out the inverse powers ip[j] = T^-j (j = 0..2R-1) it computes on the way: the emission needs T^-1 and
    // T^-R..T^-(2R-1) of the same segment again (segment_coeffs_ip)
    __device__ __forceinline__ void build(double T, double (&ip)[2 * R]) {
        const double it = fast_rcp(T);
        ip[0] = 1.0;
#pragma unroll
        for (int j = 1; j < 2 * R; ++j) ip[j] = ip[j - 1] * it;
#pragma unroll
        for (int a = 1; a < R; ++a) {
#pragma unroll
            for (int b = 1; b < R; ++b) {
                const double p = ip[2 * R - 1 - a - b];
                if (b <= a) A11[a - 1][b - 1] = p * Tab<R>::W(a, b);
                A01[a - 1][b - 1] = -p * Tab<R>::V(a, b);
            }
            gw[a - 1] = ip[2 * R - 1 - a] * Tab<R>::W(a, 0);
        }
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int b = a + 1; b < ND; ++b) A11[a][b] = A11[b][a];
    }
    // (i, c) 0-based over derivative orders d = i+1, c+1
    __device__ __forceinline__ double A00(int i, int c) const { return ((i + c) & 1) ? -A11[i][c] : A11[i][c]; }
    __device__ __forceinline__ double gv(int i) const { return (i & 1) ? gw[i] : -gw[i]; }
};

// In-place LDL' of a small SPD matrix (only the lower triangle is read), then solves.
template <int N>
struct SmallLDL {
    double l[N][N];  // strictly-lower entries of L
    double d[N];
    double dinv[N];
    __device__ __forceinline__ void factor(const double (&S)[N][N]) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double w[N];
            double dj = S[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) {
                w[k] = l[j][k] * d[k];
                dj -= l[j][k] * w[k];
            }
            d[j] = dj;
            dinv[j] = fast_rcp(dj);
#pragma unroll
            for (int i = j + 1; i < N; ++i) {
                double s = S[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) s -= l[i][k] * w[k];
                l[i][j] = s * dinv[j];
            }
        }
    }
    // x <- L^-1 x (forward substitution only);  S^-1 = L^-T D^-1 L^-1, so M' S^-1 M = Y' D^-1 Y with Y = L^-1 M
    __device__ __forceinline__ void forward(double (&x)[N]) const {
#pragma unroll
        for (int i = 1; i < N; ++i)
#pragma unroll
            for (int k = 0; k < i; ++k) x[i] -= l[i][k] * x[k];
    }
    // x <- S^-1 x
    __device__ __forceinline__ void solve(double (&x)[N]) const {
#pragma unroll
        for (int i = 1; i < N; ++i)
#pragma unroll
            for (int k = 0; k < i; ++k) x[i] -= l[i][k] * x[k];
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] *= dinv[i];
#pragma unroll
        for (int i = N - 2; i >= 0; --i)
#pragma unroll
            for (int k = i + 1; k < N; ++k) x[i] -= l[k][i] * x[k];
    }
};

// Explicit inverse of a small SPD matrix by cofactors (only the lower triangle of S is read): ONE reciprocal
// (of the determinant) instead of N dependent pivots -- a much shorter dependency chain than LDL', which is
// what matters when a single wave per SIMD runs the elimination (latency shape).  Cofactor expansion is
// invariant under symmetric diagonal scaling, so the badly scaled T^-5..T^-1 entries do not hurt; the
// blocks themselves are well conditioned after scaling (checked by the parity tests at 1e-9).
template <int N>
struct SymInv;
template <>
struct SymInv<2> {
    double i00, i10, i11;
    __device__ __forceinline__ void factor(const double (&S)[2][2]) {
        const double det = S[0][0] * S[1][1] - S[1][0] * S[1][0];
        const double r = fast_rcp(det);
        i00 = S[1][1] * r;
        i10 = -S[1][0] * r;
        i11 = S[0][0] * r;
    }
    __device__ __forceinline__ void solve(double (&x)[2]) const {
        const double a = x[0], b = x[1];
        x[0] = i00 * a + i10 * b;
        x[1] = i10 * a + i11 * b;
    }
};
template <>
struct SymInv<3> {
    double i00, i10, i11, i20, i21, i22;
    __device__ __forceinline__ void factor(const double (&S)[3][3]) {
        const double a = S[0][0], b = S[1][0], c = S[2][0], d = S[1][1], e = S[2][1], f = S[2][2];
        const double c00 = d * f - e * e, c10 = c * e - b * f, c20 = b * e - c * d;
        const double c11 = a * f - c * c, c21 = b * c - a * e, c22 = a * d - b * b;
        const double det = a * c00 + b * c10 + c * c20;
        const double r = fast_rcp(det);
        i00 = c00 * r; i10 = c10 * r; i20 = c20 * r;
        i11 = c11 * r; i21 = c21 * r; i22 = c22 * r;
    }
    __device__ __forceinline__ void solve(double (&x)[3]) const {
        const double a = x[0], b = x[1], c = x[2];
        x[0] = i00 * a + i10 * b + i20 * c;
        x[1] = i10 * a + i11 * b + i21 * c;
        x[2] = i20 * a + i21 * b + i22 * c;
    }
};

// packed access to the N (N + 1) / 2 numbers of a SmallLDL (strict lower triangle of L, inverse pivots): sweep-state records
template <int N>
struct LDLPack {
    static constexpr int NE = N * (N + 1) / 2;
    static __device__ __forceinline__ void get(const SmallLDL<N>& v, double (&o)[NE]) {
        int f = 0;
#pragma unroll
        for (int i = 1; i < N; ++i)
#pragma unroll
            for (int c = 0; c < i; ++c) o[f++] = v.l[i][c];
#pragma unroll
        for (int i = 0; i < N; ++i) o[f++] = v.dinv[i];
    }
    static __device__ __forceinline__ void set(SmallLDL<N>& v, const double (&o)[NE]) {
        int f = 0;
#pragma unroll
        for (int i = 1; i < N; ++i)
#pragma unroll
            for (int c = 0; c < i; ++c) v.l[i][c] = o[f++];
#pragma unroll
        for (int i = 0; i < N; ++i) v.dinv[i] = o[f++];
    }
    // "S^-1 = 0": solve() returns the zero vector (a knot with nothing free)
    static __device__ __forceinline__ void zero(SmallLDL<N>& v) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v.dinv[i] = 0.0;
#pragma unroll
            for (int c = 0; c < N; ++c) v.l[i][c] = 0.0;
        }
    }
};

// Monomial coefficients (ascending powers, segment-local time: the reference's coef_1d_ layout,
// minimum_control.cpp:186) of one segment of one axis from its Hermite data.
//   ys / ye : derivatives 1..R-1 at the segment start / end;  p0 / p1 : positions.
template <int R>
__device__ __forceinline__ void segment_coeffs(double p0, const double (&ys)[R - 1], double p1,
                                               const double (&ye)[R - 1], double T, double it,
                                               double (&c)[2 * R]) {
    double tp[R];  // T^d
    tp[0] = 1.0;
#pragma unroll
    for (int d = 1; d < R; ++d) tp[d] = tp[d - 1] * T;
    double s0[R], s1[R];
    s0[0] = 0.0;  // position handled through dp = p1 - p0 (translation invariance)
    s1[0] = p1 - p0;
#pragma unroll
    for (int d = 1; d < R; ++d) {
        s0[d] = tp[d] * ys[d - 1];
        s1[d] = tp[d] * ye[d - 1];
    }
    double e[R];
#pragma unroll
    for (int d = 0; d < R; ++d) {
        double acc = s1[d];
#pragma unroll
        for (int k = (d > 1 ? d : 1); k < R; ++k) acc -= s0[k] * inv_fact(k - d);
        e[d] = acc;
    }
    c[0] = p0;
#pragma unroll
    for (int d = 1; d < R; ++d) c[d] = ys[d - 1] * inv_fact(d);
    double ipw = 1.0;
#pragma unroll
    for (int d = 0; d < R; ++d) ipw *= it;  // T^-R
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double q = 0.0;
#pragma unroll
        for (int d = 0; d < R; ++d) q += Tab<R>::K(j, d) * e[d];
        c[R + j] = q * ipw;
        ipw *= it;
    }
}

// The same arithmetic with every fused multiply-add spelled out and contraction off: the result does not depend on the code the function
// is inlined into.  The corridor path emits from three places (corridor_prep_kernel: one-segment trajectories, corridor_solve_kernel in
// its template variants, corridor_emit_kernel behind the rows solvers) and their polynomials must agree to the bit for the same Hermite data.
template <int R>
__device__ __forceinline__ void segment_coeffs_det(double p0, const double (&ys)[R - 1], double p1,
                                                   const double (&ye)[R - 1], double T, double it,
                                                   double (&c)[2 * R]) {
#pragma clang fp contract(off)
    double tp[R];  // T^d
    tp[0] = 1.0;
#pragma unroll
    for (int d = 1; d < R; ++d) tp[d] = tp[d - 1] * T;
    double s0[R], s1[R];
    s0[0] = 0.0;
    s1[0] = p1 - p0;
#pragma unroll
    for (int d = 1; d < R; ++d) {
        s0[d] = tp[d] * ys[d - 1];
        s1[d] = tp[d] * ye[d - 1];
    }
    double e[R];
#pragma unroll
    for (int d = 0; d < R; ++d) {
        double acc = s1[d];
#pragma unroll
        for (int k = (d > 1 ? d : 1); k < R; ++k) acc = fma(-s0[k], inv_fact(k - d), acc);
        e[d] = acc;
    }
    c[0] = p0;
#pragma unroll
    for (int d = 1; d < R; ++d) c[d] = ys[d - 1] * inv_fact(d);
    double ipw = 1.0;
#pragma unroll
    for (int d = 0; d < R; ++d) ipw *= it;  // T^-R
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double q = Tab<R>::K(j, 0) * e[0];
#pragma unroll
        for (int d = 1; d < R; ++d) q = fma(Tab<R>::K(j, d), e[d], q);
        c[R + j] = q * ipw;
        ipw *= it;
    }
}

// segment_coeffs with the inverse powers of the duration supplied (ip[j] = T^-j, from SegBlocks::build): no reciprocal, no power chain.
template <int R>
__device__ __forceinline__ void segment_coeffs_ip(double p0, const double (&ys)[R - 1], double p1, const double (&ye)[R - 1], double T,
                                                  const double (&ip)[2 * R], double (&c)[2 * R]) {
    double tp[R];  // T^d
    tp[0] = 1.0;
#pragma unroll
    for (int d = 1; d < R; ++d) tp[d] = tp[d - 1] * T;
    double s0[R], s1[R];
    s0[0] = 0.0;
    s1[0] = p1 - p0;
#pragma unroll
    for (int d = 1; d < R; ++d) {
        s0[d] = tp[d] * ys[d - 1];
        s1[d] = tp[d] * ye[d - 1];
    }
    double e[R];
#pragma unroll
    for (int d = 0; d < R; ++d) {
        double acc = s1[d];
#pragma unroll
        for (int k = (d > 1 ? d : 1); k < R; ++k) acc -= s0[k] * inv_fact(k - d);
        e[d] = acc;
    }
    c[0] = p0;
#pragma unroll
    for (int d = 1; d < R; ++d) c[d] = ys[d - 1] * inv_fact(d);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double q = 0.0;
#pragma unroll
        for (int d = 0; d < R; ++d) q += Tab<R>::K(j, d) * e[d];
        c[R + j] = q * ip[R + j];
    }
}

}  // namespace uavqp
