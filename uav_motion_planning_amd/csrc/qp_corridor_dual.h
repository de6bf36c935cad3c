// qp_corridor_dual.h -- the starting working set of the corridor solve (qp_corridor.h) from a DUAL active-set method in POSITION
// space (round 4).  Same QP as qp_corridor.h (the reference's interior-waypoint equalities, minimum_control.cpp:34-42,118-124,
// relaxed to boxes lo <= p_k <= hi); nothing here decides a result: the set found is handed to corridor_solve_kernel as its
// cold-start guess, which verifies the KKT conditions with its own exact block solve (one iteration when the set is right, its usual
// primal iterations from there when rounding made this kernel miss a near-degenerate bound).
//
// Why: corridor_solve_kernel pays one block-tridiagonal solve over ALL knots (~550 instructions per knot and lane pair) per change
// of the working set, 12.5 of them per problem on config 3, and its batches are as slow as their slowest problem (config 5: up to
// 54 solves).  Eliminating the derivatives once leaves a dense strictly convex QP in the n = M - 1 knot positions,
//        min 1/2 p' Hp p - f' p,   lo <= p <= hi,     G = Hp^-1 = [H^-1]_pp  (H: the block-tridiagonal Hermite Hessian),
// whose inverse Hessian G depends on the time allocation only (shared by the three axes).  On G the Goldfarb-Idnani dual method is
// a sequence of symmetric SWEEPS of an n x n tableau T = sweep(G, W): T_FF = G_FF - G_FW G_WW^-1 G_WF (response of the free
// positions to a multiplier), T_FW = G_FW G_WW^-1, T_WW = -G_WW^-1.  A bound that enters or leaves the working set W is one rank-one
// update of the tableau (n^2 FMAs) -- no factorisation, no solve.  Constraint choice: steepest dual ascent, violation^2 / T_qq
// (tools/corridor_dual_probe.py: 5.9 exchanges mean / 14 max on config 3 against 11.8 / 28 for "most violated"; the working set at
// the end equals the primal method's on 450 / 450 problems of configs 3 and 5).
//
// Layout: L lanes (8 or 16: one half / one whole DPP row) per trajectory, lane l owns the tableau columns l and l + L in registers
// with compile-time row indices; 64 / L trajectories per wave, their three axes one after the other.
//   * G: the block LDL' chain (S_k, E_k = S_k^-1 X_k) is computed once per trajectory -- replicated in the lanes of the group, the
//     records go through LDS --, then Z_kk = S_k^-1 + E_k Z_{k+1,k+1} E_k' backwards and, per owned column j, the vector
//     recursion Z_{k,j} e_0 = -E_k Z_{k+1,j} e_0: G_kj = e_0' Z_{k,j} e_0.  The same pass leaves Z_{1,j} e_0 and e_0' Z_{j,n}, from
//     which the unconstrained minimiser of every axis follows by two small dot products (its right-hand side lives at the first
//     and last interior knot only).
//   * a sweep on pivot k (runtime, per group): the owner lane writes its column to LDS (the only way to index registers by a
//     run-time value is not to: the column leaves through memory), patched so that ONE generic update A[i] -= u[i] * s serves
//     every row including row k:  u_k = t_k - sign(t_k), s_c = t_c / t_k, s_k = 1 - 1 / |t_k|; the diagonal is kept apart.
//   * reductions over the group (best entering constraint, first blocking multiplier) are DPP butterflies on doubles that carry
//     the column index in their low mantissa bits.
// Rows and columns beyond n are zero and never selected; ragged batches pad the chain with decoupled identity knots.
//
// Three kernels share the method: corridor_dual_kernel / corridor_dual_mixed_kernel (the batch shapes above: they build G, and with
// prep_in_dual they are also the reset / validation / descriptor kernel of the solve), and -- for the re-solves of an outer loop whose G
// is in the cache (gcache_mode 2) -- corridor_dual_wave_kernel / corridor_dual_wave2_kernel at the end of this file: one / two
// trajectories per wave, wave- / half-uniform pivots, a trip without LDS (the wave-wide helpers below).
#pragma once
#include "qp_corridor.h"

namespace uavqp {

// v_max_f64 / v_min_f64 as they are: fmax() / fmin() first quiet a possible signalling NaN of each operand (one extra v_max_f64 x, x
// per operand and call -- a fifth of the selection and ratio-test instructions of a trip); the values here are never NaN-sensitive
__device__ __forceinline__ double raw_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double raw_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);      // (every control used here reads a valid lane: no `old` value to keep, no copy)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// butterfly over the L lanes of a group (quad swaps, half-row mirror, row mirror): every lane ends with the result
template <int L>
__device__ __forceinline__ double group_max(double v) {
    v = raw_max(v, dpp_f64<0xB1>(v));
    v = raw_max(v, dpp_f64<0x4E>(v));
    v = raw_max(v, dpp_f64<0x141>(v));
    if (L == 16) v = raw_max(v, dpp_f64<0x140>(v));
    return v;
}
template <int L>
__device__ __forceinline__ double group_min(double v) {
    v = raw_min(v, dpp_f64<0xB1>(v));
    v = raw_min(v, dpp_f64<0x4E>(v));
    v = raw_min(v, dpp_f64<0x141>(v));
    if (L == 16) v = raw_min(v, dpp_f64<0x140>(v));
    return v;
}
// a non-negative double with a 6-bit code in its low mantissa bits (order-preserving up to 64 ulp)
__device__ __forceinline__ double pack_code(double v, int code) {
    return __longlong_as_double((__double_as_longlong(v) & ~63ll) | (long long)code);
}
__device__ __forceinline__ int code_of(double v) { return (int)(__double_as_longlong(v) & 63ll); }
// 1/x: hardware seed + one Newton step (2e-15 relative, tools/ubench/rcp_accuracy.hip) -- this kernel decides nothing finally
__device__ __forceinline__ double rcp1(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

// The lanes of a group talk to each other through LDS without a barrier (single-wave workgroup: the DS operations of a wave execute
// in program order).  The COMPILER knows nothing of that: to it a store under `if (owner)` followed by a load of the same address is
// single-thread data flow (it forwards the owner's value to the owner and an earlier load to everybody else).  lds_publish() is the
// compiler-side half of the hand-over: nothing emitted, no memory access moved or forwarded across it.
__device__ __forceinline__ void lds_publish() { asm volatile("" ::: "memory"); }
typedef double2 __attribute__((may_alias)) double2_a;

// LDS of one group, in doubles.  Slots of RS doubles: the chain record of knot k (S_k^-1: NE numbers, E_k: R R) sits in slot k - 1,
// row i of G in slot i -- the backward pass reads record k first and then writes rows >= k - 1 of G over the records it no
// longer needs (slots >= k - 1), so G and the chain records share their memory.  Then the column buffer and 8 scalars.
// r = 4: E_k is NOT kept (26 doubles per knot would make the slot 28 doubles wide for a G of 16: 31.5 KB per wave, five waves per CU): the backward pass
// re-makes it as S_k^-1 X_k from the duration of segment k -- ~125 instructions per knot for 19 KB per wave, eight waves per CU (config 5's first
// prelude: the batches of both shapes fit two rounds instead of three)
constexpr bool corridor_dual_keeps_E(int R) { return R != 4; }
constexpr int corridor_dual_rec(int R) { return ((R * (R + 1) / 2 + (corridor_dual_keeps_E(R) ? R * R : 0) + 1) & ~1); }
constexpr int corridor_dual_slot(int R, int NRW) { return NRW > corridor_dual_rec(R) ? NRW : corridor_dual_rec(R); }
constexpr int corridor_dual_lds_doubles(int R, int L, int NRW) {
    return NRW * corridor_dual_slot(R, NRW) + 2 * (NRW + 2) + 8;
}

// ---- wave-wide helpers of the one-trajectory-per-wave kernels (corridor_dual_wave_kernel below, rows_dual_kernel in qp_rows_dual.h)
__device__ __forceinline__ double pack_code7(double v, int code) {
    return __longlong_as_double((__double_as_longlong(v) & ~127ll) | (long long)code);
}
__device__ __forceinline__ int code7_of(double v) { return (int)(__double_as_longlong(v) & 127ll); }

typedef double v16d __attribute__((ext_vector_type(16)));
// lane `src` (wave-uniform) of a double
__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// x = [r0 r1 r2 r3] (the four DPP rows of a wave) -> a = [r0 r0 r0 r0], b = [r1 ...], c = [r2 ...]: v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second
__device__ __forceinline__ void row_replicas32(unsigned x, unsigned& a, unsigned& b, unsigned& c) {
    const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);       // [r0 r0 r2 r2], [r1 r1 r3 r3]
    const auto u = __builtin_amdgcn_permlane32_swap(p[0], p[0], false, false); // [r0 r0 r0 r0], [r2 r2 r2 r2]
    const auto w = __builtin_amdgcn_permlane32_swap(p[1], p[1], false, false); // [r1 r1 r1 r1], [r3 ...]
    a = u[0]; c = u[1]; b = w[0];
}
__device__ __forceinline__ void row_replicas(double x, double& a, double& b, double& c) {
    unsigned al, bl, cl, ah, bh, ch;
    row_replicas32((unsigned)__double2loint(x), al, bl, cl);
    row_replicas32((unsigned)__double2hiint(x), ah, bh, ch);
    a = __hiloint2double((int)ah, (int)al); b = __hiloint2double((int)bh, (int)bl); c = __hiloint2double((int)ch, (int)cl);
}
// maximum / minimum over the wave, every lane ends with it, no LDS: DPP inside the rows, permlane swaps across them
__device__ __forceinline__ void cross_rows(double v, double& p, double& q) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    p = __hiloint2double((int)h[0], (int)l[0]); q = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void cross_halves(double v, double& p, double& q) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    p = __hiloint2double((int)h[0], (int)l[0]); q = __hiloint2double((int)h[1], (int)l[1]);
}
// acc += (lane I of the own DPP row of t) * ns, one instruction (v_fmac_f64 takes DPP row_newbcast on gfx90a and later)
// (a VGPR written by the VALU must be two wait states old before a DPP operand reads it, and the compiler's hazard recogniser does not
// look into the string: the first instruction of a run carries its own s_nop)
template <int I, bool FIRST = false>
__device__ __forceinline__ double fmac_rowbcast(double acc, double t, double ns) {
    if (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(ns), "n"(I));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(ns), "n"(I));
    return acc;
}
// row kq (wave-uniform) of a column held as three 16-row vectors
__device__ __forceinline__ double pick_row(const v16d A0, const v16d A1, const v16d A2, int kq) {
    const int e = kq & 15;
    const double x0 = A0[e], x1 = A1[e], x2 = A2[e];
    return kq < 16 ? x0 : (kq < 32 ? x1 : x2);
}
__device__ __forceinline__ void sweep16(v16d& A, double t, double ns) {
    A[0] = fmac_rowbcast<0, true>(A[0], t, ns);   A[1] = fmac_rowbcast<1>(A[1], t, ns);   A[2] = fmac_rowbcast<2>(A[2], t, ns);   A[3] = fmac_rowbcast<3>(A[3], t, ns);
    A[4] = fmac_rowbcast<4>(A[4], t, ns);   A[5] = fmac_rowbcast<5>(A[5], t, ns);   A[6] = fmac_rowbcast<6>(A[6], t, ns);   A[7] = fmac_rowbcast<7>(A[7], t, ns);
    A[8] = fmac_rowbcast<8>(A[8], t, ns);   A[9] = fmac_rowbcast<9>(A[9], t, ns);   A[10] = fmac_rowbcast<10>(A[10], t, ns); A[11] = fmac_rowbcast<11>(A[11], t, ns);
    A[12] = fmac_rowbcast<12>(A[12], t, ns); A[13] = fmac_rowbcast<13>(A[13], t, ns); A[14] = fmac_rowbcast<14>(A[14], t, ns); A[15] = fmac_rowbcast<15>(A[15], t, ns);
}

// maximum of a 32-bit key over the wave (the entering constraint: a float's bits with the column in the low mantissa bits)
template <int CTRL>
__device__ __forceinline__ unsigned umax_dpp(unsigned v) {
    return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = umax_dpp<0xB1>(v);
    v = umax_dpp<0x4E>(v);
    v = umax_dpp<0x141>(v);
    v = umax_dpp<0x140>(v);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max(r[0], r[1]);
    const auto h = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max(h[0], h[1]);
}
__device__ __forceinline__ double wave_min64(double v) {
    v = raw_min(v, dpp_f64<0xB1>(v));
    v = raw_min(v, dpp_f64<0x4E>(v));
    v = raw_min(v, dpp_f64<0x141>(v));
    v = raw_min(v, dpp_f64<0x140>(v));
    double p, q;
    cross_rows(v, p, q);
    v = raw_min(p, q);
    cross_halves(v, p, q);
    return raw_min(p, q);
}


// One group of L lanes per trajectory; the block handles 64 / L trajectories at a time, grid-stride over the batch in the dealing
// order of the solve kernel (a.order, longest first), so that the trajectories of a wave are of similar length.
template <int R, int L, int NRW>
__device__ __forceinline__ void corridor_dual_body(const CorridorArgs& a, int n_lo, int max_trips_extra, bool last, double* s_all, int block, int n_blocks) {
    constexpr int ND = R - 1, NG = 64 / L, NE = R * (R + 1) / 2, LOG2L = (L == 16) ? 4 : 3;
    constexpr int RS = corridor_dual_slot(R, NRW), CBS = NRW + 2;
    constexpr int O_CB = NRW * RS, O_SC = O_CB + 2 * CBS, GRP = corridor_dual_lds_doubles(R, L, NRW);
    static_assert(NRW % 2 == 0 && NRW <= 2 * L && NRW <= 32, "rows: even, at most two columns per lane, codes are 5 bits");
    static_assert(O_CB % 2 == 0 && GRP % 2 == 0 && RS % 2 == 0, "16-byte aligned rows");
    using Inv = SmallLDL<R>;
    const int lane = threadIdx.x, l = lane & (L - 1), grp = lane >> LOG2L;
    double* const sg = s_all + grp * GRP;
    double* const CB = sg + O_CB;     // [2][NRW + 2]: the two columns of the lane that owns the pivot; element NRW is a constant zero
    double* const SC = sg + O_SC;
    auto ES = [&](int k) -> double* { return sg + (k - 1) * RS; };   // chain record of knot k = 1..NRW
    auto GR = [&](int i) -> double* { return sg + i * RS; };         // row i of G
    const int cidx[2] = {l, l + L};                                         // owned columns (variable j <-> interior knot j + 1)
    const int crd[2] = {cidx[0] < NRW ? cidx[0] : NRW, cidx[1] < NRW ? cidx[1] : NRW};   // row of the column buffer a lane reads (NRW: the zero)
    const int crow[2] = {min(cidx[0], NRW - 1), min(cidx[1], NRW - 1)};    // clamped row of G

    const int n_eff = a.n_active ? *a.n_active : a.n_traj;     // (a masked re-solve deals only the trajectories that take part)
    const long long n_batches = ((long long)n_eff + NG - 1) / NG;
    for (long long bt = block; bt < n_batches; bt += n_blocks) {
        // ---------------- the group's trajectory ----------------
        const long long bq = bt * NG + grp;
        const bool have = bq < n_eff;
        const int b = have ? (a.order ? a.order[bq] : (int)bq) : 0;
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        // a.prep_in_dual: no corridor_reset_kernel / corridor_prep_kernel ran -- this kernel also validates the trajectories it visits (every
        // trajectory of the dealing order is `mine` for exactly one launch shape: by n_lo and NRW, the last shape also takes what is longer
        // than any tableau, which can only be an invalid segment count), resets status and iteration count, writes the problem descriptors
        // and emits the one-segment trajectories; the boxes are checked where the axes' loads bring them in anyway
        const bool prep = a.prep_in_dual != 0;
        const bool mine = have && (n_lo <= 1 || M - 1 >= n_lo) && (last || M - 1 <= NRW);
        const bool shape_ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;
        const bool fits = shape_ok && M >= 2 && M - 1 <= NRW && M - 1 >= n_lo;
        const unsigned long long gmask = ((L == 16) ? 0xFFFFull : 0xFFull) << (grp * L);
        unsigned long long dsc[3];
        bool solve_any;
        bool t_ok = true;
        if (prep) {
            // durations: every lane checks the ones it is about to stage (indices 0..NRW cover M <= NRW + 1 segments)
            const double* const TTp = a.times + s0;
            bool bad = false;
            if (mine && shape_ok && M - 1 <= NRW) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                    if (cidx[sl] < M && cidx[sl] < NRW) { const double t = TTp[cidx[sl]]; bad = bad || !((t > 0.0) && (t < INFINITY)); }
                if (l == 0 && NRW < M) { const double t = TTp[NRW]; bad = bad || !((t > 0.0) && (t < INFINITY)); }
            }
            t_ok = (__ballot(bad) & gmask) == 0ull;
            solve_any = mine && fits && t_ok;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dsc[ax] = solve_any ? 1ull : 0ull;      // (the equality rows and the box check follow below)
            if (mine && !solve_any) {
                // nothing to solve here: an invalid trajectory (left untouched), or a single segment (M = 1: its polynomial follows from the boundary data)
                const bool valid1 = shape_ok && t_ok && M == 1;
                if (l == 0) {
                    a.status[b] = (shape_ok && t_ok) ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
                    if (a.iters) a.iters[b] = 0;
                }
                if (l < 3) {
                    a.desc[3LL * b + l] = 0ull;
                    if (valid1 && a.active) { a.active[2 * (3LL * b + l)] = 0ull; a.active[2 * (3LL * b + l) + 1] = 0ull; }
                    if (valid1) {
                        constexpr int NC = 2 * R;
                        const long long base3 = 3LL * ((long long)s0 + b) + l;
                        const double* bc = a.bc + (size_t)b * 2 * ND * 3 + l;
                        double ys[ND], ye[ND], c1[NC];
#pragma unroll
                        for (int d = 0; d < ND; ++d) { ys[d] = bc[d * 3]; ye[d] = bc[(ND + d) * 3]; }
                        const double Tk = TTp[0];
                        segment_coeffs_det<R>(a.waypoints[base3], ys, a.waypoints[base3 + 3], ye, Tk, fast_rcp(Tk), c1);
                        if (!((fabs(c1[NC - 1]) < INFINITY) && (fabs(c1[R]) < INFINITY))) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                        double* o = a.coeff + ((size_t)3 * s0 + l) * NC;
#pragma unroll
                        for (int j = 0; j < NC; ++j) o[j] = c1[j];
                    }
                }
            }
        } else {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dsc[ax] = have ? a.desc[3LL * b + ax] : 0ull;
            // (a batch of mixed lengths is covered by two launches: this one takes the trajectories with n_lo <= M - 1 <= NRW interior knots)
            solve_any = ((dsc[0] | dsc[1] | dsc[2]) & 1ull) && M >= 2 && M - 1 <= NRW && M - 1 >= n_lo;
        }
        if (__ballot(solve_any) == 0ull) continue;
        if (!solve_any) M = 2;                       // (keeps every index below in range; nothing is written for this group)
        const int n = M - 1;                         // variables = interior knots 1..n
        const double* const TT = a.times + s0;
        // durations through LDS (the column buffer is free until the dual phase): one coalesced load per lane instead of a dependent
        // global load per chain knot; a group without a problem reads nothing
        lds_publish();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            if (cidx[sl] < NRW) CB[cidx[sl]] = (solve_any && cidx[sl] < M) ? TT[cidx[sl]] : 1.0;
        if (l == 0) CB[NRW] = (solve_any && NRW < M) ? TT[NRW] : 1.0;
        lds_publish();
        auto ldT = [&](int i) -> double { return CB[i]; };
        int kmax = n;                                // longest chain of the wave: uniform loop bounds
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
        kmax = __builtin_amdgcn_readfirstlane(kmax);
        const bool reuse = a.gcache_mode == 2;            // G from the cache of an earlier solve of the same outer loop: no chain
        const int kch = reuse ? 0 : kmax;
        lds_publish();

        // ---------------- forward: block LDL' chain, replicated in the lanes of the group ----------------
        FullBlocks<R> sa;
        sa.build(ldT(0));
        Inv lprev;
        LDLPack<R>::zero(lprev);
#pragma unroll 1
        for (int k = 1; k <= kch; ++k) {
            const bool vk = k <= n;
            FullBlocks<R> sb;
            sb.build(ldT(min(k, M - 1)));
            const double cpl = (vk && k >= 2) ? 1.0 : 0.0;          // coupling X_{k-1} between knots k - 1 and k
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
            // E_{k-1} = S_{k-1}^-1 X_{k-1} = L^-T (D^-1 L^-1 X): back-substitution of Zp
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
                if (corridor_dual_keeps_E(R) && l == 0) {
                    double* const rec = ES(k - 1);
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) rec[NE + i * R + c] = E[i][c];
                }
            }
            Inv ldl;
            ldl.factor(D);
            {   // S_k^-1 (symmetric, lower triangle): columns of the identity through the factors
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
        if (corridor_dual_keeps_E(R) && l == 0 && !reuse) {   // E of the last chain knot: nothing behind it
            double* const rec = ES(kmax);
#pragma unroll
            for (int i = 0; i < R * R; ++i) rec[NE + i] = 0.0;
        }
        lds_publish();

        // ---------------- backward: diagonal blocks of H^-1, the last block column, the owned columns of G ----------------
        double Zk1[R][R], Zkn[R][R], cv[2][R], wv[2][R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            cv[0][i] = 0.0; cv[1][i] = 0.0; wv[0][i] = 0.0; wv[1][i] = 0.0;
#pragma unroll
            for (int c = 0; c < R; ++c) { Zk1[i][c] = 0.0; Zkn[i][c] = 0.0; }
        }
#pragma unroll 1
        for (int k = kch; k >= 1; --k) {
            double Si[R][R], E[R][R];
            {
                const double* const rec = ES(k);
                int f = 0;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c) { Si[i][c] = rec[f]; Si[c][i] = Si[i][c]; ++f; }
                if constexpr (corridor_dual_keeps_E(R)) {
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) E[i][c] = rec[NE + i * R + c];
                } else {
                    // E_k = S_k^-1 X_k, X_k = the start/end block of segment k (it couples knots k and k + 1; none behind the last variable)
                    FullBlocks<R> sx;
                    sx.build(ldT(min(k, M - 1)));
                    const double cplx = (k + 1 <= n) ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            double v = 0.0;
#pragma unroll
                            for (int q = 0; q < R; ++q) v += Si[i][q] * sx.B01[q][c];
                            E[i][c] = v * cplx;
                        }
                }
            }
            lds_publish();    // (the record is in registers before rows of G are written over this and older slots)
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
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int kj = cidx[sl] + 1;
                const double dj = (k == kj) ? 1.0 : 0.0;
                double nv[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    double v = dj * Zkk[i][0];
#pragma unroll
                    for (int q = 0; q < R; ++q) v -= E[i][q] * cv[sl][q];
                    nv[i] = v;
                }
#pragma unroll
                for (int i = 0; i < R; ++i) { cv[sl][i] = nv[i]; wv[sl][i] = fma(dj, Zkn[0][i], wv[sl][i]); }   // w_j = e_0' Z_{j,n}: picked up at k = j + 1
                if (k <= kj && kj <= n) {    // entry (k - 1, j) and its mirror
                    GR(k - 1)[cidx[sl]] = nv[0];
                    GR(cidx[sl])[k - 1] = nv[0];
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int c = 0; c < R; ++c) Zk1[i][c] = Zkk[i][c];
        }
        // rows and columns beyond n hold what the chain records left there: they must be neutral in the sweeps
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            if (cidx[sl] < NRW) {
                double* const row = GR(cidx[sl]);
#pragma unroll 1
                for (int i = cidx[sl] >= n ? 0 : n; i < NRW; ++i) row[i] = 0.0;
            }
        lds_publish();
        // cv[sl] = Z_{1,j} e_0, wv[sl] = e_0' Z_{j,n} of the owned columns
        if (a.gcache_mode != 0 && solve_any) {
            // G and the two vector families across the solves of an outer loop (CorridorArgs::gcache): store after a build, or load and
            // rescale instead of one -- entry ((i, a), (j, b)) of H^-1 scales by s^(2R-1-a-b) when every duration is multiplied by s
            double* const gc = a.gcache + (size_t)b * corridor_gcache_stride;
            if (!reuse) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                    if (cidx[sl] < NRW) {
                        const double* const row = GR(cidx[sl]);
                        for (int i = 0; i < NRW; i += 2) *reinterpret_cast<double2_a*>(gc + cidx[sl] * NRW + i) = *reinterpret_cast<const double2_a*>(row + i);
#pragma unroll
                        for (int q = 0; q < R; ++q) { gc[NRW * NRW + cidx[sl] * 2 * R + q] = cv[sl][q]; gc[NRW * NRW + cidx[sl] * 2 * R + R + q] = wv[sl][q]; }
                    }
            } else {
                const double sc = a.gscale[b];
                double pw[R];                          // pw[q] = s^(2R-1-q)
                pw[R - 1] = sc;
#pragma unroll
                for (int e = 1; e < R; ++e) pw[R - 1] *= sc;     // s^R
#pragma unroll
                for (int q = R - 2; q >= 0; --q) pw[q] = pw[q + 1] * sc;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                    if (cidx[sl] < NRW) {
                        double* const row = GR(cidx[sl]);
                        for (int i = 0; i < NRW; i += 2) {
                            const double2 t = *reinterpret_cast<const double2_a*>(gc + cidx[sl] * NRW + i);
                            *reinterpret_cast<double2_a*>(row + i) = make_double2(t.x * pw[0], t.y * pw[0]);
                        }
#pragma unroll
                        for (int q = 0; q < R; ++q) { cv[sl][q] = gc[NRW * NRW + cidx[sl] * 2 * R + q] * pw[q]; wv[sl][q] = gc[NRW * NRW + cidx[sl] * 2 * R + R + q] * pw[q]; }
                    }
            }
            lds_publish();
        }
        const int nrows = __builtin_amdgcn_readfirstlane(min(NRW, (kmax + 1) & ~1));   // tableau rows the wave touches (even)
#ifdef UAVQP_DUAL_DEBUG
        // per trajectory (dealing position bq < 64): [0, 1024) G row-major [32][32];  [1024 + 96 ax + 32 what + col]: what 0 = p_unc, 1 = trips, 2 = p at the end
        double* const dbg = (a.dbg && have && bq < 64) ? a.dbg + bq * 2048 : nullptr;
        if (dbg) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (cidx[sl] < n)
                    for (int i = 0; i < n; ++i) dbg[i * 32 + cidx[sl]] = GR(cidx[sl])[i];
        }
#endif

        // ---------------- per axis: unconstrained minimiser and boxes of the owned columns (all loads in flight together) ----------------
        double y0[3][2], lo3[3][2], hi3[3][2];
        {
            FullBlocks<R> seg0, segl;
            seg0.build(ldT(0));
            segl.build(ldT(M - 1));
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const bool on = solve_any && (dsc[ax] & 1ull);      // (prep: provisionally every axis; the box check follows the loads)
                const long long base3 = 3LL * ((long long)s0 + b) + ax;
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                double x0[R], xM[R], r1[R], rn[R];
                x0[0] = on ? a.waypoints[base3] : 0.0;
                xM[0] = on ? a.waypoints[base3 + 3LL * M] : 0.0;
#pragma unroll
                for (int d = 0; d < ND; ++d) { x0[d + 1] = on ? bc[d * 3] : 0.0; xM[d + 1] = on ? bc[(ND + d) * 3] : 0.0; }
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
                    const bool vc = on && cidx[sl] < n;
                    const int kk = min(cidx[sl] + 1, M - 1);
                    lo3[ax][sl] = vc ? a.corr_lo[base3 + 3LL * kk] : 0.0;
                    hi3[ax][sl] = vc ? a.corr_hi[base3 + 3LL * kk] : 0.0;
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < R; ++c) v += cv[sl][c] * r1[c] + wv[sl][c] * rn[c];
                    y0[ax][sl] = vc ? v : 0.0;
                }
            }
        }
        if (prep) {
            // box check and equality rows of the three axes from the bounds just loaded: lo <= hi at every interior knot or the trajectory is
            // invalid as a whole (left untouched, like corridor_prep_kernel + the status test of the solve kernel's refill would leave it)
            bool bad = false;
            unsigned long long eqm[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const bool v0 = solve_any && cidx[0] < n, v1 = solve_any && cidx[1] < n;
                bad = bad || (v0 && !(lo3[ax][0] <= hi3[ax][0])) || (v1 && !(lo3[ax][1] <= hi3[ax][1]));
                const unsigned long long e0 = __ballot(v0 && lo3[ax][0] == hi3[ax][0]), e1 = __ballot(v1 && lo3[ax][1] == hi3[ax][1]);
                const unsigned long long gm = (L == 16) ? 0xFFFFull : 0xFFull;
                eqm[ax] = (((e0 >> (grp * L)) & gm) << 1) | (((e1 >> (grp * L)) & gm) << (1 + L));
            }
            const bool box_ok = (__ballot(bad) & gmask) == 0ull;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dsc[ax] = (solve_any && box_ok) ? (eqm[ax] | 1ull) : 0ull;
            if (solve_any && l == 0) {
                a.status[b] = box_ok ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
                if (a.iters) a.iters[b] = 0;
            }
            if (solve_any && l < 3) a.desc[3LL * b + l] = l == 0 ? dsc[0] : (l == 1 ? dsc[1] : dsc[2]);
        }
        lds_publish();
        if (l == 0) { CB[NRW] = 0.0; CB[NRW + 1] = 0.0; CB[CBS + NRW] = 0.0; CB[CBS + NRW + 1] = 0.0; }
        lds_publish();

        // ---------------- the three axes, one after the other for the whole wave ----------------
        // (Groups at their own pace -- a finished group starting its next axis while the others still iterate -- was built first:
        // the hand-over / set-up code then runs once per group and axis instead of once per axis, ~500 instructions each time for
        // the whole wave, which cost more than the idle trips of a group that waits for the slowest of its wave do.)
        // Per owned column: y = the position while the bound is free, MINUS its multiplier while it is in the working set (both move
        // by +t d along a dual step); sw = +1 / -1 while a lower / upper bound is in the working set (0: free, or an equality row).
        double A[2][NRW];
        const int max_trips = 4 * n + 16 + max_trips_extra;
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
        const unsigned long long dcur = axis == 0 ? dsc[0] : (axis == 1 ? dsc[1] : dsc[2]);
        const bool on = solve_any && (dcur & 1ull);
        int q = -1, trips = 0;
        double sdir = 0.0, muq = 0.0;        // direction of the entering constraint, its multiplier so far
        double dg[2], y[2], lo[2], hi[2], tol[2], sw[2] = {0.0, 0.0}, eqb[2];
        bool valid[2], inW[2] = {false, false};
        lds_publish();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            valid[sl] = on && cidx[sl] < n;
            const int kk = min(cidx[sl] + 1, M - 1);
            lo[sl] = axis == 0 ? lo3[0][sl] : (axis == 1 ? lo3[1][sl] : lo3[2][sl]);
            hi[sl] = axis == 0 ? hi3[0][sl] : (axis == 1 ? hi3[1][sl] : hi3[2][sl]);
            y[sl] = axis == 0 ? y0[0][sl] : (axis == 1 ? y0[1][sl] : y0[2][sl]);
            tol[sl] = 1e-12 * (1.0 + fmin(fabs(lo[sl]), fabs(hi[sl])));
            eqb[sl] = (valid[sl] && ((dcur >> kk) & 1ull)) ? 1e300 : 0.0;
            dg[sl] = valid[sl] ? GR(crow[sl])[crow[sl]] : 1.0;
            const double* const row = GR(crow[sl]);     // (rows / columns beyond n are zero)
#pragma unroll
            for (int i = 0; i < NRW; i += 2) {
                const double2 tt = *reinterpret_cast<const double2_a*>(row + i);
                A[sl][i] = tt.x;
                A[sl][i + 1] = tt.y;
            }
        }
#ifdef UAVQP_DUAL_DEBUG
        if (dbg)
            for (int sl = 0; sl < 2; ++sl) if (valid[sl]) dbg[1024 + 96 * axis + cidx[sl]] = y[sl];
#endif
        bool done = !on;
        for (;;) {
            lds_publish();
            // ---- entering constraint: steepest dual ascent, violation^2 / T_qq.  A group without one is done with this axis: every
            // position inside its box (or the trip budget spent: the solve kernel goes on from any set)
            if (__ballot(!done && q < 0) != 0ull) {
                double key = 0.0;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const double below = lo[sl] - y[sl], above = y[sl] - hi[sl];
                    const double v = raw_max(below, above);
                    const bool cand = valid[sl] && !inW[sl] && v > tol[sl] && dg[sl] > 0.0;
                    const double kv = raw_max(raw_min(v * v * __builtin_amdgcn_rcp(dg[sl]), 1e299), eqb[sl]);
                    const double pk = pack_code(kv, (below > above ? 32 : 0) | cidx[sl]);
                    key = raw_max(key, cand ? pk : 0.0);
                }
                key = group_max<L>(key);
                if (!done && q < 0) {
                    if (key > 1e-300 && trips < max_trips) { const int cd = code_of(key); q = cd & 31; sdir = (cd & 32) ? 1.0 : -1.0; muq = 0.0; }
                    else done = true;
                }
            }
            if (__ballot(!done) == 0ull) break;
            const bool go = !done;

            const int qq = go ? q : 0;
            const int lq = qq & (L - 1), slq = qq >> LOG2L;
            const bool own_q = go && (l == lq);
            // ---- direction: column q of the tableau (owner -> LDS -> everyone's own rows), full step length
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
            // ---- first multiplier of the working set to reach zero: mu / d with mu = -y, for bounds whose multiplier moves towards zero
            double rmin = 1e300;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const bool blocks = sw[sl] * d[sl] > 0.0;
                const double ratio = raw_min(raw_max(-y[sl] * rcp1(d[sl]), 0.0), 1e299);
                rmin = raw_min(rmin, blocks ? pack_code(ratio, cidx[sl]) : 1e300);
            }
            rmin = group_min<L>(rmin);
            const bool partial = go && rmin < t1;
            const double t = go ? (partial ? rmin : t1) : 0.0;
            const int kp = partial ? code_of(rmin) & 31 : qq;        // pivot of this trip
            const int lk = kp & (L - 1), slk = kp >> LOG2L;
            const bool own_k = go && (l == lk);
            // ---- step (rows beyond n and lanes beyond the tableau have d = 0)
            y[0] = fma(t, d[0], y[0]);
            y[1] = fma(t, d[1], y[1]);
            muq = fma(sdir, t, muq);
            // ---- sweep on the pivot: the constraint q enters (full step) or the blocking one leaves (partial step)
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
                CB[slk * CBS + kp] = tk - (partial ? -1.0 : 1.0);    // (entering: T_qq > 0; leaving: -[G_WW^-1]_kk < 0)
            }
            lds_publish();
            {
                const double piv = go ? SC[1] : 0.0;
                double s[2];
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const double tc = CB[slk * CBS + crd[sl]];
                    const bool pc = own_k && sl == slk;
                    s[sl] = tc * piv;                                   // (pivot column: (t_k - sign) / t_k = 1 - 1 / |t_k|, by the patch)
                    const double dn = fma(-tc, s[sl], dg[sl]);
                    dg[sl] = pc ? -piv : dn;
                    // the pivot's column changes sides: entering, y = -multiplier; leaving, y = the bound it sat on
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
        // ---- hand the working set of this axis over (bit k = interior knot k, as the solve kernel reads it)
        {
            const unsigned long long bw0 = __ballot(inW[0] && sw[0] != 0.0), bw1 = __ballot(inW[1] && sw[1] != 0.0);
            const unsigned long long bu0 = __ballot(inW[0] && sw[0] < 0.0), bu1 = __ballot(inW[1] && sw[1] < 0.0);
#ifdef UAVQP_DUAL_DEBUG
            if (on && dbg)
                for (int sl = 0; sl < 2; ++sl) if (valid[sl]) { dbg[1024 + 96 * axis + 64 + cidx[sl]] = y[sl]; dbg[1024 + 96 * axis + 32 + cidx[sl]] = (double)trips; }
#endif
            if (on && l == 0) {
                const unsigned long long gm = (L == 16) ? 0xFFFFull : 0xFFull;
                const unsigned long long act = (((bw0 >> (grp * L)) & gm) << 1) | (((bw1 >> (grp * L)) & gm) << (1 + L));
                const unsigned long long upm = (((bu0 >> (grp * L)) & gm) << 1) | (((bu1 >> (grp * L)) & gm) << (1 + L));
                a.guess[2 * (3LL * b + axis)] = act;
                a.guess[2 * (3LL * b + axis) + 1] = upm;
            }
        }
        }
    }
}

template <int R, int L, int NRW>
__global__ __launch_bounds__(64, (NRW <= 24 ? 2 : 1)) void corridor_dual_kernel(CorridorArgs a, int n_lo, int max_trips_extra, int last) {
    __shared__ __attribute__((aligned(16))) double s_all[(64 / L) * corridor_dual_lds_doubles(R, L, NRW)];
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.queue = 0u;      // the work counter of the solve kernel that follows (was a memset node of its own)
    corridor_dual_body<R, L, NRW>(a, n_lo, max_trips_extra, last != 0, s_all, (int)blockIdx.x, (int)gridDim.x);
}

// Batches of mixed lengths (up to 25 segments): ONE launch whose first `split` blocks are the 8-lane groups (trajectories of up to 17
// segments), the others whole DPP rows with 24 tableau rows (18 to 25 segments).  A wave of either shape spends its time in
// dependent chains (one batch of a ragged config-5 solve per wave: the launch lasts as long as ONE batch), so the two shapes
// overlap instead of queueing behind each other as two launches.
constexpr int corridor_dual_mixed_lds(int R) {
    return 8 * corridor_dual_lds_doubles(R, 8, 16) > 4 * corridor_dual_lds_doubles(R, 16, 24) ? 8 * corridor_dual_lds_doubles(R, 8, 16) : 4 * corridor_dual_lds_doubles(R, 16, 24);
}
template <int R>
__global__ __launch_bounds__(64, 2) void corridor_dual_mixed_kernel(CorridorArgs a, int split, int max_trips_extra) {
    __shared__ __attribute__((aligned(16))) double s_all[corridor_dual_mixed_lds(R)];
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.queue = 0u;
    if ((int)blockIdx.x < split) corridor_dual_body<R, 8, 16>(a, 1, max_trips_extra, false, s_all, (int)blockIdx.x, split);
    else corridor_dual_body<R, 16, 24>(a, 17, max_trips_extra, true, s_all, (int)blockIdx.x - split, (int)gridDim.x - split);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// One trajectory per WAVE, G from the cache of an earlier solve of the same outer loop (CorridorArgs::gcache_mode == 2: loaded and
// rescaled, no chain) -- the re-solves of the config-5 pipeline.  The batch kernels above advance eight trajectories in lockstep and
// broadcast the pivot column through LDS: a launch lasts as long as its slowest batch (72-240 us per round for a few hundred to a few
// thousand trajectories, most of the machine idle).  Here lane c owns tableau column c (two 16-row register vectors), the pivot is
// wave-uniform and a trip is the LDS-free one of rows_dual_kernel (own-row symmetry, v_fmac_f64 row_newbcast sweep, permlane
// reductions): no lockstep (a trajectory takes its own 5 exchanges per axis, not the 10-11 of the slowest of eight), no LDS at all,
// G kept in 8 KB of LDS between the axes (the tableau itself lives in registers): four waves per SIMD.  Same duties as corridor_dual_body with prep_in_dual (reset, validation, descriptors, one-segment trajectories),
// same hand-over; as there, nothing here decides a result.
template <int R>
__device__ __forceinline__ double pick_row2(const v16d A0, const v16d A1, int kq) {
    const int e = kq & 15;
    const double x0 = A0[e], x1 = A1[e];
    return kq < 16 ? x0 : x1;
}

template <int R>
__global__ __launch_bounds__(64, 3) void corridor_dual_wave_kernel(CorridorArgs a, int max_trips_extra) {
    constexpr int ND = R - 1, NC = 2 * R;
    __shared__ double s_g[32 * 32];      // [row][column]: this trajectory's G, scaled -- every axis starts its tableau from it
    const int lane = threadIdx.x, c = lane;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.queue = 0u;
    const int n_eff = a.n_active ? *a.n_active : a.n_traj;
#ifdef UAVQP_DUAL_DEBUG
    if (a.dbg && n_eff >= 1500 && n_eff <= 3000 && threadIdx.x == 0 && (blockIdx.x == gridDim.x - 1 || blockIdx.x == 0)) a.dbg[5 * 16384 + (blockIdx.x == 0 ? 2 : 3)] = (double)wall_clock64();   // entry of the first / last block
    if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned slot = atomicAdd(reinterpret_cast<unsigned int*>(a.dbg + 63 * 2048 + 1400 + 8), 1u);
        if (slot < 16) { a.dbg[63 * 2048 + 1400 + 16 + 2 * slot] = (double)wall_clock64(); a.dbg[63 * 2048 + 1400 + 17 + 2 * slot] = (double)n_eff; }
    }
#endif
    for (long long bq = blockIdx.x; bq < n_eff; bq += gridDim.x) {
#ifdef UAVQP_DUAL_DEBUG
        // debug build (tools/pipeline_round_probe.py): per trajectory of a re-solve of 1500-3000 trajectories {trips, cycles, n, entry time}
        const long long dbg_t0 = __builtin_readcyclecounter(), dbg_w0 = wall_clock64();
        int dbg_trips = 0;
#endif
        const int b = __builtin_amdgcn_readfirstlane(a.order ? a.order[bq] : (int)bq);
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        s0 = __builtin_amdgcn_readfirstlane(s0);
        M = __builtin_amdgcn_readfirstlane(M);
        const bool shape_ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;
        const bool fits = shape_ok && M >= 2 && M - 1 <= 24;          // (the cache holds tableaus of up to 24 rows)
        const double* const TT = a.times + s0;
        double Tl = 1.0;
        bool bad = false;
        if (shape_ok && M - 1 <= 24 && lane < M) { Tl = TT[lane]; bad = !((Tl > 0.0) && (Tl < INFINITY)); }
        const bool t_ok = __ballot(bad) == 0ull;
        if (!(fits && t_ok)) {
            // nothing to solve: an invalid trajectory (left untouched) or a single segment (its polynomial follows from the boundary data)
            const bool valid1 = shape_ok && t_ok && M == 1;
            if (lane == 0) {
                a.status[b] = (shape_ok && t_ok) ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
                if (a.iters) a.iters[b] = 0;
            }
            if (lane < 3) {
                a.desc[3LL * b + lane] = 0ull;
                if (valid1 && a.active) { a.active[2 * (3LL * b + lane)] = 0ull; a.active[2 * (3LL * b + lane) + 1] = 0ull; }
                if (valid1) {
                    const long long base3 = 3LL * ((long long)s0 + b) + lane;
                    const double* bc = a.bc + (size_t)b * 2 * ND * 3 + lane;
                    double ys[ND], ye[ND], c1[NC];
#pragma unroll
                    for (int d = 0; d < ND; ++d) { ys[d] = bc[d * 3]; ye[d] = bc[(ND + d) * 3]; }
                    const double Tk = TT[0];
                    segment_coeffs_det<R>(a.waypoints[base3], ys, a.waypoints[base3 + 3], ye, Tk, fast_rcp(Tk), c1);
                    if (!((fabs(c1[NC - 1]) < INFINITY) && (fabs(c1[R]) < INFINITY))) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                    double* o = a.coeff + ((size_t)3 * s0 + lane) * NC;
#pragma unroll
                    for (int j = 0; j < NC; ++j) o[j] = c1[j];
                }
            }
            continue;
        }
        const int n = M - 1;
        const bool vc = c < n;
        const int cc = vc ? c : 0;
        const bool wide = n > 16;                     // the cache was written by the shape that took the trajectory: 16 or 24 rows
        const int NRWs = wide ? 24 : 16;
        const double* const gc = a.gcache + (size_t)b * corridor_gcache_stride;
        double pw[R];                                 // pw[q] = s^(2R-1-q): entry ((i, a), (j, b)) of H^-1 scales by s^(2R-1-a-b)
        {
            const double sc = a.gscale[b];
            pw[R - 1] = sc;
#pragma unroll
            for (int e = 1; e < R; ++e) pw[R - 1] *= sc;
#pragma unroll
            for (int q = R - 2; q >= 0; --q) pw[q] = pw[q + 1] * sc;
        }
        // ---------------- column c of G, its diagonal entry, the two vector families: all loads in flight together ----------------
        double dg0, cv[R], wv[R];
        {
            const double* const rowp = gc + (size_t)cc * NRWs;
            double2 t0[8], t1[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) t0[i] = *reinterpret_cast<const double2_a*>(rowp + 2 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) t1[i] = wide ? *reinterpret_cast<const double2_a*>(rowp + 16 + 2 * i) : make_double2(0.0, 0.0);
            dg0 = rowp[cc];
#pragma unroll
            for (int q = 0; q < R; ++q) { cv[q] = gc[NRWs * NRWs + cc * 2 * R + q]; wv[q] = gc[NRWs * NRWs + cc * 2 * R + R + q]; }
            const double g = vc ? pw[0] : 0.0;
            lds_publish();
            if (c < 32) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { s_g[(2 * i) * 32 + c] = t0[i].x * g; s_g[(2 * i + 1) * 32 + c] = t0[i].y * g; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { s_g[(16 + 2 * i) * 32 + c] = t1[i].x * g; s_g[(17 + 2 * i) * 32 + c] = t1[i].y * g; }
            }
            lds_publish();
            dg0 = vc ? dg0 * pw[0] : 1.0;
#pragma unroll
            for (int q = 0; q < R; ++q) { cv[q] *= pw[q]; wv[q] *= pw[q]; }
        }
        // ---------------- per axis: unconstrained minimiser and box of this lane's knot ----------------
        double y0[3], lo3[3], hi3[3];
        {
            FullBlocks<R> seg0, segl;
            seg0.build(readlane_f64(Tl, 0));
            segl.build(readlane_f64(Tl, M - 1));
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
                    double v1 = 0.0, vn = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) { v1 -= seg0.B01[q][i] * x0[q]; vn -= segl.B01[i][q] * xM[q]; }
                    r1[i] = v1;
                    rn[i] = vn;
                }
                lo3[ax] = vc ? a.corr_lo[base3 + 3LL * (cc + 1)] : 0.0;
                hi3[ax] = vc ? a.corr_hi[base3 + 3LL * (cc + 1)] : 0.0;
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) v += cv[q] * r1[q] + wv[q] * rn[q];
                y0[ax] = vc ? v : 0.0;
            }
        }
        // box check and equality rows (as corridor_prep_kernel would: lo <= hi at every interior knot, or the trajectory is invalid as a whole)
        unsigned long long dsc[3];
        {
            bool badb = false;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                badb = badb || (vc && !(lo3[ax] <= hi3[ax]));
                dsc[ax] = ((__ballot(vc && lo3[ax] == hi3[ax]) & 0xFFFFFFFFull) << 1) | 1ull;
            }
            const bool box_ok = __ballot(badb) == 0ull;
            if (lane == 0) {
                a.status[b] = box_ok ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
                if (a.iters) a.iters[b] = 0;
            }
            if (lane < 3) a.desc[3LL * b + lane] = box_ok ? (lane == 0 ? dsc[0] : (lane == 1 ? dsc[1] : dsc[2])) : 0ull;
            if (!box_ok) continue;
        }
        // ---------------- the three axes ----------------
        const int max_trips = 4 * n + 16 + max_trips_extra;
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = axis == 0 ? lo3[0] : (axis == 1 ? lo3[1] : lo3[2]);
            const double hi = axis == 0 ? hi3[0] : (axis == 1 ? hi3[1] : hi3[2]);
            double y = axis == 0 ? y0[0] : (axis == 1 ? y0[1] : y0[2]);
            const double tol = 1e-12 * (1.0 + fmin(fabs(lo), fabs(hi)));
            const double eqb = (vc && lo == hi) ? 1e300 : 0.0;
            double dg = dg0, sw = 0.0;
            bool inW = false;
            v16d A0, A1;
            {
                const double* const col = s_g + (c & 31);
#pragma unroll
                for (int i = 0; i < 16; ++i) A0[i] = col[i * 32];
#pragma unroll
                for (int i = 0; i < 8; ++i) A1[i] = col[(16 + i) * 32];
#pragma unroll
                for (int i = 8; i < 16; ++i) A1[i] = 0.0;
            }
            int trips = 0;
            for (;;) {
                // entering constraint: steepest dual ascent, violation^2 / T_qq, ranked in single precision (a heuristic choice)
                const double below = lo - y, above = y - hi;
                const double viol = raw_max(below, above);
                const bool cand = vc && !inW && viol > tol && dg > 0.0;
                const float kf = eqb != 0.0 ? 3.0e38f : (float)raw_min(viol * viol * __builtin_amdgcn_rcp(dg), 1e38);
                const unsigned key = wave_umax(cand ? ((__float_as_uint(kf) & ~127u) | (unsigned)((below > above ? 64 : 0) | c)) : 0u);
                const int cd = __builtin_amdgcn_readfirstlane((int)key);
                if (cd < 128 || trips >= max_trips) break;
                const int q = cd & 63;
                const double sdir = (cd & 64) ? 1.0 : -1.0;
                double muq = 0.0;
                double aq = pick_row2<R>(A0, A1, q);
                for (;;) {
                    const double d = sdir * (c == q ? dg : aq);
                    const double pvl = rcp1(dg);
                    const double t1 = readlane_f64(((sdir > 0.0 ? lo : hi) - y) * sdir * pvl, q);
                    const bool blocks = sw * d > 0.0;
                    const double ratio = raw_min(raw_max(-y * rcp1(d), 0.0), 1e299);
                    const double rmin = wave_min64(blocks ? pack_code7(ratio, c) : 1e300);
                    const bool partial = __builtin_amdgcn_readfirstlane((int)(rmin < t1)) != 0;
                    const double t = partial ? rmin : t1;
                    const int kp = partial ? (__builtin_amdgcn_readfirstlane(code7_of(rmin)) & 63) : q;
                    y = fma(t, d, y);
                    muq = fma(sdir, t, muq);
                    // sweep on the pivot kp: the constraint q enters (full step) or the blocking one leaves (partial step)
                    const bool pc = c == kp;
                    const double ak = partial ? pick_row2<R>(A0, A1, kp) : aq;
                    const double tc = pc ? dg - (partial ? -1.0 : 1.0) : ak;
                    const double piv = readlane_f64(pvl, kp);
                    const double scl = tc * piv;
                    const double dn = fma(-tc, scl, dg);
                    dg = pc ? -piv : dn;
                    const double yb = sw < 0.0 ? hi : lo;
                    y = pc ? (partial ? yb : -muq) : y;
                    sw = pc ? ((partial || eqb != 0.0) ? 0.0 : sdir) : sw;
                    inW = pc ? !partial : inW;
                    {
                        double ta, tb, tunused;
                        row_replicas(tc, ta, tb, tunused);
                        const double ns = -scl;
                        sweep16(A0, ta, ns);
                        if (wide) sweep16(A1, tb, ns);
                    }
                    ++trips;
                    if (!partial || trips >= max_trips) break;
                    aq = pick_row2<R>(A0, A1, q);
                }
            }
            // ---- hand the working set of this axis over (bit k = interior knot k, as the solve kernel reads it)
            const unsigned long long bw = __ballot(inW && sw != 0.0), bu = __ballot(inW && sw < 0.0);
            if (lane == 0) {
                a.guess[2 * (3LL * b + axis)] = (bw & 0xFFFFFFFFull) << 1;
                a.guess[2 * (3LL * b + axis) + 1] = (bu & 0xFFFFFFFFull) << 1;
            }
#ifdef UAVQP_DUAL_DEBUG
            dbg_trips += trips;
#endif
        }
#ifdef UAVQP_DUAL_DEBUG
        if (a.dbg && n_eff >= 1500 && n_eff <= 3000 && bq < 16384 && lane == 0) {
            a.dbg[4 * bq] = dbg_trips; a.dbg[4 * bq + 1] = (double)(__builtin_readcyclecounter() - dbg_t0);
            a.dbg[4 * bq + 2] = n; a.dbg[4 * bq + 3] = (double)dbg_w0;
            a.dbg[4 * 16384 + bq] = (double)wall_clock64();
        }
#endif
    }
}


// Two trajectories per wave (lanes 0-31 / 32-63): corridor_dual_wave_kernel leaves half of every wave's lanes idle (a tableau has at most
// 32 columns) and the big re-solves of the pipeline are issue-bound.  Everything that was wave-uniform there is uniform per HALF here
// (a VGPR holding the same value in the 32 lanes of a half); the two halves advance in lockstep through one flat loop with a per-half
// state (select an entering constraint / step / done), as the groups of corridor_dual_body do -- a lockstep of two instead of eight.
// The own-row look-ups are done once per half (two s_set_gpr_idx moves, the lane keeps its half's); the row replicas of the sweep need
// ONE v_permlane16_swap (rows 0, 1 serve the lower half, rows 2, 3 the upper one) and the reductions stop at the half.
__device__ __forceinline__ unsigned half_umax(unsigned v) {
    v = umax_dpp<0xB1>(v);
    v = umax_dpp<0x4E>(v);
    v = umax_dpp<0x141>(v);
    v = umax_dpp<0x140>(v);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return max(r[0], r[1]);
}
__device__ __forceinline__ double half_min64(double v) {
    v = raw_min(v, dpp_f64<0xB1>(v));
    v = raw_min(v, dpp_f64<0x4E>(v));
    v = raw_min(v, dpp_f64<0x141>(v));
    v = raw_min(v, dpp_f64<0x140>(v));
    double p, q;
    cross_rows(v, p, q);
    return raw_min(p, q);
}

template <int R>
__global__ __launch_bounds__(64, 2) void corridor_dual_wave2_kernel(CorridorArgs a, int max_trips_extra) {
    constexpr int ND = R - 1, NC = 2 * R;
    const int lane = threadIdx.x, h = lane >> 5, c = lane & 31;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.queue = 0u;
    const int n_eff = a.n_active ? *a.n_active : a.n_traj;
    // (fewer trajectories than waves: one per wave, the upper half idle -- a lone trajectory does not wait for a partner's trips)
    const bool single = n_eff <= (int)gridDim.x;
    const long long n_pairs = single ? (long long)n_eff : ((long long)n_eff + 1) / 2;
    for (long long pr = blockIdx.x; pr < n_pairs; pr += gridDim.x) {
        const long long bq = single ? pr : 2 * pr + h;
        const bool have = bq < n_eff && !(single && h == 1);
        const int b = have ? (a.order ? a.order[bq] : (int)bq) : 0;
        int s0 = 0, M = 0;
        if (have) { if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; } }
        const bool shape_ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;
        const bool fits = shape_ok && M >= 2 && M - 1 <= 24;
        const double* const TT = a.times + s0;
        double Tl = 1.0;
        bool bad = false;
        if (have && shape_ok && M - 1 <= 24 && c < M) { Tl = TT[c]; bad = !((Tl > 0.0) && (Tl < INFINITY)); }
        const unsigned long long hm = 0xFFFFFFFFull << (32 * h);
        const bool t_ok = (__ballot(bad) & hm) == 0ull;
        const bool solve = have && fits && t_ok;
        if (have && !solve) {
            // nothing to solve: an invalid trajectory (left untouched) or a single segment (its polynomial follows from the boundary data)
            const bool valid1 = shape_ok && t_ok && M == 1;
            if (c == 0) {
                a.status[b] = (shape_ok && t_ok) ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
                if (a.iters) a.iters[b] = 0;
            }
            if (c < 3) {
                a.desc[3LL * b + c] = 0ull;
                if (valid1 && a.active) { a.active[2 * (3LL * b + c)] = 0ull; a.active[2 * (3LL * b + c) + 1] = 0ull; }
                if (valid1) {
                    const long long base3 = 3LL * ((long long)s0 + b) + c;
                    const double* bc = a.bc + (size_t)b * 2 * ND * 3 + c;
                    double ys[ND], ye[ND], c1[NC];
#pragma unroll
                    for (int d = 0; d < ND; ++d) { ys[d] = bc[d * 3]; ye[d] = bc[(ND + d) * 3]; }
                    const double Tk = TT[0];
                    segment_coeffs_det<R>(a.waypoints[base3], ys, a.waypoints[base3 + 3], ye, Tk, fast_rcp(Tk), c1);
                    if (!((fabs(c1[NC - 1]) < INFINITY) && (fabs(c1[R]) < INFINITY))) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                    double* o = a.coeff + ((size_t)3 * s0 + c) * NC;
#pragma unroll
                    for (int j = 0; j < NC; ++j) o[j] = c1[j];
                }
            }
        }
        if (__ballot(solve) == 0ull) continue;
        const int Ms = solve ? M : 2;                 // (keeps every index of a half without a problem in range)
        const int n = Ms - 1;
        const bool vc = solve && c < n;
        const int cc = vc ? c : 0;
        const bool wide = n > 16;                     // the cache was written by the shape that took the trajectory: 16 or 24 rows
        const bool any_wide = __ballot(solve && wide) != 0ull;
        const int NRWs = wide ? 24 : 16;
        const double* const gc = a.gcache + (size_t)b * corridor_gcache_stride;
        double pw[R];
        {
            const double sc = solve ? a.gscale[b] : 1.0;
            pw[R - 1] = sc;
#pragma unroll
            for (int e = 1; e < R; ++e) pw[R - 1] *= sc;
#pragma unroll
            for (int q = R - 2; q >= 0; --q) pw[q] = pw[q + 1] * sc;
        }
        v16d G0, G1;
        double dg0, cv[R], wv[R];
        {
            const double* const rowp = gc + (size_t)cc * NRWs;
            double2 t0[8], t1[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) t0[i] = solve ? *reinterpret_cast<const double2_a*>(rowp + 2 * i) : make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < 4; ++i) t1[i] = (solve && wide) ? *reinterpret_cast<const double2_a*>(rowp + 16 + 2 * i) : make_double2(0.0, 0.0);
            dg0 = solve ? rowp[cc] : 1.0;
#pragma unroll
            for (int q = 0; q < R; ++q) { cv[q] = solve ? gc[NRWs * NRWs + cc * 2 * R + q] : 0.0; wv[q] = solve ? gc[NRWs * NRWs + cc * 2 * R + R + q] : 0.0; }
            const double g = vc ? pw[0] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { G0[2 * i] = t0[i].x * g; G0[2 * i + 1] = t0[i].y * g; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { G1[2 * i] = t1[i].x * g; G1[2 * i + 1] = t1[i].y * g; }
#pragma unroll
            for (int i = 8; i < 16; ++i) G1[i] = 0.0;
            dg0 = vc ? dg0 * pw[0] : 1.0;
#pragma unroll
            for (int q = 0; q < R; ++q) { cv[q] *= pw[q]; wv[q] *= pw[q]; }
        }
        // ---------------- per axis: unconstrained minimiser and box of this lane's knot ----------------
        double y0[3], lo3[3], hi3[3];
        {
            // (durations 0 and M - 1 of the own half: every lane of a half reads them from that half's lanes)
            const int l0 = 32 * h, lM = 32 * h + Ms - 1;
            const double T0 = __hiloint2double(__shfl(__double2hiint(Tl), l0, 64), __shfl(__double2loint(Tl), l0, 64));
            const double TM = __hiloint2double(__shfl(__double2hiint(Tl), lM, 64), __shfl(__double2loint(Tl), lM, 64));
            FullBlocks<R> seg0, segl;
            seg0.build(solve ? T0 : 1.0);
            segl.build(solve ? TM : 1.0);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const long long base3 = 3LL * ((long long)s0 + b) + ax;
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                double x0[R], xM[R], r1[R], rn[R];
                x0[0] = solve ? a.waypoints[base3] : 0.0;
                xM[0] = solve ? a.waypoints[base3 + 3LL * Ms] : 0.0;
#pragma unroll
                for (int d = 0; d < ND; ++d) { x0[d + 1] = solve ? bc[d * 3] : 0.0; xM[d + 1] = solve ? bc[(ND + d) * 3] : 0.0; }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    double v1 = 0.0, vn = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) { v1 -= seg0.B01[q][i] * x0[q]; vn -= segl.B01[i][q] * xM[q]; }
                    r1[i] = v1;
                    rn[i] = vn;
                }
                lo3[ax] = vc ? a.corr_lo[base3 + 3LL * (cc + 1)] : 0.0;
                hi3[ax] = vc ? a.corr_hi[base3 + 3LL * (cc + 1)] : 0.0;
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) v += cv[q] * r1[q] + wv[q] * rn[q];
                y0[ax] = vc ? v : 0.0;
            }
        }
        // box check and equality rows, per half
        bool box_ok;
        {
            bool badb = false;
            unsigned long long dsc[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                badb = badb || (vc && !(lo3[ax] <= hi3[ax]));
                dsc[ax] = (((__ballot(vc && lo3[ax] == hi3[ax]) >> (32 * h)) & 0xFFFFFFFFull) << 1) | 1ull;
            }
            box_ok = (__ballot(badb) & hm) == 0ull;
            if (solve && c == 0) {
                a.status[b] = box_ok ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
                if (a.iters) a.iters[b] = 0;
            }
            if (solve && c < 3) a.desc[3LL * b + c] = box_ok ? (c == 0 ? dsc[0] : (c == 1 ? dsc[1] : dsc[2])) : 0ull;
        }
        const bool run = solve && box_ok;
        // ---------------- the three axes: both halves in lockstep ----------------
        const int max_trips = 4 * n + 16 + max_trips_extra;
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = axis == 0 ? lo3[0] : (axis == 1 ? lo3[1] : lo3[2]);
            const double hi = axis == 0 ? hi3[0] : (axis == 1 ? hi3[1] : hi3[2]);
            double y = axis == 0 ? y0[0] : (axis == 1 ? y0[1] : y0[2]);
            const double tol = 1e-12 * (1.0 + fmin(fabs(lo), fabs(hi)));
            const double eqb = (vc && lo == hi) ? 1e300 : 0.0;
            double dg = dg0, sw = 0.0, sdir = 0.0, muq = 0.0;
            bool inW = false, done = !run;
            int q = -1, trips = 0;
            v16d A0 = G0, A1 = G1;
            for (;;) {
                // ---- entering constraint of the halves that need one: steepest dual ascent, ranked in single precision
                if (__ballot(!done && q < 0) != 0ull) {
                    const double below = lo - y, above = y - hi;
                    const double viol = raw_max(below, above);
                    const bool cand = vc && !inW && viol > tol && dg > 0.0;
                    const float kf = eqb != 0.0 ? 3.0e38f : (float)raw_min(viol * viol * __builtin_amdgcn_rcp(dg), 1e38);
                    const int cd = (int)half_umax(cand ? ((__float_as_uint(kf) & ~127u) | (unsigned)((below > above ? 64 : 0) | c)) : 0u);
                    if (!done && q < 0) {
                        if (cd >= 128 && trips < max_trips) { q = cd & 63; sdir = (cd & 64) ? 1.0 : -1.0; muq = 0.0; }
                        else done = true;
                    }
                }
                if (__ballot(!done) == 0ull) break;
                const bool go = !done;
                const int qq = go ? q : 0;
                const int qA = __builtin_amdgcn_readlane(qq, 0), qB = __builtin_amdgcn_readlane(qq, 32);
                const double aqA = pick_row2<R>(A0, A1, qA), aqB = pick_row2<R>(A0, A1, qB);
                const double aq = h ? aqB : aqA;
                const double d = go ? sdir * (c == qq ? dg : aq) : 0.0;
                const double pvl = rcp1(dg);
                const double t1l = ((sdir > 0.0 ? lo : hi) - y) * sdir * pvl;
                const double t1A = readlane_f64(t1l, qA), t1B = readlane_f64(t1l, 32 + qB);
                const double t1 = h ? t1B : t1A;
                const bool blocks = sw * d > 0.0;
                const double ratio = raw_min(raw_max(-y * rcp1(d), 0.0), 1e299);
                const double rmin = half_min64(blocks ? pack_code7(ratio, c) : 1e300);
                const bool partial = go && rmin < t1;
                const double t = go ? (partial ? rmin : t1) : 0.0;
                const int kp = partial ? (code7_of(rmin) & 63) : qq;
                y = fma(t, d, y);
                muq = fma(sdir, t, muq);
                // ---- sweep on the pivot: the constraint q enters (full step) or the blocking one leaves (partial step)
                const int kA = __builtin_amdgcn_readlane(kp, 0), kB = __builtin_amdgcn_readlane(kp, 32);
                const double akA = pick_row2<R>(A0, A1, kA), akB = pick_row2<R>(A0, A1, kB);
                const double ak = h ? akB : akA;
                const bool pc = go && c == kp;
                const double tc = go ? (pc ? dg - (partial ? -1.0 : 1.0) : ak) : 0.0;
                const double pvA = readlane_f64(pvl, kA), pvB = readlane_f64(pvl, 32 + kB);
                const double piv = go ? (h ? pvB : pvA) : 0.0;
                const double scl = tc * piv;
                const double dn = fma(-tc, scl, dg);
                dg = pc ? -piv : dn;
                const double yb = sw < 0.0 ? hi : lo;
                y = pc ? (partial ? yb : -muq) : y;
                sw = pc ? ((partial || eqb != 0.0) ? 0.0 : sdir) : sw;
                inW = pc ? !partial : inW;
                {
                    double ta, tb;
                    cross_rows(tc, ta, tb);          // [r0 r0 r2 r2], [r1 r1 r3 r3]: rows 0-15 / 16-31 of each half's own t
                    const double ns = -scl;
                    sweep16(A0, ta, ns);
                    if (any_wide) sweep16(A1, tb, ns);
                }
                if (go) {
                    if (!partial) q = -1;
                    ++trips;
                }
            }
            // ---- hand the working set of this axis over (bit k = interior knot k, as the solve kernel reads it)
            const unsigned long long bw = __ballot(inW && sw != 0.0), bu = __ballot(inW && sw < 0.0);
            if (run && c == 0) {
                a.guess[2 * (3LL * b + axis)] = ((bw >> (32 * h)) & 0xFFFFFFFFull) << 1;
                a.guess[2 * (3LL * b + axis) + 1] = ((bu >> (32 * h)) & 0xFFFFFFFFull) << 1;
            }
        }
    }
}

}  // namespace uavqp
