// qp_rows_dual.h -- the starting working set of the general-rows solve (qp_rows2.h) from the dual active-set method of
// qp_corridor_dual.h, extended from knot boxes to rows lo <= p_i^(d)(tau T_i) <= hi (round 4).
//
// A row is a linear functional of the Hermite states of the two end knots of its segment, c' x = g_l' x_i + g_r' x_{i+1}
// (row_functional, qp_rows.h); a knot box is the functional e_0' x_k.  The dual method only needs the dense matrix
// G_ij = c_i' H^-1 c_j over all constraints and their unconstrained values c_j' H^-1 rhs.  Both come from the chain of
// qp_corridor_dual.h: with z_j = H^-1 c_j, the components of z_j at its own knots and at the knots before them follow the back-substitution
//     z^(k) = dR Z_kk (g_r - E_{k-1}' g_l) + dL S_k^-1 g_l - E_k z^(k+1)        (dR: k = i + 1, dL: k = i; nothing is computed behind i + 1)
// and G_ij = g_l,i' z_j^(k_i) + g_r,i' z_j^(k_i + 1) is taken for i <= j in constraint order (constraints are numbered along the knots; the
// other half by symmetry: only the lower triangle of G is stored).  This is the QP of rows_pair_kernel itself, not a relaxation: the set the dual method ends with is that
// QP's working set.  (The first version refined the time grid by a knot at every row and bounded components of the inserted knots --
// a relaxation, an inserted knot with an active row may break the higher derivatives: 3.65 verifying solves mean instead of 1.)
// BASELINE config 3 with K = 2 rows per segment: 15 knots, 47 constraints per axis, ~7 active at the solution, ~9 exchanges.
// As in qp_corridor_dual.h nothing here decides a result: the set goes to rows_pair_kernel as its starting working set.
//
// One trajectory per wave: lane c owns tableau column c (48 rows: three 16-row register vectors), lane s also prepares segment s
// (functionals of its rows); every branch of the dual loop is wave-uniform and a trip touches no LDS (own-row symmetry, v_fmac_f64
// row_newbcast sweep, permlane reductions: see the loop).  The 3 x 3 chain quantities, the same for all lanes, come from
// rows_chain_kernel (one LANE per trajectory) through HBM and an LDS-DMA copy.  Handled: at most 48 constraints ((M - 1) + used rows),
// M <= 32; anything else is left to the box phase (need_phase1) and starts the rows solve from the box set as before.
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
    unsigned int* n_phase1;          // their number (zeroed by the host)
    const double* gfun;              // [segment][K][2 R]: g_l, g_r of every row (rows_prep_kernel, launched before this kernel)
    const double* kdF;               // the chain records of rows_chain_kernel (launched before this kernel), one array per half, knot-major:
    const double* kdB;               // half h of knot k of trajectory b at kd_h + (k - 1) * kd_plane + b * rows_chain_half_mem(R)
    long long kd_plane;
    unsigned int* ticket;            // dealing counter (zeroed before the launch; null: wave w takes trajectories w, w + grid, ...)
#ifdef UAVQP_DUAL_DEBUG
    double* dbg;
#endif
};

// debug build: cycles per section of the wave that solves one of the first 16 trajectories -> dbg[3100 + k]
// (k: 0 prologue, 1 forward chain, 2 backward pass + G, 3 per-axis set-up, 4 selection, 5 direction + ratio test, 6 sweep, 7 hand-over; 8 = trips)
#ifdef UAVQP_DUAL_DEBUG
#define RD_T_DECL long long rd_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; long long rd_t = __builtin_readcyclecounter();
#define RD_T(k) do { const long long n_ = __builtin_readcyclecounter(); rd_acc[k] += n_ - rd_t; rd_t = n_; } while (0)
#else
#define RD_T_DECL
#define RD_T(k) do {} while (0)
#endif

// G (lower triangle), chain records of knots 1..31 (two halves of rows_chain_half_mem doubles: the DMA's 16-byte pieces), functionals, durations,
// int tables (the masks handed over alias the functionals, which are dead by then).  r = 3: 2554 doubles = 20 432 bytes, eight of them fit the 160 KB of a CU.
constexpr int rows_dual_lds_doubles(int R) { return 48 * 49 / 2 + 31 * 2 * ((R * (R + 1) / 2 + R * R + 1) & ~1) + 48 * 2 * R + 34 + 64; }

// What the prelude needs of the block LDL' chain of a trajectory is the same in all 64 lanes of the wave that solves it (one trajectory
// per wave: 48 columns): computing it THERE repeats every 3 x 3 recursion 64 times -- a third of the kernel's instructions.
// rows_chain_kernel computes it once, one LANE per trajectory (forward: S_k^-1 and E_{k-1}; backward: Z_kk = S_k^-1 + E_k Z_{k+1,k+1} E_k' and the
// last block column Z_kn).  Per knot k = 1..M-1 a record has two halves, {S_k^-1 (lower triangle), E_{k-1}} and {Z_kk (lower triangle), Z_kn};
// rows_dual_kernel copies the records of its trajectory into LDS by DMA.
// Layout in HBM (round 5): two arrays, one per half, KNOT-major -- half h of knot k of trajectory b at kd_h + (k - 1) * plane + b * rows_chain_half_mem(R)
// -- so that what the 64 lanes of a wave produce for one knot is ONE contiguous run (r = 3: 64 x 128 bytes): the lanes hand their values over
// through LDS and the wave stores whole lines; a half is padded to 16 bytes so that the prelude's DMA moves it in 16-byte pieces.
// Was: trajectory-major records, every lane writing its own 16 bytes at a time 3840 bytes from its neighbour's -- each store instruction 64
// partial lines, read back the same way: the kernel sat at the memory system's request rate (0.54 GB moved, 165-172 us; more waves per SIMD
// made it slower, asking for its loads a knot ahead changed nothing).  Now 65 us.  (Complete records written trajectory-major by the backward
// pass, 15 lanes per record: 111 us; halves in knot-major planes but packed, fetched by the prelude dword-wise: prelude + 60 us.)
constexpr int rows_chain_doubles(int R) { return 2 * (R * (R + 1) / 2) + 2 * R * R; }      // a whole record (the LDS layout of rows_dual_kernel)
constexpr int rows_chain_half(int R) { return R * (R + 1) / 2 + R * R; }
constexpr int rows_chain_half_mem(int R) { return (rows_chain_half(R) + 1) & ~1; }         // padded to 16 bytes (r = 3: 16 doubles = one 128-byte line)

template <int R>
__global__ __launch_bounds__(64) void rows_chain_kernel(RowsArgs a, double* __restrict__ kdF, double* __restrict__ kdB, long long plane) {
    constexpr int NH = rows_chain_half(R), NHM = rows_chain_half_mem(R);
    using Inv = SmallLDL<R>;
    __shared__ __attribute__((aligned(16))) double s_t[64 * NHM];
    const int lane = threadIdx.x;
    const long long n_waves = ((long long)a.n_traj + 63) / 64;
    // the wave's rows of one knot: LDS (row = lane) <-> one contiguous run of HBM, 16 bytes per lane and instruction
    auto rows_out = [&](double* dst, const double (&v)[NH]) __attribute__((always_inline)) {
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < NH; ++q) s_t[lane * NHM + q] = v[q];
        if (NHM > NH) s_t[lane * NHM + NH] = 0.0;
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < NHM / 2; ++p) {
            const int i = p * 64 + lane;
            *reinterpret_cast<double2_a*>(dst + 2 * i) = *reinterpret_cast<const double2_a*>(s_t + 2 * i);
        }
    };
    auto rows_in = [&](const double* src, double (&v)[NH]) __attribute__((always_inline)) {
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < NHM / 2; ++p) {
            const int i = p * 64 + lane;
            *reinterpret_cast<double2_a*>(s_t + 2 * i) = *reinterpret_cast<const double2_a*>(src + 2 * i);
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < NH; ++q) v[q] = s_t[lane * NHM + q];
    };
    for (long long wv = blockIdx.x; wv < n_waves; wv += gridDim.x) {
        const long long b0 = wv * 64;
        const bool present = b0 + lane < a.n_traj;
        const int b = present ? (int)(b0 + lane) : 0;
        int s0 = 0, M = 0;
        if (present) {
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        }
        const bool taken = present && M >= 2 && M <= 32 && (a.uniform > 0 || M <= a.max_segments);      // (a trajectory the prelude takes)
        const int n = taken ? M - 1 : 0;
        int nmax = n;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));

        const double* const T = a.times + s0;
        FullBlocks<R> sa;
        sa.build(taken ? T[0] : 1.0);
        Inv lprev;
        LDLPack<R>::zero(lprev);
#pragma unroll 1
        for (int k = 1; k <= nmax; ++k) {
            double fw[NH];
#pragma unroll
            for (int q = 0; q < NH; ++q) fw[q] = 0.0;
            if (k <= n) {
                FullBlocks<R> sb;
                sb.build(T[k]);
                double D[R][R], Yp[R][R], Zp[R][R], E[R][R];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) { D[i][q] = sa.B11[i][q] + sb.B00(i, q); E[i][q] = 0.0; }
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
                if (k >= 2) {    // E_{k-1} = S_{k-1}^-1 X_{k-1} = L^-T (D^-1 L^-1 X): back-substitution of Zp
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
                }
                Inv ldl;
                ldl.factor(D);
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
                int f = 0;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int cc = 0; cc <= i; ++cc) fw[f++] = Si[i][cc];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int cc = 0; cc < R; ++cc) fw[f++] = E[i][cc];
                lprev = ldl;
                sa = sb;
            }
            rows_out(kdF + (size_t)(k - 1) * plane + (size_t)b0 * NHM, fw);
        }
        double Zk1[R][R], Zkn[R][R], Ek[R][R];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int q = 0; q < R; ++q) { Zk1[i][q] = 0.0; Zkn[i][q] = 0.0; Ek[i][q] = 0.0; }
#pragma unroll 1
        for (int k = nmax; k >= 1; --k) {
            double fw[NH], bw[NH];
            rows_in(kdF + (size_t)(k - 1) * plane + (size_t)b0 * NHM, fw);
#pragma unroll
            for (int q = 0; q < NH; ++q) bw[q] = 0.0;
            if (k <= n) {
                double Si[R][R], Em[R][R];
                int f = 0;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q <= i; ++q) { Si[i][q] = fw[f]; Si[q][i] = fw[f]; ++f; }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) Em[i][q] = fw[f++];
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
                f = 0;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q <= i; ++q) bw[f++] = Zkk[i][q];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) bw[f++] = Zkn[i][q];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int q = 0; q < R; ++q) { Zk1[i][q] = Zkk[i][q]; Ek[i][q] = Em[i][q]; }
            }
            rows_out(kdB + (size_t)(k - 1) * plane + (size_t)b0 * NHM, bw);
        }
    }
}

template <int R, int K>
__global__ __launch_bounds__(64, 2) void rows_dual_kernel(RowsDualArgs aa, int max_trips_extra) {
    const RowsArgs& a = aa.r;
    // G is symmetric: only its lower triangle is kept (entry (i, c) at max (max + 1) / 2 + min; 9.4 KB instead of 18.8), next to the chain
    // records -- 16.4 KB per wave in all, so that the register file (256 VGPRs: two waves per SIMD), not LDS, sets the 8 waves per CU.
    // (First version: full rows with the records aliased underneath, 22 KB: 7 waves per CU, one SIMD of four with a single wave.)
    constexpr int ND = R - 1, NRW = 48, NE = R * (R + 1) / 2, HB = rows_chain_half_mem(R), RS = 2 * HB;    // (record in LDS: the second half starts at HB)
    constexpr int O_ER = NRW * (NRW + 1) / 2, O_GF = O_ER + 31 * RS, O_TB = O_GF + NRW * 2 * R, O_IT = O_TB + 34, NMK = 2 + 2 * K;
    static_assert(O_ER % 2 == 0 && O_GF % 2 == 0 && O_TB % 2 == 0 && O_IT % 2 == 0, "16-byte functionals, 8-byte tables on even offsets");
    __shared__ __attribute__((aligned(16))) double sg[rows_dual_lds_doubles(R)];
    const int lane = threadIdx.x, c = lane;
    double* const GF = sg + O_GF;      // [48][2 R]: g_l, g_r of every constraint (a box: e_0, 0)
    double* const TB = sg + O_TB;      // [33]: durations
    // [3][2 + 2 K] masks handed over, per axis: on top of the functionals -- nothing reads those after the backward pass
    unsigned long long* const MK = reinterpret_cast<unsigned long long*>(sg + O_GF);
    static_assert(3 * NMK <= NRW * 2 * R, "the masks of the three axes fit the functionals' space");
    int* const KT = reinterpret_cast<int*>(sg + O_IT);       // [34] per knot: first constraint that sits there | count << 8
    int* const CD = KT + 34;                                   // [48] per constraint: left knot | kind << 8 (0 box, 1 + slot) | segment << 12
    int* const CNT = CD + 48;                                  // [33] rows per segment
    auto ES = [&](int k) -> double* { return sg + O_ER + (k - 1) * RS; };      // chain record of knot k = 1..31: S_k^-1, E_{k-1}, Z_kk, Z_kn
    auto GP = [&](int i, int j) -> double& { const int hi = max(i, j), lo = min(i, j); return sg[hi * (hi + 1) / 2 + lo]; };
    const int crow = min(c, NRW - 1);

    // Dealing (round 6): the first trajectory of a wave is its block index, every further one a TICKET drawn from a counter.  The exchanges a
    // trajectory needs vary (config 3 + K = 2: 27 +- 8 over the three axes); with the round-robin of rounds 4-5 every wave summed 32 of them and the
    // launch lasted as long as the unluckiest of 2048 sums -- PMC: the mean wave was resident 83 % of the kernel's cycles.  The ticket is asked for behind
    // the last load of a trajectory (in front of the three axes' exchanges: no wait for a load also waits for the atomic) and read at its end.
    unsigned int tk = 0;
    for (long long bq = blockIdx.x; bq < a.n_traj;
         bq = aa.ticket ? (long long)gridDim.x + (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)tk) : bq + (long long)gridDim.x) {
        auto draw = [&]() __attribute__((always_inline)) { if (aa.ticket && lane == 0) tk = atomicAdd(aa.ticket, 1u); };
        const int b = aa.order ? aa.order[bq] : (int)bq;
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        s0 = __builtin_amdgcn_readfirstlane(s0);
        M = __builtin_amdgcn_readfirstlane(M);
        const bool shape_ok = M >= 2 && M <= 32 && (a.uniform > 0 || M <= a.max_segments);
        if (!shape_ok) { if (lane == 0) { aa.need_phase1[b] = 1; atomicAdd(aa.n_phase1, 1u); } draw(); continue; }
        const int n = M - 1;
        RD_T_DECL
        lds_publish();
        // the chain records of this trajectory (rows_chain_kernel): HBM -> LDS by DMA, 16 bytes per lane and instruction; they land while the
        // functionals are prepared
        {
            // (16-byte piece pp of the trajectory's n records = piece `w` of knot kk + 1: the first HB / 2 of a record come from the array of
            // the forward halves, the others from that of the backward halves)
            const int pieces = n * (RS / 2);
            for (int p0 = 0; p0 < pieces; p0 += 64) {
                const int pp = p0 + lane;
                if (pp < pieces) {
                    const int kk = pp / (RS / 2), w = pp - kk * (RS / 2), hf = w >= HB / 2 ? 1 : 0;
                    const double* const src = (hf ? aa.kdB : aa.kdF) + (size_t)kk * aa.kd_plane + (size_t)b * HB + 2 * (w - hf * (HB / 2));
                    __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(sg + O_ER + 2 * p0), 16, 0, 0);
                }
            }
        }
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
        if (!handled) { if (lane == 0) { aa.need_phase1[b] = 1; atomicAdd(aa.n_phase1, 1u); } wait_vmcnt0(); draw(); continue; }
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

        RD_T(0);
        wait_vmcnt0();      // the chain records (DMA issued at the top) have landed
        lds_publish();
        RD_T(1);

        // ---------------- backward: z_c = H^-1 c_c at and above its knots, the entries of G that do not sit behind it ----------------
        const bool vc = c < NC;
        const int cdc = vc ? CD[crow] : 0;
        const int kLc = vc ? (cdc & 255) : -100;       // left knot of this column's functional (0: the boundary knot)
        double gLc[R], gRc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { gLc[q] = vc ? GF[crow * 2 * R + q] : 0.0; gRc[q] = vc ? GF[crow * 2 * R + R + q] : 0.0; }
        const int tri_c = crow * (crow + 1) / 2;       // row c of the packed lower triangle
        double Ek[R][R], v[R], wv[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i] = 0.0; wv[i] = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) Ek[i][q] = 0.0;
        }
        // One knot of the backward pass.  Its LDS operands -- the chain record (S_k^-1, E_{k-1}, Z_kk, Z_kn), the knot's constraint table entry
        // and the functionals of its first K + 1 constraints -- are loaded one knot ahead (two register sets, the loop below alternates
        // between them): the compiler cannot lift a load above the G entries written just before it, and every exposed LDS round trip was
        // ~10 % on top of an issue-bound phase.
        // Round 6: what a column INJECTS into the recursion it injects at its own two knots only -- k_R = k_L + 1 (the functional's right knot) and k_L:
        //     z^(k) = [k = k_R] Z_kk (g_r - E_{k-1}' g_l) + [k = k_L] S_k^-1 g_l - E_k z^(k+1),      e_0' Z_{.,n} part: Z_{k_R,n}' g_r + Z_{k_L,n}' g_l
        // -- so both injections (u1, u2) and the whole last-block-column vector wv are made ONCE per lane from the records of ITS two knots (lane-private
        // LDS reads), and a knot of the loop is the 9 FMAs of -E_k z plus two selects, not 45 FMAs of which 36 multiply the zeros of the other 13 knots
        // (backward pass + G: 130 -> 90 instructions per knot; the loop needs E_{k-1} of its knot and nothing else of the record).
        constexpr bool PF = R == 3;     // (r = 4: the second register set does not fit)
        double u1[R], u2[R];
        {
            const bool hasR = vc && kLc + 1 >= 1 && kLc + 1 <= n, hasL = vc && kLc >= 1 && kLc <= n;
            const double* const recR = ES(hasR ? kLc + 1 : 1);
            const double* const recL = ES(hasL ? kLc : 1);
            double tR[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double t = gRc[i];
#pragma unroll
                for (int p_ = 0; p_ < R; ++p_) t -= recR[NE + p_ * R + i] * gLc[p_];      // g_r - E_{k_R - 1}' g_l   (E_0 = 0)
                tR[i] = t;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double a1 = 0.0, a2 = 0.0, w = 0.0;
#pragma unroll
                for (int p_ = 0; p_ < R; ++p_) {
                    const int hi_ = i > p_ ? i : p_, lo_ = i > p_ ? p_ : i, f_ = hi_ * (hi_ + 1) / 2 + lo_;       // packed lower triangle of a symmetric block
                    a1 += recR[HB + f_] * tR[p_];                     // Z_kk (k_R)
                    a2 += recL[f_] * gLc[p_];                         // S_k^-1 (k_L)
                    w = fma(recR[HB + NE + p_ * R + i], hasR ? gRc[p_] : 0.0, w);      // Z_kn (k_R)' g_r
                    w = fma(recL[HB + NE + p_ * R + i], hasL ? gLc[p_] : 0.0, w);      // Z_kn (k_L)' g_l
                }
                u1[i] = hasR ? a1 : 0.0;
                u2[i] = hasL ? a2 : 0.0;
                wv[i] = w;
            }
        }
        struct KnotOps {
            double Em[R][R];
            int kt;
        };
        auto load_ops = [&](int k, KnotOps& o) __attribute__((always_inline)) {
            const double* const rec = ES(k);
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int q = 0; q < R; ++q) o.Em[i][q] = rec[NE + i * R + q];
            o.kt = KT[k];
        };
        auto knot = [&](int k, const KnotOps& o, KnotOps& nx) __attribute__((always_inline)) {
            if (PF && k >= 2) load_ops(k - 1, nx);
            const int kt = __builtin_amdgcn_readfirstlane(o.kt);
            const int cf = kt & 255, cnt = (kt >> 8) & 255;
            int cdv[K + 1];
            double gfv[K + 1][2 * R];
            // (from knot 2 on the first constraint of a knot is its BOX -- functional e_0 on this knot: its entry of G is the position component of
            //  z^(k) itself, no functional to load, no dot product; knot 1 may start with the rows of segment 0)
            const bool box0 = k >= 2;
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                const int i = min(cf + t, NRW - 1);
                cdv[t] = CD[i];
                if (t == 0 && box0) {
#pragma unroll
                    for (int p = 0; p < 2 * R; ++p) gfv[t][p] = 0.0;
                } else {
#pragma unroll
                    for (int p = 0; p < 2 * R; p += 2) {
                        const double2 g2 = *reinterpret_cast<const double2_a*>(GF + i * 2 * R + p);
                        gfv[t][p] = g2.x; gfv[t][p + 1] = g2.y;
                    }
                }
            }
            lds_publish();
            // this lane's column
            double vn[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double t = (k == kLc + 1) ? u1[i] : ((k == kLc) ? u2[i] : 0.0);
#pragma unroll
                for (int p = 0; p < R; ++p) t -= Ek[i][p] * v[p];
                vn[i] = t;
            }
            // entries of G: constraints that sit at knot k against every column that does not sit in front of them
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                if (t < cnt) {
                    const int i = cf + t;
                    const int kLi = cdv[t] & 255;
                    double val = 0.0;
                    if (t == 0 && box0) val = vn[0];
                    else {
#pragma unroll
                        for (int p = 0; p < R; ++p) val += kLi == k ? (gfv[t][p] * vn[p] + gfv[t][R + p] * v[p]) : gfv[t][R + p] * vn[p];   // (kLi = 0 at k = 1: only its right knot is a variable)
                    }
                    if (vc && i <= c) sg[tri_c + i] = val;      // (constraints are numbered along the knots: i <= c never sits behind c)
                }
            }
            for (int t = K + 1; t < cnt; ++t) {      // (knot 1 also hosts the rows of segment 0)
                const int i = cf + t;
                const int kLi = CD[i] & 255;
                double val = 0.0;
#pragma unroll
                for (int p = 0; p < R; ++p) {
                    const double ga = GF[i * 2 * R + p], gb = GF[i * 2 * R + R + p];
                    val += kLi == k ? (ga * vn[p] + gb * v[p]) : gb * vn[p];
                }
                if (vc && i <= c) sg[tri_c + i] = val;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                v[i] = vn[i];
#pragma unroll
                for (int q = 0; q < R; ++q) Ek[i][q] = o.Em[i][q];
            }
        };
        if constexpr (PF) {
            KnotOps oa, ob;
            load_ops(n, oa);
#pragma unroll 1
            for (int k = n; k >= 1; k -= 2) {
                knot(k, oa, ob);
                if (k >= 2) knot(k - 1, ob, oa);
            }
        } else {
#pragma unroll 1
            for (int k = n; k >= 1; --k) {
                KnotOps oa;
                load_ops(k, oa);
                knot(k, oa, oa);
            }
        }
        // rows and columns beyond the constraints must be neutral in the sweeps
        if (c < NRW) {
#pragma unroll 1
            for (int i = max(NC, c); i < NRW; ++i) sg[i * (i + 1) / 2 + c] = 0.0;
        }
        lds_publish();
        RD_T(2);

        // ---------------- per axis: unconstrained value and bounds of this lane's constraint ----------------
        double y0[3], lo3[3], hi3[3];
        {
            FullBlocks<R> seg0, segl;
            seg0.build(TB[0]);
            segl.build(TB[M - 1]);
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
                for (int i = 0; i < NC; ++i) dbg[i * 48 + c] = GP(c, i);
                dbg[2304 + 576 + c] = (double)cdc;
            }
            if (lane == 0) { dbg[2304 + 640] = NC; dbg[2304 + 641] = n; }
        }
#endif

        // ---------------- the three axes: the dual method, every branch wave-uniform ----------------
        // The tableau column of this lane: three 16-row register vectors (contiguous registers: a row picked by a wave-uniform run-time
        // index is one s_set_gpr_idx move).  The tableau stays symmetric under the sweeps, so everything a trip needs of the pivot's
        // COLUMN is the pivot's ROW entry every lane already holds -- no column travels through LDS:
        //   * direction d_c = T[q][c] and pivot-row entry t_c = T[k][c]: own_row(q), own_row(k);
        //   * the sweep's u_i = T[i][k] = t_i of lane i: three replicas of t (columns 0-15 / 16-31 / 32-47 copied into all four DPP
        //     rows by two permlane swaps each) feed v_fmac_f64 ... row_newbcast:i -- ONE instruction per tableau row and lane.
        // (first version: the owner wrote its column to LDS, 24 ds_write_b128 + 24 broadcast ds_read_b128 per trip and wave; with 7
        // waves per CU the LDS pipe was ~45 % busy and a trip cost 4.5 k cycles: tools/rows_dual_gpu_probe.py stamps)
        v16d A0, A1, A2;
#define own_row(kq) pick_row(A0, A1, A2, (kq))
        const int max_trips = 4 * NC + 16 + max_trips_extra;
        if (lane < 3 * NMK) MK[lane] = 0ull;      // (behind the lds_publish that ends the backward pass; the axis loop publishes before it ORs into them)
        draw();
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = axis == 0 ? lo3[0] : (axis == 1 ? lo3[1] : lo3[2]);
            const double hi = axis == 0 ? hi3[0] : (axis == 1 ? hi3[1] : hi3[2]);
            double y = axis == 0 ? y0[0] : (axis == 1 ? y0[1] : y0[2]);
            const double tol = 1e-12 * (1.0 + fmin(fabs(lo), fabs(hi)));
            const double eqb = (vc && lo == hi) ? 1e300 : 0.0;
            double dg = vc ? GP(crow, crow) : 1.0, sw = 0.0;
            bool inW = false;
            lds_publish();
            {
                // column c of G (lanes 48..63 carry a copy of column 47: nothing ever reads them)
                const int tri = crow * (crow + 1) / 2;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    A0[i] = sg[i >= crow ? i * (i + 1) / 2 + crow : tri + i];
                    A1[i] = sg[16 + i >= crow ? (16 + i) * (17 + i) / 2 + crow : tri + 16 + i];
                    A2[i] = sg[32 + i >= crow ? (32 + i) * (33 + i) / 2 + crow : tri + 32 + i];
                }
            }
#ifdef UAVQP_DUAL_DEBUG
            if (dbg && vc) dbg[2304 + 192 * axis + c] = y;
#endif
            int trips = 0;
            RD_T(3);
            for (;;) {
                // entering constraint: steepest dual ascent, violation^2 / T_qq
                const double below = lo - y, above = y - hi;
                const double viol = raw_max(below, above);
                const bool cand = vc && !inW && viol > tol && dg > 0.0;
                // (the choice among the violated constraints is a heuristic: ranked in single precision, 17 bits of it, equality rows first)
                const float kf = eqb != 0.0 ? 3.0e38f : (float)raw_min(viol * viol * __builtin_amdgcn_rcp(dg), 1e38);
                const unsigned key = wave_umax(cand ? ((__float_as_uint(kf) & ~127u) | (unsigned)((below > above ? 64 : 0) | c)) : 0u);
                const int cd = __builtin_amdgcn_readfirstlane((int)key);
                if (cd < 128 || trips >= max_trips) break;
                const int q = cd & 63;
                const double sdir = (cd & 64) ? 1.0 : -1.0;
                double muq = 0.0;
                RD_T(4);
                // the constraint moves towards its bound until it reaches it (it enters) -- each time a multiplier of the working set would
                // change sign first, that constraint leaves and the move goes on
                double aq = own_row(q);
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
                    RD_T(5);
                    // sweep on the pivot kp: the constraint q enters (full step) or the blocking one leaves (partial step)
                    const bool pc = c == kp;
                    const double ak = partial ? own_row(kp) : aq;
                    const double tc = pc ? dg - (partial ? -1.0 : 1.0) : ak;      // (entering: T_qq > 0; leaving: -[G_WW^-1]_kk < 0)
                    const double piv = readlane_f64(pvl, kp);
                    const double sc = tc * piv;                                    // (pivot column: (t_k - sign) / t_k = 1 - 1 / |t_k|, by the patch)
                    const double dn = fma(-tc, sc, dg);
                    dg = pc ? -piv : dn;
                    const double yb = sw < 0.0 ? hi : lo;
                    y = pc ? (partial ? yb : -muq) : y;
                    sw = pc ? ((partial || eqb != 0.0) ? 0.0 : sdir) : sw;
                    inW = pc ? !partial : inW;
                    {
                        double ta, tb, tcc;
                        row_replicas(tc, ta, tb, tcc);
                        const double ns = -sc;
                        sweep16(A0, ta, ns);
                        sweep16(A1, tb, ns);
                        sweep16(A2, tcc, ns);
                    }
                    ++trips;
                    RD_T(6);
                    if (!partial || trips >= max_trips) break;
                    aq = own_row(q);
                }
            }
            RD_T(4);
#ifdef UAVQP_DUAL_DEBUG
            rd_acc[8] += trips;
#endif
            // ---- the working set of this axis in the rows kernel's layout: boxes by interior knot, rows by slot and segment (LDS; stored for the three
            // axes at once below)
            if (vc && inW && sw != 0.0) {
                const int kind = (cdc >> 8) & 15, seg = (cdc >> 12) & 255;
                const int word = axis * NMK + (kind == 0 ? 0 : 2 * kind);
                const unsigned long long bit = 1ull << (kind == 0 ? kLc : seg);
                atomicOr(&MK[word], bit);
                if (sw < 0.0) atomicOr(&MK[word + 1], bit);
            }
#ifdef UAVQP_DUAL_DEBUG
            if (dbg && vc) { dbg[2304 + 192 * axis + 96 + c] = y; dbg[2304 + 192 * axis + 48 + c] = (double)trips; }
#endif
            RD_T(7);
        }
        // hand-over: the masks of the three axes are contiguous per trajectory in both arrays -- 48 bytes of boxes, 48 K bytes of rows: two store
        // instructions per trajectory.  (Rounds 4-5: lane 0 stored 2 + 2 K words per axis, eighteen 8-byte stores each a write transaction of its own:
        // 114 MB of counted write traffic for 9.4 MB of masks.)
        lds_publish();
#ifdef UAVQP_DUAL_DEBUG
        if (dbg && lane == 0) for (int ax_ = 0; ax_ < 3; ++ax_) for (int j = 0; j < NMK; ++j) dbg[2304 + 700 + 8 * ax_ + j] = (double)MK[ax_ * NMK + j];
#endif
        if (lane < 6) aa.warm_box[6 * (size_t)b + lane] = MK[(lane >> 1) * NMK + (lane & 1)];
        if (lane < 6 * K) aa.warm_rows[6 * K * (size_t)b + lane] = MK[(lane / (2 * K)) * NMK + 2 + lane % (2 * K)];
#ifdef UAVQP_DUAL_DEBUG
        if (dbg && lane == 0) for (int k_ = 0; k_ < 9; ++k_) dbg[3100 + k_] = (double)rd_acc[k_];
#endif
    }
}

}  // namespace uavqp
