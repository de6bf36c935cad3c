// qp_rows2.h -- general inequality rows (qp_rows.h) on the corridor kernel's architecture: TWO lanes per (trajectory, axis) problem,
// sweep state in LDS, persistent waves with a work counter.
//
// qp_rows.h ran one lane per problem with the sweep state of every knot in an HBM workspace: 56 GB of traffic per config-3 + K = 2
// dispatch for 0.37 GB of problem (150x), HBM-bound at 2.3 TB/s, 25 ms.  Same method here (exact dual active set, every solve one
// block-Thomas pass with blocks [x_k ; mu(rows of the segment the block closes)] of size R + K), same decisions, but:
//   * lane L eliminates the own knots 1..m_L-1 of the trajectory forward, lane R those of the TIME-REVERSED problem (derivative d
//     picks up (-1)^d; a row functional g_l' x_k + g_r' x_{k+1} becomes (F g_r)' x'_start + (F g_l)' x'_end) with the same code; the
//     rows of a segment ride in the block of the knot that closes it IN THE LANE'S OWN DIRECTION, so every row has exactly one owner
//     (L: segments 0..c-1, R: c..M-1, c = ceil(M/2) the meeting knot);
//   * the meeting block [x_c ; mu(segment c-1) ; mu(segment c)] (size R + 2 K) is assembled from the two partial blocks exchanged
//     through DPP, in ONE frame and ONE ordering, so both lanes solve bit-identical systems;
//   * per own knot the LDL' factors of the block and its solution (F = B (B + 1) / 2 + B doubles) live in LDS -- 2 waves per CU,
//     80 KiB each: 8 knots for r = 3, K = 2, i.e. the whole state of a 16-segment problem; longer halves keep their far knots in the
//     HBM workspace (per-lane branch); only the dual state (current / new multiplier per constraint) goes through HBM;
//   * decisions (most violated constraint, first multiplier to reach zero, singular working set) are made per half and combined
//     across the pair with order-independent tie rules (largest violation / smallest step, then kind, then index), so which lane
//     sees a constraint never matters;
//   * validation and the permanent (equality) masks come from rows_prep_kernel, one lane per problem, off the solver's path.
// Checked against qp_rows.h's kernel (uavqp_settings.rows_lanes_per_problem = 1; tests/test_gpu_rows.py::test_pair_and_one_lane_rows_kernels_take_the_same_path) and the exact-rational fixtures incl. working sets.
#pragma once
#include "qp_rows.h"

#ifdef UAVQP_ROWS2_TIMING   // probe build (tools/rows_sections.py): cycles of wave 0 per section -> the queue block, bytes 64..
#define R2_CT_DECL long long r2_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long r2_t = __builtin_readcyclecounter();
#define R2_CT(k) do { const long long n_ = __builtin_readcyclecounter(); r2_acc[k] += n_ - r2_t; r2_t = n_; } while (0)
#define R2_CT_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) reinterpret_cast<long long*>(a.queue)[8 + k_] = r2_acc[k_]; } while (0)
#else
#define R2_CT_DECL
#define R2_CT(k) do {} while (0)
#define R2_CT_FLUSH do {} while (0)
#endif

namespace uavqp {

#ifndef UAVQP_ROWS2_LDS_KB
#define UAVQP_ROWS2_LDS_KB 80   // LDS per single-wave workgroup: 80 KiB = two waves per CU with the whole state of a 16-segment problem on chip
#endif
// Round 6: the LDS of the VERIFYING pass is its own knob.  Measured on config 3 + K = 2 (docs/measurement_log.md R6.1): 80 / 53 / 40 / 32 / 26 KiB per wave
// = 2 / 3 / 4 / 5 / 6 waves per CU: 426 / 405 / 364 / 373 / 397 us -- the kernel is bound by the dependent issue of a single wave per SIMD, and at 40 KiB
// every SIMD has one; but the sweep records that no longer fit go through the HBM workspace: + 0.9 GB of counter traffic (2.05 -> 2.98 GB per step) for
// - 0.08 ms (2.17 -> 2.09 ms).  Default: everything on chip (80); -DUAVQP_ROWS2_VER_LDS_KB=40 buys the time with the bytes.
#ifndef UAVQP_ROWS2_VER_LDS_KB
#define UAVQP_ROWS2_VER_LDS_KB 40
#endif
// ... and the records that 40 KiB no longer hold stay ON CHIP all the same: the verifying pass keeps the UAVQP_ROWS2_VER_REG_KNOTS own knots farthest from
// the meeting knot in REGISTERS (the kernel runs one wave per SIMD: 512 registers per lane, 336 in use).  Both sweeps iterate over the slot index
// s = m - j, which is the same for every lane of the wave at any moment (a lane with a shorter half starts its forward sweep later and ends its
// backward sweep earlier), so "slot s of the register file" is a scalar branch on a wave-uniform s, never an indexed register access.  Four waves per CU
// (every SIMD busy) with the whole state of a 16-segment problem on chip: no workspace traffic.  0: LDS (+ HBM workspace) only.
#ifndef UAVQP_ROWS2_VER_REG_KNOTS
#define UAVQP_ROWS2_VER_REG_KNOTS 4
#endif
constexpr int rows2_reg_knots(bool ver) { return ver ? UAVQP_ROWS2_VER_REG_KNOTS : 0; }
constexpr int rows2_lds_kb(bool ver) { return ver ? UAVQP_ROWS2_VER_LDS_KB : UAVQP_ROWS2_LDS_KB; }
constexpr int rows2_waves_per_cu(bool ver = false) { return 160 / rows2_lds_kb(ver); }
constexpr int rows2_lds_knots(int R, int K, bool ver = false) { return (rows2_lds_kb(ver) * 1024) / (64 * 8 * ((R + K) * (R + K + 1) / 2 + (R + K))); }

struct Rows2Args {
    RowsArgs r;
    unsigned long long* desc;   // [problem][2 + 2 K]: {valid | M >= 2 flag, eqmask, rused[K], req[K]} from rows_prep_kernel
    unsigned int* redo;         // [problems]: the (trajectory, axis) problems the first pass leaves to the second (count: queue[1]; tickets of the second: queue[2])
    const int32_t* order;       // dealing order of the trajectories (longest first), may be null
    double* lam;                // dual state [wave][own knot 0..kown][2 (1 + K)][lane]: current / new multiplier of the knot box and the rows
    double* gfun;               // row functionals [segment][K][2 R] (g_l, g_r of p^(d)(tau T) = g_l' x_k + g_r' x_{k+1}, ORIGINAL frame), made once per
                                // solve by rows_prep_kernel: they depend on the time allocation only (not on the axis, not on the working set)
    int ws_knots;               // own knots per lane kept in the HBM workspace (beyond the LDS slots)
    int lam_knots;              // own knots per lane in `lam` (= own segments per lane in `gfun`)
    double* coeff;              // round 6: the pair kernel writes the polynomials of a finished problem itself (segment_coeffs_det on its sweep state, as
                                // corridor_solve_kernel does since round 4); null: the Hermite solution goes to RowsArgs::xsol for corridor_emit_kernel
    // round 6: rows_prep_kernel is the FIRST kernel of the step and also (i) makes the row functionals (was: rows_gfun_kernel, a second pass over
    // the same rows), (ii) initialises status / iteration counts (was: fill_i32_kernel + a memset) and (iii) clears what the dual prelude only
    // writes for the trajectories it takes (was: two memsets) -- any of these may be null
    int prep_gfun;              // 1: write `gfun`
    unsigned long long* init_warm_box;    // [n_traj][3][2]  -> 0
    unsigned long long* init_warm_rows;   // [n_traj][3][2 K] -> 0
    unsigned int* init_counters;          // 128 words -> 0 (block 0): the prelude's counter block and the pair kernel's queue behind it
};

// validation + permanent masks.  desc[0]: bit 0 = valid, bit 1 = has a free knot (M >= 2), bits 8.. = M; desc[1] = knot boxes with
// lo == hi; then per row slot: used mask, equality mask (bit = ORIGINAL segment).
// One trajectory per wave, lane s = segment s (and interior knot s), all three axes: every array is read in contiguous runs and the
// masks are ballots.  (Round 2-4: one lane per (trajectory, axis) walking its own 1.3 KB of strided inputs -- 0.77 GB of fetch for
// 0.24 GB of inputs on config 3 + K = 2, 132 us.)
template <int R, int K, int TW = 1>
__global__ __launch_bounds__(256) void rows_prep_kernel(Rows2Args aa) {
    // TW = 2 (round 6; the host picks it when no trajectory has more than 32 segments): TWO trajectories per wave, lanes 0-31 / 32-63 -- with one per
    // wave a 16-segment trajectory keeps 16 lanes busy in the validation and 32 in the functionals, and the kernel is bound by the instructions it
    // issues (config 3 + K = 2: 100 us for 0.31 GB); every ballot below is read through the half's 32-bit window.
    static_assert(TW == 1 || TW == 2, "one or two trajectories per wave");
    constexpr int HL = 64 / TW;                                     // lanes per trajectory
    const RowsArgs& a = aa.r;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int half = TW == 2 ? lane >> 5 : 0, hl = TW == 2 ? lane & 31 : lane;
    const unsigned long long hmask = TW == 2 ? 0xFFFFFFFFull : ~0ull;
    auto hballot = [&](bool x) -> unsigned long long { return (__ballot(x) >> (TW == 2 ? 32 * half : 0)) & hmask; };
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long n_units = ((long long)a.n_traj + TW - 1) / TW;
    if (aa.init_counters && blockIdx.x == 0 && threadIdx.x < 128) aa.init_counters[threadIdx.x] = 0u;
    for (long long bq = (long long)blockIdx.x * (blockDim.x >> 6) + wib; bq < n_units; bq += n_waves) {
        const bool present = TW * bq + half < a.n_traj;
        const int b = present ? (int)(TW * bq + half) : 0;
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        const bool shape_ok = present && (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= HL - 1;
        const bool seg = shape_ok && hl < M;                    // this lane has a segment
        // Every load of the trajectory is issued HERE, before anything is decided (round 6): the kernel moves 0.31 GB and was bound by the four
        // memory round trips a wave made one after the other (durations -> ballot -> knot boxes -> rows -> the functionals' inputs; PMC: VALU
        // active 5 % of the wave cycles, waiting 54 %).  What a load brings in is only used if the checks that used to guard it pass.
        double t_seg = 1.0;
        if (seg) t_seg = a.times[s0 + hl];
        const bool kn = shape_ok && hl >= 1 && hl < M;          // lane k = interior knot k = 1..M-1, the three axes are 24 contiguous bytes per array
        double kl[3] = {0.0, 0.0, 0.0}, kh[3] = {0.0, 0.0, 0.0};
        if (kn) {
            const long long at = 3LL * ((long long)s0 + b + hl);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                kl[ax] = a.corr_lo ? a.corr_lo[at + ax] : a.waypoints[at + ax];
                kh[ax] = a.corr_hi ? a.corr_hi[at + ax] : a.waypoints[at + ax];
            }
        }
        int rd[K];
        double rtau[K], rl[K][3], rh[K][3];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            rd[j] = -1; rtau[j] = 0.0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { rl[j][ax] = 0.0; rh[j][ax] = 0.0; }
            if (seg) {
                const size_t e = (size_t)(s0 + hl) * K + j;
                rd[j] = a.row_deriv[e];
                rtau[j] = a.row_tau[e];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) { rl[j][ax] = a.row_lo[e * 3 + ax]; rh[j][ax] = a.row_hi[e * 3 + ax]; }
            }
        }
        // (the functionals' first pass: one (segment, row slot) per lane -- 32 of them per half-wave, one trip for up to 16 segments with K = 2)
        const int mk = (aa.prep_gfun && shape_ok) ? M * K : 0;
        int gd = -1;
        double gtau = 0.0, gtq = 1.0;
        if (hl < mk) {
            const size_t e = (size_t)s0 * K + hl;
            gd = a.row_deriv[e];
            gtau = a.row_tau[e];
            gtq = a.times[s0 + hl / K];
        }

        const bool t_bad = seg && !((t_seg > 0.0) && (t_seg < INFINITY));
        const bool t_ok = shape_ok && hballot(t_bad) == 0ull;
        // first kernel of the step: the outputs every later kernel only lowers / raises / ORs into start here
        if (present && hl == 0 && a.iters) a.iters[b] = 0;
        if (present && aa.init_warm_box && hl < 6) aa.init_warm_box[(size_t)b * 6 + hl] = 0ull;
        if (present && aa.init_warm_rows && hl < 6 * K) aa.init_warm_rows[(size_t)b * 6 * K + hl] = 0ull;
        const bool knot = t_ok && kn;
        bool kbad[3] = {false, false, false}, keq[3] = {false, false, false};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            kbad[ax] = knot && !(kl[ax] <= kh[ax]);
            keq[ax] = knot && kl[ax] == kh[ax];
        }
        // rows of segment `hl`
        bool rused[K], rbad_any[K], rbad[K][3], req[K][3];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            rused[j] = false; rbad_any[j] = false;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { rbad[j][ax] = false; req[j][ax] = false; }
        }
        if (aa.prep_gfun) {
            // the row functionals of this trajectory (they depend on the time allocation only: the same for the three axes, for the dual
            // prelude and for every solve of the rows kernel): one (segment, row slot) per lane and trip -- contiguous loads and stores, every
            // lane busy (lane = segment would leave three quarters of the wave idle in the longest part of this kernel); zeros for an unused or
            // invalid row
            int mk_max = mk;
            if (TW == 2) mk_max = max(mk_max, __shfl_xor(mk_max, 32, 64));
            for (int e0 = 0; e0 < mk_max; e0 += HL) {
                const int el = e0 + hl;
                if (el < mk) {
                    const size_t e = (size_t)s0 * K + el;
                    int d = gd;
                    double tau = gtau, tq = gtq;
                    if (e0 > 0) { d = a.row_deriv[e]; tau = a.row_tau[e]; tq = a.times[s0 + el / K]; }      // (more than HL rows: a round trip per further trip)
                    double gl[R], gr[R];
#pragma unroll
                    for (int c = 0; c < R; ++c) { gl[c] = 0.0; gr[c] = 0.0; }
                    if (d >= 0 && d < R && tq > 0.0 && tq < INFINITY && tau >= 0.0 && tau < 1.0) {
                        // (one straight-line instance per derivative order: with d a run-time value the factorial and power loops of row_functional are
                        //  some thirty divergent little loops per call)
                        switch (d) {
                            case 0: row_functional<R>(tq, tau, 0, gl, gr); break;
                            case 1: row_functional<R>(tq, tau, 1, gl, gr); break;
                            case 2: row_functional<R>(tq, tau, 2, gl, gr); break;
                            default: row_functional<R>(tq, tau, R - 1, gl, gr); break;
                        }
                    }
                    double* o = aa.gfun + e * 2 * R;
#pragma unroll
                    for (int c = 0; c < R; ++c) { o[c] = gl[c]; o[R + c] = gr[c]; }
                }
            }
        }
        if (t_ok && hl < M) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int d = rd[j];
                if (d < 0) continue;
                const double tau = rtau[j];
                rused[j] = true;
                rbad_any[j] = !((d < R) && (tau >= 0.0) && (tau < 1.0) && !(tau == 0.0 && d == 0));
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const double l = rl[j][ax], h = rh[j][ax];
                    rbad[j][ax] = !(l <= h);
                    req[j][ax] = l == h;
                }
            }
        }
        unsigned long long used[K], anybad = 0ull;
#pragma unroll
        for (int j = 0; j < K; ++j) { used[j] = hballot(rused[j]); anybad |= hballot(rbad_any[j]); }
        bool all_ok = true;      // (uniform per trajectory: ballots)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            unsigned long long bad = anybad | hballot(kbad[ax]);
            const unsigned long long eq = hballot(keq[ax]);
            unsigned long long rq[K];
#pragma unroll
            for (int j = 0; j < K; ++j) { bad |= hballot(rbad[j][ax]); rq[j] = hballot(req[j][ax]); }
            const bool ok = t_ok && bad == 0ull;
            all_ok = all_ok && ok;
            if (present && hl == ax) {
                unsigned long long* o = aa.desc + ((size_t)3 * b + ax) * (2 + 2 * K);
                o[0] = ok ? (1ull | (M >= 2 ? 2ull : 0ull) | ((unsigned long long)M << 8)) : 0ull;
                o[1] = eq;
#pragma unroll
                for (int j = 0; j < K; ++j) { o[2 + 2 * j] = used[j]; o[3 + 2 * j] = rq[j]; }
            }
        }
        // the trajectory's status starts here (the first kernel of the step: ONE store, nothing to lower yet)
        if (present && hl == 0) a.status[b] = all_ok ? (int32_t)UAVQP_SOLVED : (int32_t)UAVQP_INVALID_INPUT;
    }
}

// GI = false: the FIRST pass over all problems.  A problem whose working set goes singular (the entering constraint depends on it, or a
// singular starting set) is not decided here: it is put on the redo list (queue[1] = count, aa.redo) with no output written, and the
// second launch, GI = true, solves the listed problems from their start with Goldfarb-Idnani's zero-primal-step route, the restart
// and the infeasibility certificate (qp_rows.h's header).  So the pass that sees every problem carries none of that machinery --
// selects on a mode flag in the sweeps, the entering functional, the certificate's bookkeeping: measured +10 % on config 3 + K = 2,
// where not one of 196 608 problems needs it -- and the second launch finds an empty list (a few microseconds).
//
// VER = true (round 6): the VERIFYING pass of a step whose starting sets come from the dual prelude (qp_rows_dual.h) -- on BASELINE config 3 +
// K = 2 every one of the 196 608 problems is confirmed by its first block solve.  That solve needs no dual state: nothing was in the set
// before (every multiplier starts at 0), so the pass neither reads nor writes `lam` (the HBM round trip of 2 (1 + K) doubles per own knot and
// lane was 0.3 GB of the kernel's 0.7 GB); a problem the first solve does not confirm -- a wrong-signed multiplier, a violated constraint, a
// singular set -- goes to the redo list and is solved from its start by the GI = true launch, which keeps the general machinery and takes
// exactly the path the general first pass would have taken.
// aa.coeff != null (round 6): a finished problem turns the Hermite states of its sweep slots into the polynomials of its own segments
// (segment_coeffs_det, as corridor_solve_kernel does since round 4: bit-identical to corridor_emit_kernel on the same numbers) -- through LDS
// and out as whole lines when the wave finishes together (always, in the verifying pass), straight from the lanes otherwise.
template <int R, int K, bool WS, bool GI, bool VER = false>
__global__ __launch_bounds__(64, 1) void rows_pair_kernel(Rows2Args aa) {
    static_assert(!(VER && GI), "the verifying pass leaves everything beyond one solve to the GI = true launch");
    const RowsArgs& a = aa.r;
    constexpr int ND = R - 1, B = R + K, NL = B * (B + 1) / 2, NCN = 1 + K, F = NL + B, BM = R + 2 * K;
    constexpr int NT = rows2_lds_knots(R, K, VER);
    constexpr int NRS = (VER && !WS) ? rows2_reg_knots(true) : 0;      // slots NT .. NT + NRS - 1 live in registers (kernels without workspace slots only)
    constexpr int NONE = 1 << 30;
    __shared__ __attribute__((aligned(16))) double s_rec[NT * F * 64];
    const int lane = threadIdx.x;
    const int isR = lane & 1;
    // state slot s = own knot m - s (slot 0: the meeting knot).  LDS for s < NT, else the HBM workspace (per-lane branch: the lanes
    // of a wave may sit at different slots)
    double* const ws = a.ws + (size_t)blockIdx.x * (size_t)(aa.ws_knots > 0 ? aa.ws_knots : 1) * F * 64 + lane;
    auto RL = [&](int s, int f) -> double& { return s_rec[(s * F + f) * 64 + lane]; };
    auto RG = [&](int s, int f) -> double& { return ws[((size_t)(s - NT) * F + f) * 64]; };
    auto rec_ld = [&](int s, int f) -> double { return (!WS || s < NT) ? RL(s, f) : RG(s, f); };
    auto rec_st = [&](int s, int f, double v) { if (!WS || s < NT) RL(s, f) = v; else RG(s, f) = v; };
    // register slots (NRS > 0): rr[q] = the record of slot NT + q; `s` below is wave-uniform wherever these are used
    double rr[NRS > 0 ? NRS : 1][F];
    auto slot_st = [&](int s, const double (&e)[NL], const double (&h)[B]) __attribute__((always_inline)) {
        bool in_reg = false;
        if constexpr (NRS > 0) {
#pragma unroll
            for (int q = 0; q < NRS; ++q)
                if (s == NT + q) {
#pragma unroll
                    for (int i = 0; i < NL; ++i) rr[q][i] = e[i];
#pragma unroll
                    for (int i = 0; i < B; ++i) rr[q][NL + i] = h[i];
                    in_reg = true;
                }
        }
        if (!in_reg) {
#pragma unroll
            for (int i = 0; i < NL; ++i) rec_st(s, i, e[i]);
#pragma unroll
            for (int i = 0; i < B; ++i) rec_st(s, NL + i, h[i]);
        }
    };
    auto slot_ld = [&](int s, double (&e)[NL], double (&h)[B]) __attribute__((always_inline)) {
        bool in_reg = false;
        if constexpr (NRS > 0) {
#pragma unroll
            for (int q = 0; q < NRS; ++q)
                if (s == NT + q) {
#pragma unroll
                    for (int i = 0; i < NL; ++i) e[i] = rr[q][i];
#pragma unroll
                    for (int i = 0; i < B; ++i) h[i] = rr[q][NL + i];
                    in_reg = true;
                }
        }
        if (!in_reg) {
#pragma unroll
            for (int i = 0; i < NL; ++i) e[i] = rec_ld(s, i);
#pragma unroll
            for (int i = 0; i < B; ++i) h[i] = rec_ld(s, NL + i);
        }
    };
    auto slot_st_y = [&](int s, const double (&y)[B]) __attribute__((always_inline)) {
        bool in_reg = false;
        if constexpr (NRS > 0) {
#pragma unroll
            for (int q = 0; q < NRS; ++q)
                if (s == NT + q) {
#pragma unroll
                    for (int i = 0; i < B; ++i) rr[q][NL + i] = y[i];
                    in_reg = true;
                }
        }
        if (!in_reg) {
#pragma unroll
            for (int i = 0; i < B; ++i) rec_st(s, NL + i, y[i]);
        }
    };
    // dual state: slot s, constraint c (0: the knot box, 1 + j: row slot j of the segment the block closes), cur / new
    double* const lamb = aa.lam + (size_t)blockIdx.x * (size_t)aa.lam_knots * 2 * NCN * 64 + lane;
    auto LC = [&](int s, int c) -> double& { return lamb[((size_t)s * 2 * NCN + c) * 64]; };
    auto LN = [&](int s, int c) -> double& { return lamb[((size_t)s * 2 * NCN + NCN + c) * 64]; };

    const long long total = GI ? (long long)a.queue[1] : (long long)a.n_traj * 3;    // (GI: the redo list of the first pass)
    unsigned int* const ticket = GI ? a.queue + 2 : a.queue;
    bool queue_empty = false;
    bool redo = false;           // !GI: this problem goes to the second pass
    R2_CT_DECL

    // ---- problem state (identical in both lanes of a pair unless noted)
    long long g = -1;
    int b = 0, ax = 0, s0 = 0, M = 0, m = 0;   // m: own knots up to and including the meeting knot (lane-specific for odd M)
    long long base3 = 0;
    unsigned long long eqmask = 0ull, pin = 0ull, upper = 0ull;
    unsigned long long rused[K], req[K], ract[K], rup[K], pract[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { rused[j] = 0ull; req[j] = 0ull; ract[j] = 0ull; rup[j] = 0ull; pract[j] = 0ull; }
    unsigned long long ppin = 0ull;
    double x0[R];   // own boundary knot, own frame (lane-specific)
#pragma unroll
    for (int c = 0; c < R; ++c) x0[c] = 0.0;
    int it = 0;
    bool capped = false;
    double tpend = 1.0;
    int new_kind = -1, new_idx = -1;
    // direction mode (qp_rows.h's header: the entering constraint depends on the working set) -- pair-uniform state
    bool dirm = false, restarted = false, pend_inf = false;
    int fail = 0;                // status of a problem that ends without a solution
    double dyn_keep = 1.0;       // |dy|_inf of a pending Farkas certificate

    // own frame -> original data
    auto korig = [&](int j) -> int { return isR ? M - j : j; };                   // original knot of own knot j
    auto sorig = [&](int s) -> int { return isR ? M - 1 - s : s; };               // original segment of own segment s (joins own knots s, s + 1)
    auto Tseg = [&](int s) -> double { return a.times[s0 + sorig(s)]; };
    auto klo = [&](int j) -> double { const long long i = base3 + 3LL * korig(j); return a.corr_lo ? a.corr_lo[i] : a.waypoints[i]; };
    auto khi = [&](int j) -> double { const long long i = base3 + 3LL * korig(j); return a.corr_hi ? a.corr_hi[i] : a.waypoints[i]; };
    auto kbit = [&](unsigned long long msk, int j) -> bool { return (msk >> (korig(j) & 63)) & 1ull; };
    auto sbit = [&](unsigned long long msk, int s) -> bool { return (msk >> (sorig(s) & 63)) & 1ull; };
    auto rlo = [&](int s, int j) -> double { return a.row_lo[((size_t)(s0 + sorig(s)) * K + j) * 3 + ax]; };
    auto rhi = [&](int s, int j) -> double { return a.row_hi[((size_t)(s0 + sorig(s)) * K + j) * 3 + ax]; };
    // functional of row slot j of own segment s in the OWN frame: value = gl' x'_s + gr' x'_{s+1}
    // functional of row slot j of own segment s in the OWN frame: value = gl' x'_s + gr' x'_{s+1}; for the reversed lane the two end
    // knots swap and odd derivatives change sign
    auto rowf = [&](int s, int j, double (&gl)[R], double (&gr)[R]) {
        const double* gp = aa.gfun + ((size_t)(s0 + sorig(s)) * K + j) * 2 * R;
        double ol[R], orr[R];
#pragma unroll
        for (int c = 0; c < R; ++c) { ol[c] = gp[c]; orr[c] = gp[R + c]; }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const double fl = (c & 1) ? -orr[c] : orr[c], fr = (c & 1) ? -ol[c] : ol[c];
            gl[c] = isR ? fl : ol[c];
            gr[c] = isR ? fr : orr[c];
        }
    };

    for (;;) {
        // ================= hand out problems to the free lane pairs =================
        {
            const bool need = g < 0;
            const unsigned long long needm = __ballot(need);
            if (needm != 0ull && !queue_empty) {
                const int npairs = __popcll(needm) >> 1;
                const int leader = (int)__builtin_ctzll(needm);
                unsigned int qb = 0;
                if (lane == leader) qb = atomicAdd(ticket, (unsigned int)npairs);
                qb = __shfl(qb, leader, 64);
                if ((long long)qb + npairs >= total) queue_empty = true;
                if (need) {
                    const int rank = __popcll(needm & ((1ull << lane) - 1ull)) >> 1;
                    const long long q = (long long)qb + rank;
                    if (q < total) {
                        int bq = (int)(q / 3), axn = (int)(q - 3LL * bq);
                        int bn = aa.order ? aa.order[bq] : bq;
                        if (GI) { const unsigned int gr_ = aa.redo[q]; bn = (int)(gr_ / 3u); axn = (int)(gr_ - 3u * (unsigned int)bn); }
                        const long long gn = 3LL * bn + axn;
                        const unsigned long long* dsc = aa.desc + (size_t)gn * (2 + 2 * K);
                        const unsigned long long d0 = dsc[0];
                        // (with the fused emission a trajectory that rows_prep_kernel flagged on ANY axis is left untouched on all three, as
                        // corridor_emit_kernel left it)
                        if ((d0 & 1ull) && !(aa.coeff && a.status[bn] == (int32_t)UAVQP_INVALID_INPUT)) {
                            int sn, Mn;
                            if (a.uniform > 0) { Mn = a.uniform; sn = bn * Mn; } else { sn = a.seg_offsets[bn]; Mn = a.seg_offsets[bn + 1] - sn; }
                            g = gn; b = bn; ax = axn; M = Mn; s0 = sn;
                            base3 = 3LL * ((long long)sn + bn) + axn;
                            m = isR ? M / 2 : (M + 1) / 2;
                            eqmask = dsc[1];
#pragma unroll
                            for (int j = 0; j < K; ++j) { rused[j] = dsc[2 + 2 * j]; req[j] = dsc[3 + 2 * j]; ract[j] = req[j]; rup[j] = 0ull; pract[j] = 0ull; }
                            const double* bc = a.bc + (size_t)bn * 2 * ND * 3 + axn;
                            x0[0] = a.waypoints[base3 + (isR ? 3 * M : 0)];
#pragma unroll
                            for (int d = 0; d < ND; ++d) {
                                const double v = bc[((isR ? ND : 0) + d) * 3];
                                x0[d + 1] = (isR && ((d & 1) == 0)) ? -v : v;
                            }
                            pin = eqmask; upper = 0ull;
                            if (a.warm) {
                                const unsigned long long valid = M >= 2 ? ((1ull << M) - 2ull) : 0ull;
                                const unsigned long long w0 = a.warm[(size_t)gn * 2] & valid & ~eqmask;
                                pin |= w0;
                                upper = a.warm[(size_t)gn * 2 + 1] & w0;
                            }
                            if (a.warm_rows) {
                                // rows of the starting set (qp_rows_dual.h): like the boxes above they enter with multiplier 0 -- a wrong guess
                                // leaves again with a zero-length step, a right one is confirmed by the first solve
#pragma unroll
                                for (int j = 0; j < K; ++j) {
                                    const unsigned long long w = a.warm_rows[(size_t)gn * 2 * K + 2 * j] & rused[j] & ~req[j];
                                    ract[j] |= w;
                                    rup[j] = a.warm_rows[(size_t)gn * 2 * K + 2 * j + 1] & w;
                                }
                            }
                            it = 0; capped = false; tpend = 1.0; ppin = 0ull; new_kind = -1; new_idx = -1;
                            dirm = false; restarted = false; pend_inf = false; fail = 0; dyn_keep = 1.0; redo = false;
                        }
                    }
                }
            }
        }
        const bool act = g >= 0;
        R2_CT(0);
        if (__ballot(act) == 0ull) {
            if (queue_empty) break;
            continue;
        }
        const int mm = act ? m : 0;
        bool finish = false;
        bool single = act && (M == 1);   // no free knot: the rows can only be CHECKED (by the L lane, own segment 0)
        // ---- direction mode of this trip (rare: every extra below sits behind the wave-uniform any_dirm).  The entering constraint
        // q = (new_kind, new_idx) as a functional qgl' x'_qk + qgr' x'_{qk+1} of the OWN frame, in the lane that owns it (a box: e_0 on the
        // right knot; the meeting knot's box: the L lane); qs = -+1 towards the violated side there, 0 in the other lane
        const bool dm = GI && act && dirm;
        const bool any_dirm = GI && __ballot(dm) != 0ull;
        double qs = 0.0, qsg = 0.0;
        int qk = -100;
        if (any_dirm && dm) {
            if (new_kind == 0) {
                qsg = ((upper >> new_idx) & 1ull) ? -1.0 : 1.0;
                const int jq = isR ? M - new_idx : new_idx;
                if (jq >= 1 && (jq < mm || (jq == mm && !isR))) { qk = jq - 1; qs = qsg; }
            } else {
                unsigned long long rupq = rup[0];
#pragma unroll
                for (int j = 1; j < K; ++j) rupq = (new_kind - 1) == j ? rup[j] : rupq;
                qsg = ((rupq >> new_idx) & 1ull) ? -1.0 : 1.0;
                const int sq = isR ? M - 1 - new_idx : new_idx;
                if (sq >= 0 && sq < mm) { qk = sq; qs = qsg; }
            }
        }
        // the part of qs c_q that sits on own knot j (only ever called behind any_dirm: the functional is fetched again at each use --
        // nothing of the rare path stays in registers over the sweeps)
        auto qinj = [&](int j, double (&o)[R]) {
#pragma unroll
            for (int c = 0; c < R; ++c) o[c] = 0.0;
            if (qs != 0.0 && (j == qk + 1 || (j == qk && j >= 1))) {
                if (new_kind == 0) { o[0] = qs; }
                else {
                    double gl_[R], gr_[R];
                    if (K == 1 || new_kind == 1) rowf(qk, 0, gl_, gr_);
                    else rowf(qk, K - 1, gl_, gr_);
#pragma unroll
                    for (int c = 0; c < R; ++c) o[c] = qs * (j == qk + 1 ? gr_[c] : gl_[c]);
                }
            }
        };

        // ================= forward sweep: own knots j = 1 .. m - 1 (slot m - j) =================
        FullBlocks<R> sa;                 // blocks of own segment j - 1
        SmallLDL<B> lprev;
        LDLPack<B>::zero(lprev);          // own knot 0: nothing free, h = its Hermite data
        double hprev[B];
#pragma unroll
        for (int i = 0; i < B; ++i) hprev[i] = i < R ? x0[i] : 0.0;
        if (any_dirm) {
#pragma unroll
            for (int i = 0; i < R; ++i) hprev[i] = dm ? 0.0 : hprev[i];
        }
        bool pprev = false;
        double zprev = 0.0;
        // own partial block of a knot: rows of own segment j - 1 (mu part), coupling to the previous block eliminated.
        // full = true adds the next segment's start block and the coupling to known neighbours behind it (interior own knots);
        // the meeting knot leaves both to the partner lane.
        // every global input of a block step: requested together and ONE STEP AHEAD of its use (the loop below loads the inputs of
        // block j + 1 while block j is eliminated), unconditionally -- what a lane does not need is discarded by the selects
        struct FIn { double T, gl[K][R], gr[K][R], rl[K], rh[K], bl, bh, nl, nh; };
        auto load_fin = [&](int j, FIn& f) {
            const int jc = j < 1 ? 1 : (j > mm ? (mm < 1 ? 1 : mm) : j);       // clamped: a speculative load stays inside the trajectory
            f.T = Tseg(jc < M ? jc : M - 1);
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
                rowf(jc - 1, jj, f.gl[jj], f.gr[jj]);
                f.rl[jj] = rlo(jc - 1, jj);
                f.rh[jj] = rhi(jc - 1, jj);
            }
            f.bl = klo(jc); f.bh = khi(jc);
            const int jn = jc + 1 <= mm ? jc + 1 : jc;
            f.nl = klo(jn); f.nh = khi(jn);
            if (any_dirm) {      // direction mode: the homogeneous system -- every bound and pinned value is 0
                f.bl = dm ? 0.0 : f.bl; f.bh = dm ? 0.0 : f.bh; f.nl = dm ? 0.0 : f.nl; f.nh = dm ? 0.0 : f.nh;
#pragma unroll
                for (int jj = 0; jj < K; ++jj) { f.rl[jj] = dm ? 0.0 : f.rl[jj]; f.rh[jj] = dm ? 0.0 : f.rh[jj]; }
            }
        };
        // Known components need no case analysis (the steps stay single basic blocks):
        //   * an INACTIVE row has its functionals zeroed and a unit diagonal: its mu row / column is decoupled and solves to 0;
        //   * a PINNED position (interior knots only; the meeting knot's is applied to the combined block) moves D[.][0] z to the right-hand
        //     side, becomes the identity row, and is masked out of the coupling to the previous block -- all by selects on pk.
        auto block = [&](int j, bool full, const FIn& in, double (&D)[B][B], double (&rhs)[B], bool (&racv)[K]) {
            const double Tj = in.T;
            double gl[K][R], gr[K][R], rb[K];
            FullBlocks<R> sb;
            if (full) sb.build(Tj);
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
                racv[jj] = sbit(ract[jj], j - 1);
                rb[jj] = racv[jj] ? (sbit(rup[jj], j - 1) ? in.rh[jj] : in.rl[jj]) : 0.0;
#pragma unroll
                for (int c = 0; c < R; ++c) { gl[jj][c] = racv[jj] ? in.gl[jj][c] : 0.0; gr[jj][c] = racv[jj] ? in.gr[jj][c] : 0.0; }
            }
            const bool interior = korig(j) >= 1 && korig(j) <= M - 1;
            const bool pk = full & interior & kbit(pin, j);
            const double zc = pk ? (kbit(upper, j) ? in.bh : in.bl) : 0.0;
            const double zpm = pprev ? zprev : 0.0;
#pragma unroll
            for (int i = 0; i < B; ++i) {
                rhs[i] = 0.0;
#pragma unroll
                for (int c = 0; c < B; ++c) D[i][c] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
#pragma unroll
                for (int c = 0; c < R; ++c) D[i][c] = sa.B11[i][c] + (full ? sb.B00(i, c) : 0.0);
                rhs[i] -= sa.B01[0][i] * zpm;
            }
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
#pragma unroll
                for (int c = 0; c < R; ++c) D[R + jj][c] = gr[jj][c];
                D[R + jj][R + jj] = racv[jj] ? 0.0 : 1.0;
                rhs[R + jj] = rb[jj] - gl[jj][0] * zpm;
            }
            if (any_dirm) {
                double qv[R];
                qinj(j, qv);
#pragma unroll
                for (int i = 0; i < R; ++i) rhs[i] += qv[i];
            }
            if (full) {
                const bool pnext = kbit(pin, j + 1);   // own knot j + 1 <= m: an interior knot (the meeting knot at the latest)
                const double zn = pnext ? (kbit(upper, j + 1) ? in.nh : in.nl) : 0.0;
#pragma unroll
                for (int i = 0; i < R; ++i) rhs[i] -= sb.B01[i][0] * zn;
            }
#pragma unroll
            for (int i = 1; i < B; ++i) {
                rhs[i] -= D[i][0] * zc;
                D[i][0] = pk ? 0.0 : D[i][0];
            }
            D[0][0] = pk ? 1.0 : D[0][0];
            // coupling to the previous block (rows = x_{j-1}); (pk: a pinned position is a known value -- its coupling went to the previous
            // block's right-hand side; at the meeting knot too, where pk comes from the caller through the combined block)
            const bool pkm = interior & kbit(pin, j);
            double Mp[R][B];
#pragma unroll
            for (int c = 0; c < R; ++c) {
#pragma unroll
                for (int i = 0; i < R; ++i) Mp[c][i] = sa.B01[c][i];
#pragma unroll
                for (int jj = 0; jj < K; ++jj) Mp[c][R + jj] = gl[jj][c];
            }
#pragma unroll
            for (int i = 0; i < B; ++i) Mp[0][i] = pprev ? 0.0 : Mp[0][i];
#pragma unroll
            for (int c = 0; c < R; ++c) Mp[c][0] = pkm ? 0.0 : Mp[c][0];
            double Ep[B][B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double col[B];
#pragma unroll
                for (int c = 0; c < B; ++c) col[c] = c < R ? Mp[c][i] : 0.0;
                lprev.solve(col);
#pragma unroll
                for (int c = 0; c < B; ++c) Ep[c][i] = col[c];
            }
#pragma unroll
            for (int i = 0; i < B; ++i)
#pragma unroll
                for (int q = 0; q < R; ++q) {
#pragma unroll
                    for (int c = 0; c <= i; ++c) D[i][c] -= Mp[q][i] * Ep[q][c];
                    rhs[i] -= Mp[q][i] * hprev[q];
                }
            rhs[0] = pk ? zc : rhs[0];
            if (full) { sa = sb; pprev = pk; zprev = zc; }
        };
        if (mm >= 1 && !single) sa.build(Tseg(0));
        FIn fnx;
        if (mm >= 1 && !single) load_fin(1, fnx);
        // (NRS > 0: the sweeps run over the slot index s = m - j, wave-uniform: a lane whose half is shorter than the wave's longest joins late)
        const int mtop = NRS > 0 ? __builtin_amdgcn_readfirstlane(wave_max_int(mm)) : 0;
        for (int sq = NRS > 0 ? mtop - 1 : mm - 1; sq >= 1; --sq) {
            const int j = mm - sq;         // NRS == 0: j = 1 .. m - 1 as before; NRS > 0: j < 1 = not started yet
            if (NRS > 0 && j < 1) continue;
            double D[B][B], rhs[B];
            bool racv[K];
            const FIn fcur = fnx;
            load_fin(j + 1, fnx);          // (j + 1 <= mm: the last one is the meeting block's)
            block(j, true, fcur, D, rhs, racv);
            SmallLDL<B> ldl;
            ldl.factor(D);
            ldl.solve(rhs);
            {
                double e[NL];
                LDLPack<B>::get(ldl, e);
                slot_st(sq, e, rhs);
            }
#pragma unroll
            for (int i = 0; i < B; ++i) hprev[i] = rhs[i];
            lprev = ldl;
        }

        R2_CT(1);
        // ================= meeting block [x_c ; mu_L ; mu_R], assembled in the L frame =================
        double ym[B];             // solution of the own meeting block [x_c (own frame) ; mu_own]
#pragma unroll
        for (int i = 0; i < B; ++i) ym[i] = 0.0;
        {
            double D[B][B], rhs[B];
            bool racm[K];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                rhs[i] = 0.0;
#pragma unroll
                for (int c = 0; c < B; ++c) D[i][c] = 0.0;
            }
            const bool has = act && !single && mm >= 1;
            if (has) block(mm, false, fnx, D, rhs, racm);
            // exchange lower triangles and right-hand sides; conj = F . F on the x part of the partner's (reversed-frame) block
            double C[BM][BM], r7[BM];
#pragma unroll
            for (int i = 0; i < BM; ++i) {
                r7[i] = 0.0;
#pragma unroll
                for (int c = 0; c < BM; ++c) C[i][c] = 0.0;
            }
            auto sgn = [](int c) -> double { return (c & 1) ? -1.0 : 1.0; };
#pragma unroll
            for (int i = 0; i < B; ++i) {
#pragma unroll
                for (int c = 0; c <= i; ++c) {
                    const double own = D[i][c], oth = swap_pair(D[i][c]);
                    // L-frame value of each lane's entry: x-x entries of the R lane pick up F_i F_c, mu-x entries F_c
                    const double fo = (i < R ? sgn(i) : 1.0) * (c < R ? sgn(c) : 1.0);
                    const double vL = isR ? oth : own, vR = isR ? own * fo : oth * fo;   // the L lane's / the R lane's block, L frame
                    if (i < R) C[i][c] = vL + vR;                                         // x - x: the two halves add
                    else {
                        C[i][c] = (c < R || c == i || c < i) ? vL : 0.0;                  // mu_L rows: positions R .. R + K - 1
                        C[K + i][c < R ? c : K + c] = vR;                                 // mu_R rows: positions R + K .. R + 2 K - 1
                    }
                }
                const double own = rhs[i], oth = swap_pair(rhs[i]);
                const double fo = i < R ? sgn(i) : 1.0;
                const double vL = isR ? oth : own, vR = isR ? own * fo : oth * fo;
                if (i < R) r7[i] = vL + vR;
                else { r7[i] = vL; r7[K + i] = vR; }
            }
            // the pinned meeting position (inactive rows of either side are decoupled unit rows already, see block())
            const bool pc = has && mm >= 1 && kbit(pin, mm) && korig(mm) >= 1 && korig(mm) <= M - 1;
            double zc = pc ? (kbit(upper, mm) ? khi(mm) : klo(mm)) : 0.0;
            if (any_dirm) zc = dm ? 0.0 : zc;
#pragma unroll
            for (int i = 1; i < BM; ++i) {
                r7[i] -= C[i][0] * zc;
                C[i][0] = pc ? 0.0 : C[i][0];
            }
            C[0][0] = pc ? 1.0 : C[0][0];
            r7[0] = pc ? zc : r7[0];
            SmallLDL<BM> ldl;
            ldl.factor(C);
            ldl.solve(r7);
            if (has) {
#pragma unroll
                for (int c = 0; c < R; ++c) ym[c] = isR ? r7[c] * sgn(c) : r7[c];
#pragma unroll
                for (int jj = 0; jj < K; ++jj) ym[R + jj] = isR ? r7[R + K + jj] : r7[R + jj];
#pragma unroll
                for (int q = 0; q < B; ++q) rec_st(0, NL + q, ym[q]);
            }
        }

        R2_CT(2);
        // ================= backward sweep + the decisions of this iteration (own constraints only) =================
        double vmax = 0.0, vq_now = 0.0;   // most violated inactive constraint; the violation of (new_kind, new_idx) (a pending certificate's margin)
        int vkind = -1, vidx = NONE;
        bool vupper = false;
        double tmin = dm ? 1e300 : 2.0;   // first multiplier to reach zero (direction mode: along the dependency, unbounded)
        double hz = 0.0, dymax = 0.0;     // direction mode: |H z|_inf by its diagonal part, largest rate of a multiplier of the working set
        int tkind = -1, tidx = NONE;
        bool inconsistent = false;
        double lam_meet = 0.0, mag_meet = 0.0;   // own half of the meeting knot's box multiplier
        const bool any_pend = GI && __ballot(act && pend_inf) != 0ull;
        auto viol_cand = [&](double sc, double viol, int kind, int idx, bool up) {
            if (any_dirm && dm) return;      // (no constraint is looked for: the one that enters is known)
            if (sc > 1e-12 && (sc > vmax || (sc == vmax && (kind < vkind || (kind == vkind && idx < vidx))))) { vmax = sc; vkind = kind; vidx = idx; vupper = up; }
            if (any_pend && kind == new_kind && idx == new_idx) vq_now = viol;
        };
        auto step_cand = [&](double t, int kind, int idx) {
            if (t < tmin || (t == tmin && (kind < tkind || (kind == tkind && idx < tidx)))) { tmin = t; tkind = kind; tidx = idx; }
        };
        // multiplier bookkeeping of one constraint: current value by the pending interpolation, ratio test; lc / ln: its stored current /
        // new multiplier in, the values to store out
        // (bad: the wrong-signed amount of lam_new, mag: scale of its rounding; rows add |current multiplier| to the scale as qp_rows.h does)
        // (direction mode: lam_new is the RATE of the multiplier per unit step of q's; its "new" value is lc + rate, the ratio test is unbounded,
        // the entering constraint itself stores the unit rate: + for a box multiplier towards the violated side, - for a row's (row convention))
        auto dual = [&](double& lc, double& ln, bool active, bool equality, bool was, double lam_new, double bad, double mag, bool add_lc, int kind, int idx) {
            if (VER) {            // first solve from the starting set: every current multiplier is 0; a wrong-signed new one sends the problem to the second pass
                if (active && !equality && bad > 1e-13 * mag && bad > 0.0) step_cand(0.0, kind, idx);
                return;
            }
            if (!any_dirm) {      // the common path: a wave without a lane pair in direction mode
                lc = was ? lc + tpend * (ln - lc) : 0.0;
                const bool wrong = bad > 1e-13 * (mag + (add_lc ? fabs(lc) : 0.0)) && bad > 0.0;
                if (active && !equality && wrong && !(new_kind == kind && new_idx == idx)) {
                    const double den = lc - lam_new;
                    double t = den != 0.0 ? lc * fast_rcp(den) : 0.0;   // (fast_rcp: full float64 accuracy for normal arguments, a third of the instructions of a division)
                    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
                    step_cand(t, kind, idx);
                }
                lc = active ? lc : 0.0;
                ln = active ? lam_new : 0.0;
                return;
            }
            lc = was ? (tpend == 0.0 ? lc : lc + tpend * (ln - lc)) : 0.0;
            const double lnew = dm ? lc + lam_new : lam_new;
            const bool wrong = dm ? (bad > 1e-9 + (add_lc ? 0.0 : 1e-11 * mag)) : (bad > 1e-13 * (mag + (add_lc ? fabs(lc) : 0.0)) && bad > 0.0);
            if (active && !equality && wrong && !(new_kind == kind && new_idx == idx)) {
                const double den = lc - lnew;
                double t = den != 0.0 ? lc * fast_rcp(den) : 0.0;
                t = t < 0.0 ? 0.0 : ((t > 1.0 && !dm) ? 1.0 : t);
                step_cand(t, kind, idx);
            }
            if (dm && active) dymax = fmax(dymax, fabs(lam_new));
            lc = active ? lc : 0.0;
            ln = active ? lnew : ((dm && new_kind == kind && new_idx == idx) ? (kind == 0 ? qsg : -qsg) : 0.0);
        };
        if (act && !single) {
            double yn[B], ynn[B];    // blocks of own knots j + 1, j + 2
#pragma unroll
            for (int i = 0; i < B; ++i) { yn[i] = ym[i]; ynn[i] = 0.0; }
            FullBlocks<R> snn;       // own segment j + 1
            double glp0[K];          // position components of g_l of the ACTIVE rows of own segment j + 1 (0 otherwise)
#pragma unroll
            for (int jj = 0; jj < K; ++jj) glp0[jj] = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int c = 0; c < R; ++c) { snn.B11[i][c] = 0.0; snn.B01[i][c] = 0.0; }
            // global inputs of a backward step (own segment j): duration, functionals and bounds of its rows, box of own knot j + 1, stored
            // multipliers of block j + 1
            // The part the x_j chain needs at once (duration, functionals) is requested ONE STEP AHEAD; the rest (bounds, stored multipliers:
            // "late" work) together at the top of its step.  (Everything one step ahead was measured slower: 51.7 k vs 41.0 k cycles per trip
            // -- the second full input set pushes the loop into the accumulator registers.)
            struct BInA { double T, gl[K][R], gr[K][R]; };
            struct BInB { double rl[K], rh[K], bl, bh, lc[NCN], ln[NCN]; };
            auto load_bina = [&](int j, BInA& f) {
                const int jc = j < 0 ? 0 : j;
                f.T = Tseg(jc);
#pragma unroll
                for (int jj = 0; jj < K; ++jj) rowf(jc, jj, f.gl[jj], f.gr[jj]);
            };
            auto load_binb = [&](int j, BInB& f) {
#pragma unroll
                for (int jj = 0; jj < K; ++jj) {
                    f.rl[jj] = rlo(j, jj);
                    f.rh[jj] = rhi(j, jj);
                }
                f.bl = klo(j + 1); f.bh = khi(j + 1);
                if (any_dirm) {
#pragma unroll
                    for (int jj = 0; jj < K; ++jj) { f.rl[jj] = dm ? 0.0 : f.rl[jj]; f.rh[jj] = dm ? 0.0 : f.rh[jj]; }
                }
                const int sl = mm - j - 1;
#pragma unroll
                for (int c = 0; c < NCN; ++c) { f.lc[c] = VER ? 0.0 : LC(sl, c); f.ln[c] = VER ? 0.0 : LN(sl, c); }
            };
            BInA anx;
            load_bina(mm - 1, anx);
            for (int sq = 1; sq <= (NRS > 0 ? mtop : mm); ++sq) {
                const int j = mm - sq;          // (NRS > 0: the slot index is the wave-uniform loop variable; a shorter half is done early)
                if (NRS > 0 && j < 0) continue;
                // own segment j joins own knots j and j + 1; block j + 1 (= yn) carries its rows' multipliers
                const int sl1 = mm - j - 1;                 // slot of own knot j + 1
                const BInA ba_ = anx;
                load_bina(j - 1, anx);
                BInB bc_;
                load_binb(j, bc_);
                const double Tj = ba_.T;
                double gl[K][R], gr[K][R], rlv[K], rhv[K], lcv[NCN], lnv[NCN];
                bool usedj[K];
#pragma unroll
                for (int jj = 0; jj < K; ++jj) {
#pragma unroll
                    for (int c = 0; c < R; ++c) { gl[jj][c] = ba_.gl[jj][c]; gr[jj][c] = ba_.gr[jj][c]; }
                    rlv[jj] = bc_.rl[jj];
                    rhv[jj] = bc_.rh[jj];
                }
                const double bl = bc_.bl, bh = bc_.bh;
#pragma unroll
                for (int c = 0; c < NCN; ++c) { lcv[c] = bc_.lc[c]; lnv[c] = bc_.ln[c]; }
                FullBlocks<R> sn;
                sn.build(Tj);
#pragma unroll
                for (int jj = 0; jj < K; ++jj) {
                    usedj[jj] = sbit(rused[jj], j);
#pragma unroll
                    for (int c = 0; c < R; ++c) { gl[jj][c] = usedj[jj] ? gl[jj][c] : 0.0; gr[jj][c] = usedj[jj] ? gr[jj][c] : 0.0; }
                }
                double y[B];
                const int s = sq;
                if (j >= 1) {
                    double h[B], t[B], e[NL];
                    slot_ld(s, e, h);
#pragma unroll
                    for (int i = 0; i < B; ++i) t[i] = 0.0;
                    const bool pk = kbit(pin, j), pn = kbit(pin, j + 1);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        if (pk && i == 0) continue;
                        double acc = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c)
                            if (!(pn && c == 0)) acc += sn.B01[i][c] * yn[c];
#pragma unroll
                        for (int jj = 0; jj < K; ++jj)
                            if (sbit(ract[jj], j)) acc += gl[jj][i] * yn[R + jj];
                        t[i] = acc;
                    }
                    SmallLDL<B> ldl;
                    LDLPack<B>::set(ldl, e);
                    ldl.solve(t);
#pragma unroll
                    for (int i = 0; i < B; ++i) y[i] = h[i] - t[i];
                    slot_st_y(s, y);
                } else {
#pragma unroll
                    for (int i = 0; i < B; ++i) y[i] = i < R ? x0[i] : 0.0;
                    if (any_dirm) {
#pragma unroll
                        for (int i = 0; i < R; ++i) y[i] = dm ? 0.0 : y[i];
                    }
                }
                // ---- rows of own segment j: values (inactive: violation; active: must sit on the bound), multipliers (block j + 1)
                double la = 0.0, ma = 0.0;   // own knot j + 1's box multiplier: the part of own segment j
#pragma unroll
                for (int jj = 0; jj < K; ++jj) {
                    const int os = sorig(j);
                    const bool aj = sbit(ract[jj], j), ej = sbit(req[jj], j), uj = sbit(rup[jj], j);
                    if (usedj[jj]) {
                        double v = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) v += gl[jj][c] * y[c] + gr[jj][c] * yn[c];
                        if (aj) {
                            const double bnd = uj ? rhv[jj] : rlv[jj];
                            if (!(fabs(v - bnd) <= 1e-8 * (1.0 + fabs(bnd)))) inconsistent = true;
                            const double t4 = gr[jj][0] * yn[R + jj];
                            la += t4; ma += fabs(t4);
                        } else {
                            const double l = rlv[jj], h = rhv[jj];
                            const double below = l - v, above = v - h;
                            const double viol = below > above ? below : above;
                            viol_cand(viol * fast_rcp(1.0 + fabs(below > above ? l : h)), viol, 1 + jj, os, above > below);
                        }
                        const double mu = yn[R + jj];
                        const bool was = (pract[jj] >> (os & 63)) & 1ull;
                        dual(lcv[1 + jj], lnv[1 + jj], aj, ej, was, mu, uj ? -mu : mu, fabs(mu), true, 1 + jj, os);
                    } else {
                        lcv[1 + jj] = 0.0;
                        lnv[1 + jj] = 0.0;
                    }
                }
                // ---- box of own knot j + 1: multiplier = row 0 of the unmasked block row (own segments j and j + 1, x_j, x_{j+1}, x_{j+2},
                // the multipliers of the active rows of both segments); free position outside its box?
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const double t1 = sn.B01[c][0] * y[c], t2 = sn.B11[0][c] * yn[c];
                    la += t1 + t2;
                    ma += fabs(t1) + fabs(t2);
                }
                if (any_dirm) {
                    double qv[R];
                    qinj(j + 1, qv);
                    la -= qv[0];               // (the injected right-hand side is no part of the multiplier)
                    if (dm) {
#pragma unroll
                        for (int c = 0; c < R; ++c) hz = fmax(hz, fabs((sn.B11[c][c] + (j + 1 == mm ? 0.0 : snn.B00(c, c))) * yn[c]));
                    }
                }
                if (j + 1 == mm) {
                    lam_meet = la;     // own half of the meeting knot's multiplier: finished across the pair below
                    mag_meet = ma;
                } else {
                    double lam = la, mag = ma;
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const double t2 = snn.B00(0, c) * yn[c], t3 = snn.B01[0][c] * ynn[c];
                        lam += t2 + t3;
                        mag += fabs(t2) + fabs(t3);
                    }
#pragma unroll
                    for (int jj = 0; jj < K; ++jj) { const double t5 = glp0[jj] * ynn[R + jj]; lam += t5; mag += fabs(t5); }
                    const int kk = korig(j + 1);
                    const bool pj = kbit(pin, j + 1), ej = kbit(eqmask, j + 1), uj = kbit(upper, j + 1);
                    const bool was = (ppin >> (kk & 63)) & 1ull;
                    dual(lcv[0], lnv[0], pj, ej, was, lam, uj ? lam : -lam, mag, false, 0, kk);
                    if (!pj) {
                        const double l = bl, h = bh, v = yn[0];
                        const double below = l - v, above = v - h;
                        const double viol = below > above ? below : above;
                        viol_cand(viol * fast_rcp(1.0 + fabs(below > above ? l : h)), viol, 0, kk, above > below);
                    }
                }
                // the block's multipliers back (the meeting knot's box -- slot 0, constraint 0 -- belongs to the pair step below)
                if (!VER) {
#pragma unroll
                    for (int c = 0; c < NCN; ++c)
                        if (c > 0 || j + 1 < mm) { LC(sl1, c) = lcv[c]; LN(sl1, c) = lnv[c]; }
                }
                // ---- rotate
#pragma unroll
                for (int i = 0; i < B; ++i) { ynn[i] = yn[i]; yn[i] = y[i]; }
                snn = sn;
#pragma unroll
                for (int jj = 0; jj < K; ++jj) glp0[jj] = sbit(ract[jj], j) ? gl[jj][0] : 0.0;
            }
        }
        R2_CT(3);
        R2_CT(4);
        // ---- the meeting knot's box: both halves of its multiplier
        {
            const double ol = swap_pair(lam_meet), om = swap_pair(mag_meet);
            if (act && !single && mm >= 1 && M >= 2) {
                const int kk = korig(mm);
                const double lam = (isR ? ol + lam_meet : lam_meet + ol), mag = (isR ? om + mag_meet : mag_meet + om);
                const bool pj = kbit(pin, mm), ej = kbit(eqmask, mm), uj = kbit(upper, mm);
                const bool was = (ppin >> (kk & 63)) & 1ull;
                double lc0 = VER ? 0.0 : LC(0, 0), ln0 = VER ? 0.0 : LN(0, 0);
                dual(lc0, ln0, pj, ej, was, lam, uj ? lam : -lam, mag, false, 0, kk);
                if (!VER) {
                    LC(0, 0) = lc0;
                    LN(0, 0) = ln0;
                }
                if (!pj) {
                    const double l = klo(mm), h = khi(mm), v = ym[0];
                    const double below = l - v, above = v - h;
                    const double viol = below > above ? below : above;
                    viol_cand(viol * fast_rcp(1.0 + fabs(below > above ? l : h)), viol, 0, kk, above > below);
                }
            }
        }
        if (single && !isR) {
            // the polynomial is fixed by the boundary data; a row it violates makes the problem infeasible
            bool viol = false;
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
                if (!(rused[jj] & 1ull)) continue;
                double gl[R], gr[R];
                rowf(0, jj, gl, gr);
                double xM[R];
                xM[0] = a.waypoints[base3 + 3 * M];
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
#pragma unroll
                for (int d = 0; d < ND; ++d) xM[d + 1] = bc[(ND + d) * 3];
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < R; ++c) v += gl[c] * x0[c] + gr[c] * xM[c];
                const double l = rlo(0, jj), h = rhi(0, jj);
                viol = viol || (l - v > 1e-9 * (1.0 + fabs(l))) || (v - h > 1e-9 * (1.0 + fabs(h)));
            }
            capped = viol;
            fail = (int)UAVQP_PRIMAL_INFEASIBLE;      // (if capped: the only polynomial there is violates a row)
        }
        // ================= combine the halves' decisions (order-independent rules: identical in both lanes afterwards) =================
        {
            const double ov = swap_pair(vmax), ot = swap_pair(tmin);
            const int ovk = swap_pair_i(vkind), ovi = swap_pair_i(vidx), ovu = swap_pair_i(vupper ? 1 : 0);
            const int otk = swap_pair_i(tkind), oti = swap_pair_i(tidx), oinc = swap_pair_i(inconsistent ? 1 : 0), ocap = swap_pair_i(capped ? 1 : 0);
            if (ovk >= 0 && (vkind < 0 || ov > vmax || (ov == vmax && (ovk < vkind || (ovk == vkind && ovi < vidx))))) { vmax = ov; vkind = ovk; vidx = ovi; vupper = ovu != 0; }
            if (otk >= 0 && (tkind < 0 || ot < tmin || (ot == tmin && (otk < tkind || (otk == tkind && oti < tidx))))) { tmin = ot; tkind = otk; tidx = oti; }
            inconsistent = inconsistent || (oinc != 0);
            capped = capped || (ocap != 0);
            if (any_dirm) { hz = fmax(hz, swap_pair(hz)); dymax = fmax(dymax, swap_pair(dymax)); }
            if (any_pend) vq_now = fmax(vq_now, swap_pair(vq_now));
        }
        // ================= dual active-set step (as qp_rows.h) =================
        bool done = false;
        if (act) {
            if (single) {
                finish = true;
            } else {
                ++it;
                ppin = pin;
#pragma unroll
                for (int j = 0; j < K; ++j) pract[j] = ract[j];
                if (capped) {
                    finish = true;
                    // (a pending certificate is accepted as OSQP accepts one: the violation of q at this minimiser >= eps_prim_inf |dy|_inf)
                    if (pend_inf && vq_now >= a.eps_prim_inf * dyn_keep) fail = (int)UAVQP_PRIMAL_INFEASIBLE;
                } else if (dirm) {
                    // ---- the direction-mode solve: rates of the multipliers of W along the dependency of q on W (qp_rows.h)
                    dirm = false;
                    const unsigned long long qbit = 1ull << new_idx;
                    const double dyn = fmax(1.0, dymax);
                    const bool dep = !inconsistent && hz <= fmax(a.eps_prim_inf, 1e-6) * dyn;     // (NaNs fail the test)
                    if (!inconsistent && tkind >= 0) {
                        tpend = tmin;
                        if (tkind == 0) pin &= ~(1ull << tidx);
                        else ract[tkind - 1] &= ~(1ull << tidx);
                        if (new_kind == 0) { pin |= qbit; ppin |= qbit; }
                        else { ract[new_kind - 1] |= qbit; pract[new_kind - 1] |= qbit; }
                    } else {
                        fail = (int)UAVQP_MAX_ITER_REACHED;
                        pend_inf = dep;
                        dyn_keep = dyn;
                        tpend = 0.0;
                        capped = true;
                    }
                } else if (inconsistent && !GI) {
                    redo = true;         // decided by the second pass
                    finish = true;
                } else if (VER && (tkind >= 0 || vkind >= 0)) {
                    redo = true;         // the starting set is not the final one: the second pass solves the problem from its start
                    finish = true;
                } else if (inconsistent) {
                    if (new_kind >= 0) {
                        // the constraint that has just entered depends on the working set: out again, one solve for the direction
                        if (new_kind == 0) pin &= ~(1ull << new_idx);
                        else ract[new_kind - 1] &= ~(1ull << new_idx);
                        dirm = true;
                        tpend = 0.0;
                    } else if (!restarted) {
                        // a singular STARTING set: once more from the boxes' equalities alone (the equality rows enter like violated constraints)
                        restarted = true;
                        pin = eqmask; upper = 0ull; ppin = 0ull;
#pragma unroll
                        for (int j = 0; j < K; ++j) { ract[j] = 0ull; rup[j] = 0ull; pract[j] = 0ull; }
                        tpend = 1.0;
                    } else {
                        tpend = 1.0;
                        capped = true;
                        fail = (int)UAVQP_MAX_ITER_REACHED;
                    }
                } else if (tkind >= 0) {
                    tpend = tmin;
                    if (tkind == 0) pin &= ~(1ull << tidx);
                    else ract[tkind - 1] &= ~(1ull << tidx);
                } else {
                    tpend = 1.0;
                    new_kind = -1;
                    new_idx = -1;
                    if (vkind < 0) {
                        done = true;
                    } else if (vkind == 0) {
                        pin |= 1ull << vidx;
                        if (vupper) upper |= 1ull << vidx; else upper &= ~(1ull << vidx);
                        new_kind = 0; new_idx = vidx;
                    } else {
                        ract[vkind - 1] |= 1ull << vidx;
                        if (vupper) rup[vkind - 1] |= 1ull << vidx; else rup[vkind - 1] &= ~(1ull << vidx);
                        new_kind = vkind; new_idx = vidx;
                    }
                }
                if (!done && !finish && !capped && it >= a.max_iter) {
                    if (!dirm) {
                        if (new_kind == 0) pin &= ~(1ull << new_idx);
                        else if (new_kind > 0) ract[new_kind - 1] &= ~(1ull << new_idx);
                    }
                    dirm = false;
                    new_kind = -1;
                    capped = true;
                    pend_inf = false;
                    fail = (int)UAVQP_MAX_ITER_REACHED;
                }
                if (done) finish = true;
            }
        }
        R2_CT(5);
#ifdef UAVQP_ROWS2_TIMING
        r2_acc[7] += 1;
#endif
        // ================= hand-over: a finished problem leaves its polynomials (or its Hermite solution) and frees the pair =================
        if (finish && redo && !isR) aa.redo[atomicAdd(a.queue + 1, 1u)] = (unsigned int)g;
        const bool emitp = finish && !redo;
        if (__ballot(emitp) != 0ull) {
            if (!aa.coeff) {
                static_assert(NRS == 0 || true, "");
                if (emitp) {
                    if constexpr (NRS > 0) {      // (register slots: compile-time slot indices)
#pragma unroll
                        for (int sx = 0; sx < NT + NRS; ++sx) {
                            const int j = mm - sx;
                            const int kk = korig(j);
                            if (j >= 1 && !(sx == 0 && isR) && kk >= 1 && kk <= M - 1) {
                                double* o = a.xsol + (base3 + 3LL * kk) * R;
#pragma unroll
                                for (int c = 0; c < R; ++c) {
                                    const double v = sx < NT ? RL(sx < NT ? sx : 0, NL + c) : rr[sx < NT ? 0 : sx - NT][NL + c];
                                    o[c] = (isR && (c & 1)) ? -v : v;
                                }
                            }
                        }
                    } else {
                    for (int j = 1; j <= mm; ++j) {
                        if (j == mm && isR) continue;      // the meeting knot is written by the L lane
                        const int kk = korig(j);
                        if (kk < 1 || kk > M - 1) continue;
                        double* o = a.xsol + (base3 + 3LL * kk) * R;
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const double v = rec_ld(mm - j, NL + c);
                            o[c] = (isR && (c & 1)) ? -v : v;
                        }
                    }
                    }
                }
            } else {
                // own segment j joins own knots j (slot m - j) and j + 1 (slot m - j - 1; own knot 0 is the boundary knot x0): Hermite data back in
                // the ORIGINAL frame (the reversed lane flips the odd derivatives and swaps the ends), then the same segment_coeffs_det on the
                // same numbers corridor_emit_kernel would read back from xsol
                constexpr int NC = 2 * R;
                const bool al16 = (reinterpret_cast<uintptr_t>(aa.coeff) & 15u) == 0;
                // (NX: own knots per lane the staged path handles -- all of the LDS slots, plus, in a kernel with workspace slots, those up to 8;
                //  the staging area [32 pairs][segments][2 R] must fit the LDS of the records)
                constexpr int NX = WS ? (NT > 8 ? NT : 8) : NT + NRS;
                const bool whole = a.uniform >= 2 && (a.uniform + 1) / 2 <= NX && 32 * a.uniform * NC <= NT * F * 64 && al16 && __ballot(act && !finish) == 0ull;
                if (whole) {
                    // the whole wave hands over in this trip (the rule: one verifying solve): the coefficients go through LDS -- over the sweep records
                    // nobody needs any more, [pair][segment][2 R] -- and leave as linear 16-byte-per-lane stores, whole lines
                    const int Mu = a.uniform;
                    double X[NX + 1][R];
#pragma unroll
                    for (int s = 0; s <= NX; ++s) {
#pragma unroll
                        for (int q = 0; q < R; ++q) X[s][q] = x0[q];
                        if (s < NX && s < mm) {
#pragma unroll
                            for (int q = 0; q < R; ++q)
                                X[s][q] = (s < NT) ? RL(s < NT ? s : 0, NL + q) : (NRS > 0 ? rr[(NRS > 0 && s >= NT && s < NT + NRS) ? s - NT : 0][NL + q] : RG(s < NT ? NT : s, NL + q));
                        }
#pragma unroll
                        for (int q = 1; q < R; q += 2) X[s][q] = isR ? -X[s][q] : X[s][q];
                    }
                    wave_lds_sync();
                    bool finite = true;
                    double* const stage = s_rec + (size_t)(lane >> 1) * Mu * NC;
#pragma unroll
                    for (int s = 1; s <= NX; ++s) {
                        const int j = mm - s;
                        if (s <= mm && emitp) {
                            const int sg = isR ? M - 1 - j : j;
                            double ys[ND], ye[ND], c[NC];
#pragma unroll
                            for (int d = 0; d < ND; ++d) { ys[d] = isR ? X[s - 1][d + 1] : X[s][d + 1]; ye[d] = isR ? X[s][d + 1] : X[s - 1][d + 1]; }
                            const double Tk = a.times[s0 + sg];
                            segment_coeffs_det<R>(isR ? X[s - 1][0] : X[s][0], ys, isR ? X[s][0] : X[s - 1][0], ye, Tk, fast_rcp(Tk), c);
                            finite = finite && (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
#pragma unroll
                            for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(stage + sg * NC + q) = make_double2(c[q], c[q + 1]);
                        }
                    }
                    if (emitp && !finite) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                    wave_lds_sync();
                    const int PP = Mu * R;                                   // 16-byte pieces per problem
                    const unsigned inv = 0xFFFFFFFFu / (unsigned)PP + 1u;    // t / PP = (t * inv) >> 32 for t < 2^16
                    const int gi = emitp ? (int)g : -1;
                    for (int t0 = 0; t0 < 32 * PP; t0 += 64) {
                        const int t = t0 + lane;
                        const int pr = (int)(((unsigned long long)(unsigned)t * inv) >> 32);
                        const int gp = __shfl(gi, 2 * (pr < 32 ? pr : 31), 64);
                        if (pr < 32 && gp >= 0) {
                            const int off = t - pr * PP;
                            const double2 v = *reinterpret_cast<const double2*>(s_rec + 2 * (size_t)t);
                            *reinterpret_cast<double2*>(aa.coeff + (size_t)gp * Mu * NC + 2 * off) = v;
                        }
                    }
                    wave_lds_sync();
                } else {
                    bool finite = true;
                    if (emitp && single) {
                        if (!isR) {      // no free knot: the polynomial of the boundary data
                            double ys[ND], ye[ND], c[NC];
                            const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
#pragma unroll
                            for (int d = 0; d < ND; ++d) { ys[d] = x0[d + 1]; ye[d] = bc[(ND + d) * 3]; }
                            const double Tk = a.times[s0];
                            segment_coeffs_det<R>(x0[0], ys, a.waypoints[base3 + 3 * M], ye, Tk, fast_rcp(Tk), c);
                            finite = (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
                            double* o = aa.coeff + ((size_t)3 * s0 + (size_t)ax * M) * NC;
#pragma unroll
                            for (int q = 0; q < NC; ++q) o[q] = c[q];
                        }
                    } else {
                        const int me = emitp ? mm : -1;
                        double xn[R];
#pragma unroll
                        for (int q = 0; q < R; ++q) xn[q] = 0.0;
                        // one slot of the walk from the meeting knot outwards: xo = the Hermite data of own knot m - s (original frame), xn = the previous
                        // slot's; own segment j = m - s joins the two
                        auto seg_step = [&](int s, const double (&xrec)[R]) __attribute__((always_inline)) {
                            const int j = mm - s;
                            double xo[R];
#pragma unroll
                            for (int q = 0; q < R; ++q) xo[q] = (s < mm) ? xrec[q] : x0[q];
#pragma unroll
                            for (int q = 1; q < R; q += 2) xo[q] = isR ? -xo[q] : xo[q];     // back to the original frame
                            if (s >= 1) {
                                const int sg = isR ? M - 1 - j : j;                           // original segment of own segment j
                                double ys[ND], ye[ND], c[NC];
#pragma unroll
                                for (int d = 0; d < ND; ++d) { ys[d] = isR ? xn[d + 1] : xo[d + 1]; ye[d] = isR ? xo[d + 1] : xn[d + 1]; }
                                const double Tk = a.times[s0 + sg];
                                segment_coeffs_det<R>(isR ? xn[0] : xo[0], ys, isR ? xo[0] : xn[0], ye, Tk, fast_rcp(Tk), c);
                                finite = finite && (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
                                double* o = aa.coeff + ((size_t)3 * s0 + (size_t)ax * M + sg) * NC;
                                if (al16) {
#pragma unroll
                                    for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(o + q) = make_double2(c[q], c[q + 1]);
                                } else {
#pragma unroll
                                    for (int q = 0; q < NC; ++q) o[q] = c[q];
                                }
                            }
#pragma unroll
                            for (int q = 0; q < R; ++q) xn[q] = xo[q];
                        };
                        if constexpr (NRS > 0) {
                            // (register slots: compile-time slot indices, static trip count; the lanes' own bounds are predicates)
#pragma unroll
                            for (int s = 0; s <= NT + NRS; ++s) {
                                if (s <= me) {
                                    double xr[R];
#pragma unroll
                                    for (int q = 0; q < R; ++q) xr[q] = s < NT ? RL(s < NT ? s : 0, NL + q) : rr[(s >= NT && s < NT + NRS) ? s - NT : 0][NL + q];
                                    seg_step(s, xr);
                                }
                            }
                        } else {
                            for (int s = 0; s <= me; ++s) {
                                double xr[R];
#pragma unroll
                                for (int q = 0; q < R; ++q) xr[q] = s < mm ? rec_ld(s, NL + q) : 0.0;
                                seg_step(s, xr);
                            }
                        }
                    }
                    if (emitp && !finite) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                }
            }
            if (emitp && !isR) {
                if (capped) atomicMin(&a.status[b], (int32_t)(fail != 0 ? fail : (int)UAVQP_MAX_ITER_REACHED));     // (UAVQP_PRIMAL_INFEASIBLE < UAVQP_MAX_ITER_REACHED: an infeasible axis decides; 0 is no status: a capped path that set no verdict reads "undecided")
                if (a.iters) atomicMax(&a.iters[b], (int32_t)it);
                if (a.active) {
                    unsigned long long* o = a.active + (size_t)g * (2 + 2 * K);
                    const unsigned long long valid = M >= 2 ? ((1ull << M) - 2ull) : 0ull;
                    o[0] = pin & ~eqmask & valid;
                    o[1] = upper & o[0];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        o[2 + 2 * j] = ract[j] & ~req[j];
                        o[3 + 2 * j] = rup[j] & o[2 + 2 * j];
                    }
                }
            }
        }
        if (finish) {
            g = -1;
            m = 0;
        }
        R2_CT(6);
    }
    R2_CT_FLUSH;
}

}  // namespace uavqp
