// qp_corridor.h -- corridor-constrained variant (north-star extension, SURVEY.md section 8-a'; BASELINE
// configs 3 and 5): the interior-waypoint equalities p_i(T_i) = w_{i+1} of the reference QP
// (minimum_control.cpp:34-42,118-124) become boxes lo <= p_i(T_i) <= hi -- the only inequality rows.
//
// In the Hermite variables the problem per axis is a strictly convex QP in x_k = (p_k, v_k, a_k[, j_k]) at
// the interior knots with an SPD block-tridiagonal Hessian (r x r blocks, function of the time
// allocation only) and simple bounds on the position components.  It is solved EXACTLY by a primal
// active-set method: pinned positions are eliminated symmetrically, every iteration is one block
// Thomas solve, a blocking bound is added on a partial step, the worst wrong-signed multiplier is
// released -- no ADMM tolerance, the result is the QP's minimiser to rounding (OSQP in the reference
// formulation converges to the same point, which is how the tests check it).
//
// Round-2 layout (round 1 kept the sweep state of every knot in an HBM workspace and moved 26-38x the algorithmic bytes):
//   * TWO lanes per (trajectory, axis) problem -- "twisted" two-sided block elimination, as in qp_twisted.h: lane L eliminates
//     the knots 1..c-1 in the forward direction, lane R the time-reversed problem (derivative d picks up (-1)^d, positions and
//     their bounds are unchanged) from knot M-1 down to c+1 with the SAME instruction stream; the two partial Schur
//     complements of the meeting knot c = ceil(M/2) are exchanged through DPP (lane ^ 1), both lanes solve it and
//     back-substitute their own half.  Half the sequential depth, half the per-lane state.
//   * the per-knot sweep state (LDL' factors, h_k / x_k, the feasible iterate z_k) of the first NT own knots of a lane lives
//     in LDS: 8 knots x 10 doubles x 64 lanes = 40 KiB per single-wave workgroup for r = 3, i.e. FOUR waves per CU = one per
//     SIMD with the whole state of a 16-segment problem on chip (config 3: no workspace traffic at all).  Longer halves keep
//     the remaining knots in an HBM workspace [wave][own knot][field][lane] as before (r = 4: NT = 5).
//   * persistent waves pull problems from a global work counter: a lane pair that has converged hands over its Hermite solution
//     and takes the next problem at once, so a wave no longer runs as long as its slowest lane (mean 11 iterations per problem,
//     19.4 mean over the waves' maxima on config 3).  Lane pairs of one wave are therefore at different iterations of different
//     problems; every iteration is the same instruction stream (one forward, one backward sweep), so nothing diverges.
//   * the solver writes the HERMITE solution (r doubles per interior knot and axis); corridor_emit_kernel turns it into the
//     reference's monomial coefficients with one lane per (trajectory, axis, segment), fully coalesced.
// Checked against exact-rational fixtures including the active sets (tests/test_corridor_golden.py), the OSQP-faithful port and
// a KKT certificate (tests/test_gpu_corridor.py), at BASELINE sizes in tests/test_gpu_baseline_sizes.py.
#pragma once
#include "qp_core_kernels.h"
#include "qp_wave_utils.h"

namespace uavqp {

constexpr int corridor_gcache_stride = 24 * 24 + 2 * 32 * 4;   // G of up to 24 rows, then the two vector families of up to 32 columns (r <= 4)

struct CorridorArgs {
    int n_traj, uniform, max_segments, max_iter, pdas_rounds;
    const int32_t* seg_offsets;
    const double* waypoints;
    const double* times;
    const double* bc;
    const double* corr_lo;
    const double* corr_hi;
    double* coeff;
    int32_t* status;  // pre-filled with UAVQP_SOLVED; failing axes atomicMin their code in
    int32_t* iters;   // pre-filled with 0; atomicMax over axes (may be null)
    double* ws;       // HBM part of the sweep state: [wave][own knot beyond NT][field][lane] (null when every half fits LDS)
    int ws_knots;     // own knots per lane held in the workspace
    double* xsol;     // [waypoint row][axis][r]: Hermite solution at the interior knots (hand-off to corridor_emit_kernel)
    unsigned long long* desc;      // [n_traj][3]: bit 0 = solve this problem (valid, M >= 2), bits 1..M-1 = equality rows (corridor_prep_kernel)
    unsigned int* queue;           // work counter, zeroed before the launch
    const int32_t* order;          // optional dealing order of the trajectories (null = index order)
    unsigned long long* active;    // [n_traj][3][2] working set in/out (may be null)
    int warm;                      // 1: read `active` as the initial working set; 2: also start the free positions from the knot positions of the
                                   // polynomials found in `coeff` (the previous solve's output, clipped into the boxes) instead of the waypoints
    unsigned long long* guess;     // [n_traj][3][2] cold start: the closed-form starting set of corridor_prep_kernel (may be null)
    const int32_t* only_i32;       // optional masks of an outer loop (uavqp_pipeline.h): when either is given, only the trajectories with a non-zero
    const unsigned char* only_u8;  // entry in one of them take part; the others keep their coefficients, status, iteration count and working set
    const int* n_active;           // with a mask: `order` lists only the trajectories that take part, *n_active of them (compact_order_kernel); null: all n_traj
    // G of the dual prelude across the solves of an outer loop whose durations only change by ONE factor per trajectory (the time
    // re-allocation): [H^-1]_{(i,a),(j,b)}(s T) = s^(2r-1-a-b) [H^-1]_{(i,a),(j,b)}(T).  mode 1: build as usual and store; 2: load and rescale
    // by gscale[b] (the factor since the store) instead of running the chain; 0: off
    double* gcache;                // [n_traj][corridor_gcache_stride]
    const double* gscale;          // [n_traj]
    int gcache_mode;
    int prep_in_dual;              // 1: no corridor_reset_kernel / corridor_prep_kernel launch: the dual prelude validates, resets and describes what it visits
    int fused_emit;                // 1: corridor_solve_kernel writes the polynomials itself when a problem is done (and corridor_prep_kernel those of the
                                   // one-segment trajectories); 0: it leaves the Hermite solution in xsol for corridor_emit_kernel (the rows solvers' path)
    int guess_closed_form;         // 1: corridor_prep_kernel fills `guess` with the closed-form set; 0: it only zeroes it (corridor_dual_kernel, qp_corridor_dual.h, writes it)
#ifdef UAVQP_DUAL_DEBUG
    double* dbg;                   // debug build only (tools/corridor_dual_gpu_probe.py): G, unconstrained minimisers, trip counts of the first trajectories
#endif
#ifdef UAVQP_CORRIDOR_TIMING
    long long* stamps;             // debug build only (tools/): cycles per section of wave 0 -> [refill, forward, meeting, backward, decide, hand-over, iterations]
#endif
};

#ifdef UAVQP_CORRIDOR_TIMING
#define UAVQP_CT_DECL long long ct_acc[7] = {0, 0, 0, 0, 0, 0, 0}; long long ct_t = __builtin_readcyclecounter();
#define UAVQP_CT(k) do { const long long n_ = __builtin_readcyclecounter(); ct_acc[k] += n_ - ct_t; ct_t = n_; } while (0)
#define UAVQP_CT_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0 && a.stamps) for (int k_ = 0; k_ < 7; ++k_) a.stamps[k_] = ct_acc[k_]; } while (0)
#else
#define UAVQP_CT_DECL
#define UAVQP_CT(k) do {} while (0)
#define UAVQP_CT_FLUSH do {} while (0)
#endif

// r x r blocks of one segment including the position component (index 0):
//   B11 end/end = T^(a+b+1-2R) W[a][b],  B00 start/start = (-1)^(a+b) B11,  B01 start/end = -T^(a+b+1-2R) V[a][b]
template <int R>
struct FullBlocks {
    double B11[R][R];
    double B01[R][R];
    __device__ __forceinline__ void build(double T) {
        const double it = fast_rcp(T);
        double ip[2 * R];
        ip[0] = 1.0;
#pragma unroll
        for (int j = 1; j < 2 * R; ++j) ip[j] = ip[j - 1] * it;
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) {
                const double p = ip[2 * R - 1 - a - b];
                B11[a][b] = p * Tab<R>::W(a, b);
                B01[a][b] = -p * Tab<R>::V(a, b);
            }
    }
    __device__ __forceinline__ double B00(int i, int c) const { return ((i + c) & 1) ? -B11[i][c] : B11[i][c]; }
};

#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
#endif
// does trajectory b take part in this solve?  (no mask: all do)
__device__ __forceinline__ bool corridor_takes_part(const int32_t* only_i32, const unsigned char* only_u8, int b) {
    if (!only_i32 && !only_u8) return true;
    return (only_i32 && only_i32[b] != 0) || (only_u8 && only_u8[b] != 0);
}
// Dealing order of a masked re-solve: the entries of `order` (null: 0, 1, 2, ...) whose trajectory takes part, in the same sequence (the
// order is by segment count: waves keep trajectories of similar length), and their number.  One workgroup: a block scan over n_traj flags.
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void compact_order_kernel(const int32_t* __restrict__ order, int n_traj, const int32_t* __restrict__ only_i32,
                                                             const unsigned char* __restrict__ only_u8, int32_t* __restrict__ out, int* __restrict__ n_out,
                                                             const unsigned int* __restrict__ none_if_zero = nullptr,
                                                             const int* __restrict__ n_dev = nullptr, volatile unsigned long long* host_slot = nullptr,
                                                             unsigned int host_seq = 0u) {
    // (host_slot: a word of host-coherent pinned memory that receives (host_seq << 32 | count) in one store -- the pipeline's host loop
    // polls it for the sequence number of the round: no copy command and no event packet between the kernels of two rounds)
    // (a caller that counted the participating trajectories while it flagged them passes the count: nothing to scan when it is zero --
    // the rows solve, whose prelude usually takes every trajectory and leaves the box phase none)
    if (none_if_zero && *none_if_zero == 0u) {
        if (threadIdx.x == 0) {
            *n_out = 0;
            if (host_slot) { __threadfence_system(); *host_slot = (unsigned long long)host_seq << 32; __threadfence_system(); }
        }
        return;
    }
    // (n_dev: `order` is itself a compacted list whose length is on the device -- the pipeline's later rounds compact the previous
    // round's list, not the whole batch; `out` must then be another buffer)
    if (n_dev) n_traj = *n_dev;
    // wave w takes the contiguous slice [w per, (w + 1) per) in sub-chunks of 64 (all loads of a lane in flight together: two round trips
    // per pass of 16 sub-chunks, not one per element), ranks by ballot; the 16 wave totals are scanned through LDS
    __shared__ int s_tot[16];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = ((n_traj + 15) / 16 + 63) / 64 * 64, w0 = w * per, w1 = min(w0 + per, n_traj);
    constexpr int CH = 16;
    if (per <= 64 * CH) {
        // the whole slice of a wave is one chunk (n_traj <= 16384): indices and flags stay in registers between counting and writing
        int bidx[CH];
        bool keep[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = w0 + 64 * c + lane;
            bidx[c] = i < w1 ? (order ? order[i] : i) : -1;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) keep[c] = bidx[c] >= 0 && corridor_takes_part(only_i32, only_u8, bidx[c]);
        unsigned long long m[CH];
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            m[c] = __ballot(keep[c]);
            cnt += __popcll(m[c]);
        }
        if (lane == 0) s_tot[w] = cnt;
        __syncthreads();
        int base = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = s_tot[k];
            base += k < w ? t : 0;
            all += t;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (keep[c]) out[base + __popcll(m[c] & ((1ull << lane) - 1ull))] = bidx[c];
            base += __popcll(m[c]);
        }
        if (threadIdx.x == 0) {
            *n_out = all;
            if (host_slot) { __threadfence_system(); *host_slot = ((unsigned long long)host_seq << 32) | (unsigned)all; __threadfence_system(); }
        }
        return;
    }
    int tot = 0;
    for (int pass = 0; pass < 2; ++pass) {      // pass 0: count, pass 1: write
        int base = 0;
        if (pass == 1) {
            for (int k = 0; k < w; ++k) base += s_tot[k];
        }
        for (int c0 = w0; c0 < w1; c0 += 64 * CH) {
            int bidx[CH];
            bool keep[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int i = c0 + 64 * c + lane;
                bidx[c] = i < w1 ? (order ? order[i] : i) : -1;
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) keep[c] = bidx[c] >= 0 && corridor_takes_part(only_i32, only_u8, bidx[c]);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const unsigned long long m = __ballot(keep[c]);
                if (pass == 1 && keep[c]) out[base + __popcll(m & ((1ull << lane) - 1ull))] = bidx[c];
                base += __popcll(m);
            }
        }
        if (pass == 0) {
            tot = base;
            if (lane == 0) s_tot[w] = tot;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        int all = 0;
        for (int k = 0; k < 16; ++k) all += s_tot[k];
        *n_out = all;
        if (host_slot) { __threadfence_system(); *host_slot = ((unsigned long long)host_seq << 32) | (unsigned)all; __threadfence_system(); }
    }
}
#endif
// The same compaction for lists beyond one workgroup's reach (16 384 entries): block b takes entries [16384 b, 16384 (b + 1)) -- phase 0
// leaves the number it keeps in block_counts[b], phase 1 (a second launch of the same grid) writes them behind those of the blocks before it.
// Order preserved; block 0 of phase 1 reports the total.  (one workgroup walked a million flags per pipeline round.)
constexpr int COMPACT_BLOCK = 16384;
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void compact_order_blocks_kernel(const int32_t* __restrict__ order, int n_traj, const int32_t* __restrict__ only_i32,
                                                                    const unsigned char* __restrict__ only_u8, int32_t* __restrict__ out, int* __restrict__ n_out,
                                                                    const int* __restrict__ n_dev, int* __restrict__ block_counts, int phase,
                                                                    volatile unsigned long long* host_slot, unsigned int host_seq) {
    if (n_dev) n_traj = *n_dev;
    __shared__ int s_tot[16], s_before[16], s_all[16];
    constexpr int CH = 16;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w0 = (int)blockIdx.x * COMPACT_BLOCK + w * 64 * CH, w1 = min(w0 + 64 * CH, n_traj);
    int bidx[CH];
    bool keep[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = w0 + 64 * c + lane;
        bidx[c] = i < w1 ? (order ? order[i] : i) : -1;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) keep[c] = bidx[c] >= 0 && corridor_takes_part(only_i32, only_u8, bidx[c]);
    unsigned long long m[CH];
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        m[c] = __ballot(keep[c]);
        cnt += __popcll(m[c]);
    }
    if (lane == 0) s_tot[w] = cnt;
    if (phase == 1) {
        // what the blocks before this one keep, and what all keep (the grid has at most 1024 blocks: one count per thread)
        int before = 0, all = 0;
        if (threadIdx.x < gridDim.x) {
            const int v = block_counts[threadIdx.x];
            all = v;
            before = (int)threadIdx.x < (int)blockIdx.x ? v : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += __shfl_xor(before, d, 64);
            all += __shfl_xor(all, d, 64);
        }
        if (lane == 0) { s_before[w] = before; s_all[w] = all; }
    }
    __syncthreads();
    if (phase == 0) {
        if (threadIdx.x == 0) {
            int t = 0;
            for (int k = 0; k < 16; ++k) t += s_tot[k];
            block_counts[blockIdx.x] = t;
        }
        return;
    }
    int base = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        base += s_before[k] + (k < w ? s_tot[k] : 0);
        all += s_all[k];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (keep[c]) out[base + __popcll(m[c] & ((1ull << lane) - 1ull))] = bidx[c];
        base += __popcll(m[c]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *n_out = all;
        if (host_slot) { __threadfence_system(); *host_slot = ((unsigned long long)host_seq << 32) | (unsigned)all; __threadfence_system(); }
    }
}
#endif
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ void corridor_reset_kernel(int32_t* status, int32_t* iters, int n, const int32_t* only_i32, const unsigned char* only_u8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && corridor_takes_part(only_i32, only_u8, i)) {
        status[i] = (int32_t)UAVQP_SOLVED;
        if (iters) iters[i] = 0;
    }
}
#endif

__device__ __forceinline__ int swap_pair_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned long long swap_pair_u64(unsigned long long v) {
    const unsigned lo = (unsigned)swap_pair_i((int)(unsigned)v), hi = (unsigned)swap_pair_i((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Dealing order of a RAGGED batch: trajectories by descending segment count (global counting sort: histogram, scan, scatter).
// A wave sweeps as long as its longest half, so problems of similar length should share a wave; and the longest problems --
// which also need the most iterations -- start first instead of stretching the tail of the launch.  The order inside a bin is
// whatever the atomics give: it decides which lane pair solves a problem, never its result.
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void seg_hist_kernel(const int32_t* __restrict__ seg_offsets, int n_traj, int* __restrict__ hist) {
    __shared__ int s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n_traj; b += gridDim.x * blockDim.x) {
        int M = seg_offsets[b + 1] - seg_offsets[b];
        M = M < 0 ? 0 : (M > 255 ? 255 : M);
        atomicAdd(&s_h[M], 1);
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
#endif
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void seg_scan_kernel(const int* __restrict__ hist, int* __restrict__ cursor) {
    __shared__ int s_c[256];
    s_c[threadIdx.x] = hist[255 - threadIdx.x];   // position t <-> segment count 255 - t: longest first
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int c = s_c[t]; s_c[t] = run; run += c; }
    }
    __syncthreads();
    cursor[255 - threadIdx.x] = s_c[threadIdx.x];
}
#endif
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void seg_scatter_kernel(const int32_t* __restrict__ seg_offsets, int n_traj, int* __restrict__ cursor,
                                                          int32_t* __restrict__ order) {
    // a batch has few distinct segment counts: one global add per trajectory on ~20 addresses serialises (38 us for 16 384
    // trajectories); the block counts its contiguous slice in LDS, reserves a range per non-empty bin with ONE global add and hands
    // out the positions with LDS atomics
    __shared__ int s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int per = (n_traj + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = min(b0 + per, n_traj);
    auto key = [&](int b) -> int {
        const int M = seg_offsets[b + 1] - seg_offsets[b];
        return M < 0 ? 0 : (M > 255 ? 255 : M);
    };
    for (int b = b0 + threadIdx.x; b < b1; b += 256) atomicAdd(&s_h[key(b)], 1);
    __syncthreads();
    {
        const int c = s_h[threadIdx.x];
        s_h[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0;
    }
    __syncthreads();
    for (int b = b0 + threadIdx.x; b < b1; b += 256) order[atomicAdd(&s_h[key(b)], 1)] = b;
}
#endif

// One lane per (trajectory, axis): validation of the inputs and the permanent pins (lo == hi: a true equality row, as in the
// reference), once per solve and off the solver's critical path -- a persistent wave that takes a new problem must not stall
// the other 31 problems of the wave behind serial validation loads.
//
// It also makes the COLD start's working set (a.guess): the polynomial of degree 2r - 1 that only matches the two end states -- the
// minimiser if there were no waypoints at all -- evaluated at the knot times; a knot whose box it misses is guessed active on
// that side.  A crude guess (it over-activates: 11 of 15 knots on config 3 where the solution has 5, half of the knot decisions
// wrong), yet the active-set method converges faster from it than from the empty set: 13.79 -> 12.46 iterations mean on config 3
// (tools/corridor_warm_guess_probe.py; guesses from the neighbouring waypoints are closer to the solution's set and save
// nothing).  Any set is an admissible start, the result is the same to the last bit.
template <int R>
__global__ __launch_bounds__(256) void corridor_prep_kernel(CorridorArgs a) {
    constexpr int ND = R - 1, NC = 2 * R;
    const long long total = (long long)a.n_traj * 3;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(q / 3), ax = (int)(q - 3LL * b);
        if (!corridor_takes_part(a.only_i32, a.only_u8, b)) {   // not part of this solve: no problem, nothing reported
            a.desc[q] = 0ull;
            if (a.guess) { a.guess[2 * q] = 0ull; a.guess[2 * q + 1] = 0ull; }
            continue;
        }
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        const long long base3 = 3LL * ((long long)s0 + b) + ax;
        const double* lo = a.corr_lo + base3;
        const double* hi = a.corr_hi + base3;
        const double* T = a.times + s0;
        bool ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments) && M <= 63;  // pin masks are 64-bit
        unsigned long long eq = 0ull, g_act = 0ull, g_up = 0ull;
        double Ttot = 0.0;
        if (ok) {
            for (int i = 0; i < M; ++i) {
                const double t = T[i];
                ok = ok & (t > 0.0) & (t < INFINITY);
                Ttot += t;
            }
        }
        if (ok) {
            // one pass over the boxes: bounds check, equality rows, and the starting set
            double c[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) c[j] = 0.0;
            const bool want = a.guess != nullptr && a.guess_closed_form && M >= 2;
            if (want) {
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                double ys[ND], ye[ND];
#pragma unroll
                for (int d = 0; d < ND; ++d) { ys[d] = bc[d * 3]; ye[d] = bc[(ND + d) * 3]; }
                segment_coeffs<R>(a.waypoints[base3], ys, a.waypoints[base3 + 3 * M], ye, Ttot, fast_rcp(Ttot), c);
            }
            double t = 0.0;
            for (int k = 1; k < M; ++k) {
                const double l = lo[3 * k], h = hi[3 * k];
                ok = ok & (l <= h);
                eq |= (unsigned long long)(l == h) << k;
                t += T[k - 1];
                double p = c[NC - 1];
#pragma unroll
                for (int j = NC - 2; j >= 0; --j) p = fma(p, t, c[j]);
                const bool above = p > h, below = p < l;
                g_act |= (unsigned long long)(above | below) << k;
                g_up |= (unsigned long long)above << k;
            }
            g_act &= want ? ~eq : 0ull;
            g_up &= g_act;
        }
        if (!ok) atomicMin(&a.status[b], (int32_t)UAVQP_INVALID_INPUT);
        // no interior knot (M = 1): nothing to solve, the segment follows from the boundary data (here, or in the emission kernel)
        if (ok && M == 1 && a.active) { a.active[2 * q] = 0ull; a.active[2 * q + 1] = 0ull; }
        if (ok && M == 1 && a.fused_emit) {   // (validity of a one-segment trajectory is its duration: the same verdict on all three axes)
            const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
            double ys[ND], ye[ND], c1[NC];
#pragma unroll
            for (int d = 0; d < ND; ++d) { ys[d] = bc[d * 3]; ye[d] = bc[(ND + d) * 3]; }
            const double Tk = T[0];
            segment_coeffs_det<R>(a.waypoints[base3], ys, a.waypoints[base3 + 3], ye, Tk, fast_rcp(Tk), c1);
            if (!((fabs(c1[NC - 1]) < INFINITY) && (fabs(c1[R]) < INFINITY))) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
            double* o = a.coeff + ((size_t)3 * s0 + ax) * NC;
#pragma unroll
            for (int j = 0; j < NC; ++j) o[j] = c1[j];
        }
        a.desc[q] = (ok && M >= 2) ? (eq | 1ull) : 0ull;
        if (a.guess) {
            a.guess[2 * q] = (ok && M >= 2) ? g_act : 0ull;
            a.guess[2 * q + 1] = (ok && M >= 2) ? g_up : 0ull;
        }
    }
}

#ifndef UAVQP_CORRIDOR_WAVES_PER_CU
#define UAVQP_CORRIDOR_WAVES_PER_CU 4   // single-wave workgroups per CU = 160 KiB of LDS / the state kept on chip per wave (4: one wave per SIMD)
#endif
constexpr int corridor_waves_per_cu() { return UAVQP_CORRIDOR_WAVES_PER_CU; }
// own knots per lane whose record (F = r (r + 1) / 2 + r + 1 doubles) stays in LDS: 40 KiB per wave -> 8 (r = 3) / 5 (r = 4).
// WPC = waves per CU the kernel is built for; the host picks 2 (twice the knots on chip, half the waves) for small batches of long
// r = 4 problems, whose solve time is the slowest problem's iteration count times the duration of ONE iteration.
constexpr int corridor_lds_knots(int R, int WPC = UAVQP_CORRIDOR_WAVES_PER_CU) { return (160 * 1024 / WPC) / (64 * 8 * (R * (R + 1) / 2 + R + 1)); }

// wave-uniform maximum of a small non-negative per-lane integer (< 64): six ballots
__device__ __forceinline__ int wave_max_small(int v) {
    int r = 0;
#pragma unroll
    for (int bit = 5; bit >= 0; --bit) {
        const int t = r | (1 << bit);
        if (__ballot(v >= t) != 0ull) r = t;
    }
    return r;
}

__device__ __forceinline__ double sel(bool c, double a, double b) { return c ? a : b; }

// Every active-set iteration is exactly ONE forward and ONE backward pass over the own knots of a lane; everything else is
// folded into them, and every state access of a pass is issued one knot ahead of its use:
//   * the update of the feasible iterate z decided by the previous iteration (block-pivot clip, partial step of
//     the ratio test, full step) is applied lazily in the forward sweep, one knot ahead of the elimination;
//   * the decisions of the iteration -- block-pivot sets, ratio test over the free positions, multipliers of the
//     active bounds (row 0 of the unmasked block row, finished one knot late when x_{k-1} appears) -- are
//     accumulated in the backward sweep and combined across the lane pair.
// Own frame of a lane: own knot j = 0 is its boundary knot (knot 0 for L, knot M for R), own knot m is the meeting knot
// (m_L = ceil(M/2), m_R = floor(M/2)); own segment j joins own knots j and j+1.  Working-set masks are kept in ORIGINAL knot
// numbering (bit k = interior waypoint k) and are identical in both lanes of a pair.
//
// The lane pairs of a wave work on different problems at different iterations, and on ragged batches on halves of different
// length, so every data-dependent choice (pinned or free, which z update is pending, ...) is a SELECT, not a branch: the knot
// steps are single basic blocks the scheduler can interleave freely -- with one wave per SIMD nothing else hides a stall.
// Sweeps are aligned at the MEETING knot: state slot s holds own knot m - s, the forward sweep of a shorter half starts late,
// the backward sweep ends early; slot numbers (and with them the LDS-or-workspace test, slot < NT) are wave-uniform.
// WS: some halves are longer than NT knots, their far slots live in the HBM workspace (one uniform branch per record access);
// WS = false compiles every such test away (config 3: the whole state is in LDS).
template <int R, bool WS, int WPC = UAVQP_CORRIDOR_WAVES_PER_CU>
__global__ __launch_bounds__(64, (WPC > 4 ? 2 : 1)) void corridor_solve_kernel(CorridorArgs a) {
    constexpr int ND = R - 1;
    constexpr int NT = corridor_lds_knots(R, WPC);
    // sweep state per own knot: LDL' factors of S_j (strict lower triangle + inverse pivots: R (R + 1) / 2 numbers), x_j (first
    // h_j, overwritten by the solution in the backward sweep) and the current position iterate z_j.  E_j = S_j^-1 M_j is NOT
    // stored: it is re-derived where needed.  (Explicit inverses -- cofactors / 2 x 2 blocks instead of the factorisation, one
    // reciprocal on the dependency chain -- were measured: no faster, 1.693 vs 1.701 ms on config 3, and the 4 x 4 case lost
    // the 1e-10 agreement with the exact-rational fixtures.)
    using Inv = SmallLDL<R>;
    using IP = LDLPack<R>;
    constexpr int NE = IP::NE;
    constexpr int F_I = 0, F_X = NE, F_Z = NE + R, F = NE + R + 1;
    __shared__ double s_rec[NT * F * 64];
    const int lane = threadIdx.x;
    const int isR = lane & 1;
    double* const ws = a.ws + (size_t)blockIdx.x * (size_t)a.ws_knots * F * 64 + lane;  // only touched for slots >= NT
    // record accessors: slot s is wave-uniform, so "LDS or workspace" is ONE scalar branch per record, not one per field
    auto in_lds = [&](int s) -> bool { return !WS || s < NT; };
    auto L = [&](int s, int f) -> double& { return s_rec[(s * F + f) * 64 + lane]; };
    auto G = [&](int s, int f) -> double& { return ws[((size_t)(s - NT) * F + f) * 64]; };
    auto ld_xz = [&](int s, double& x, double& z) {
        if (in_lds(s)) { x = L(s, F_X); z = L(s, F_Z); } else { x = G(s, F_X); z = G(s, F_Z); }
    };
    auto ld_rec = [&](int s, double (&r)[F]) {
        if (in_lds(s)) {
#pragma unroll
            for (int f = 0; f < F; ++f) r[f] = L(s, f);
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) r[f] = G(s, f);
        }
    };
    auto st_x = [&](int s, const double (&x)[R]) {
        if (in_lds(s)) {
#pragma unroll
            for (int q = 0; q < R; ++q) L(s, F_X + q) = x[q];
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) G(s, F_X + q) = x[q];
        }
    };
    auto st_xz = [&](int s, const double (&x)[R], double z) {
        st_x(s, x);
        if (in_lds(s)) L(s, F_Z) = z; else G(s, F_Z) = z;
    };
    auto st_rec = [&](int s, const Inv& inv, const double (&h)[R], double z) {
        double r[F], e[NE];
        IP::get(inv, e);
#pragma unroll
        for (int i = 0; i < NE; ++i) r[F_I + i] = e[i];
#pragma unroll
        for (int i = 0; i < R; ++i) r[F_X + i] = h[i];
        r[F_Z] = z;
        if (in_lds(s)) {
#pragma unroll
            for (int q = 0; q < F; ++q) L(s, q) = r[q];
        } else {
#pragma unroll
            for (int q = 0; q < F; ++q) G(s, q) = r[q];
        }
    };

    const long long total = (long long)(a.n_active ? *a.n_active : a.n_traj) * 3;
#ifdef UAVQP_DUAL_DEBUG
    if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) {   // (tools/pipeline_round_probe.py: when every verifying solve of a pipeline call started)
        if (total >= 4500 && total <= 9000) a.dbg[5 * 16384 + 1] = (double)wall_clock64();
        const unsigned slot = atomicAdd(reinterpret_cast<unsigned int*>(a.dbg + 63 * 2048 + 1400 + 9), 1u);
        if (slot < 16) { a.dbg[63 * 2048 + 1400 + 64 + 2 * slot] = (double)wall_clock64(); a.dbg[63 * 2048 + 1400 + 65 + 2 * slot] = (double)total; }
    }
#endif
    bool queue_empty = false;
    UAVQP_CT_DECL

    // ---- state of the problem this lane pair works on (identical in both lanes unless noted)
    long long g = -1;            // problem = 3 * trajectory + axis, -1: none
    int b = 0, M = 0, m = 0;     // m: own knots up to and including the meeting knot (lane-specific for odd M)
    long long base3 = 0;         // bounds / Hermite solution of ORIGINAL knot k of this axis at index base3 + 3 k
    int s0 = 0;                  // first segment of the trajectory in `times`
    double x0[R];                // Hermite data of the own boundary knot, own frame (lane-specific)
#pragma unroll
    for (int i = 0; i < R; ++i) x0[i] = 0.0;
    unsigned long long eqmask = 0ull, pin = 0ull, upper = 0ull;
    int it = 0, pdas_left = 0;
    bool final_pass = false;
    // pending update of z, applied by the next forward sweep (zmode 0: none; 1: block-pivot round; 2: partial step
    // of length zalpha blocked at knot zblock; 3: full step; 4: first sweep of a problem -- z from the clipped initial guess)
    int zmode = 0, zblock = -1;
    bool zblock_upper = false;
    unsigned long long zpin = 0ull;
    double zalpha = 1.0;

    // original knot of own knot j / duration of own segment j, indices clamped into the trajectory (speculative loads of the
    // branch-free sweeps stay in bounds; their values are discarded by the selects)
    auto korig = [&](int j) -> int {
        const int jc = j < 0 ? 0 : (j > M ? M : j);
        return isR ? M - jc : jc;
    };
    auto bit = [&](unsigned long long mask, int k) -> bool { return (mask >> (k & 63)) & 1ull; };

    for (;;) {
        // ================= hand out problems to the free lane pairs =================
        {
            const bool need = g < 0;
            const unsigned long long needm = __ballot(need);
            if (needm != 0ull && !queue_empty) {
                const int npairs = __popcll(needm) >> 1;
                const int leader = (int)__builtin_ctzll(needm);
                unsigned int qb = 0;
                if (lane == leader) qb = atomicAdd(a.queue, (unsigned int)npairs);
                qb = __shfl(qb, leader, 64);
                if ((long long)qb + npairs >= total) queue_empty = true;
                if (need) {
                    const int rank = __popcll(needm & ((1ull << lane) - 1ull)) >> 1;  // pairs in front of this one
                    const long long q = (long long)qb + rank;
                    if (q < total) {
                        const int bq = (int)(q / 3), ax = (int)(q - 3LL * bq);
                        const int bn = a.order ? a.order[bq] : bq;
                        const long long gn = 3LL * bn + ax;
                        // everything a new problem needs is fetched by INDEPENDENT loads (one round trip); validation and
                        // the equality rows come ready-made from corridor_prep_kernel
                        unsigned long long dsc = a.desc[gn];
                        // (a trajectory with an invalid axis is left untouched as a whole: with the emission in this kernel its other axes must not be solved)
                        const int32_t st_in = a.fused_emit ? a.status[bn] : (int32_t)UAVQP_SOLVED;
                        if (st_in == (int32_t)UAVQP_INVALID_INPUT) dsc = 0ull;
                        int sn, Mn;
                        if (a.uniform > 0) { Mn = a.uniform; sn = bn * Mn; } else { sn = a.seg_offsets[bn]; Mn = a.seg_offsets[bn + 1] - sn; }
                        unsigned long long wpin = 0ull, wupper = 0ull;
                        if (a.active && a.warm) { wpin = a.active[2 * gn]; wupper = a.active[2 * gn + 1]; }
                        else if (a.guess) { wpin = a.guess[2 * gn]; wupper = a.guess[2 * gn + 1]; }
                        if (dsc & 1ull) {
                            b = bn; M = Mn; s0 = sn;
                            base3 = 3LL * ((long long)sn + bn) + ax;
                            g = gn;
                            m = isR ? M / 2 : (M + 1) / 2;
                            const double* bc = a.bc + (size_t)bn * 2 * ND * 3 + ax;
                            // own boundary knot in the own frame: derivative d of the reversed problem picks up (-1)^d
                            x0[0] = a.waypoints[base3 + (isR ? 3 * M : 0)];
#pragma unroll
                            for (int d = 0; d < ND; ++d) {
                                const double v = bc[((isR ? ND : 0) + d) * 3];
                                x0[d + 1] = (isR && ((d & 1) == 0)) ? -v : v;
                            }
                            // permanent pins and the first working set.  A warm start only supplies that set (bounds guessed
                            // active sit on their bound); wrong guesses are repaired by the iterations below like any other
                            // intermediate working set.
                            const unsigned long long valid = (1ull << M) - 2ull;  // bits 1..M-1
                            eqmask = dsc & valid;
                            pin = eqmask | (wpin & valid);
                            upper = wupper & wpin & valid & ~eqmask;
                            it = 0;
                            pdas_left = a.pdas_rounds;
                            final_pass = false;
                            // the first forward sweep (zmode 4) makes the initial feasible iterate z from the waypoints
                            zmode = 4; zblock = -1; zblock_upper = false; zpin = pin; zalpha = 1.0;
                        }
                    }
                }
            }
        }
        const bool act = g >= 0;
        UAVQP_CT(0);
        if (__ballot(act) == 0ull) {
            if (queue_empty) break;
            continue;
        }
        const int mm = act ? m : 0;              // own knots of this lane: a free lane sweeps nothing
        const int mmax = wave_max_small(mm);     // the wave sweeps as long as its longest half
        const int off = mmax - mm;

        const double* const LO = a.corr_lo + base3;
        const double* const HI = a.corr_hi + base3;
        const double* const TT = a.times + s0;
        const double* const WP = a.waypoints + base3;
        // initial positions of a problem's first sweep: the waypoints, or (warm = 2) the start of every segment of the polynomials the
        // caller left in `coeff` -- c_0 of segment k of this axis = p(knot k); whatever is found there is clipped into the box first
        const bool zprev = a.warm == 2;
        const int axis = (int)(base3 - 3LL * ((long long)s0 + b));
        const double* const ZS = zprev ? a.coeff + ((size_t)3 * s0 + (size_t)axis * M) * (2 * R) : WP;
        const int zstride = zprev ? 2 * R : 3;
        // slot s <-> original knot kbase + ksign s, original segment of own segment (m - s) = tbase + ksign s
        const int ksign = isR ? 1 : -1, kbase = isR ? M - mm : mm, tbase = isR ? M - 1 - mm : mm;
        auto kslot = [&](int s) -> int { return kbase + ksign * s; };
        auto kclamp = [&](int k) -> int { return min(max(k, 0), M); };                 // v_med3: speculative loads stay in bounds
        auto tclamp = [&](int t) -> int { return max(min(t, M - 1), 0); };
        // working-set masks in SLOT order (bit s = own knot m - s): the sweeps test them with wave-uniform shift counts
        auto toslot = [&](unsigned long long msk) -> unsigned long long {
            return isR ? (msk >> ((M - mm) & 63)) : (__brevll(msk) >> ((63 - mm) & 63));
        };
        auto fromslot = [&](unsigned long long msk) -> unsigned long long {
            return isR ? (msk << ((M - mm) & 63)) : __brevll(msk << ((63 - mm) & 63));
        };
        const unsigned long long s_pin = toslot(pin), s_zpin = toslot(zpin), s_eq = toslot(eqmask), s_up = toslot(upper);
        const int zslot = act ? (isR ? zblock - (M - mm) : mm - zblock) : -1;  // slot of the blocking knot (negative / beyond m: the other half's)

        // z update of the knot in slot s (znew of the header comment), select form.
        //   1: block-pivot round -- every (newly) pinned position sits on its bound, the free ones keep a feasible iterate for the
        //      safe phase: the clipped subspace minimiser;  2: partial step, the blocking knot lands on its bound;  3: full step;
        //   4: first sweep -- x0old is the waypoint w; equality rows and free positions start at the clipped waypoint, bounds guessed
        //      active by a warm start at that bound;  0: nothing pending
        const bool m14 = (zmode == 1) || (zmode == 4), m2 = zmode == 2, m0 = zmode == 0, m4 = zmode == 4;
        auto znew = [&](int s, double xold, double zold, double l, double h, double w) -> double {
            const bool zp = (s_zpin >> s) & 1ull, eq = (s_eq >> s) & 1ull, up = (s_up >> s) & 1ull;
            const double x0old = m4 ? w : xold;   // first sweep of a problem: the slot still holds the previous problem's record
            const double clipped = fmin(fmax(x0old, l), h);
            const double bound = up ? h : l;
            const bool usebound = zp & !eq & m14;   // (bitwise on purpose: no short-circuit control flow in the sweeps)
            const bool useold = m0 | (zp & !usebound & !m4);
            const double step = (s == zslot) ? (zblock_upper ? h : l) : zold + zalpha * (x0old - zold);
            const double freeval = m2 ? step : clipped;
            return usebound ? bound : (useold ? zold : freeval);
        };

        // ================= forward sweep: lazy z update + pinned block elimination of own knots 1..m-1 =================
        // Trip u works on own knot j = u - off - 1 (slot mmax - u + 1): (1) z of knot j+1 from the raw fields fetched one trip
        // earlier, (2) fetch of the raw fields (old x[0], old z, bounds) of knot j+2 and of the next duration, (3) elimination of
        // knot j (j >= 1).  The fetches of (2) are issued AFTER everything of the previous trip has been consumed (the waits the
        // compiler places for those would otherwise cover the fresh loads too) and have the whole of (3) to land.
        FullBlocks<R> sa;
        // The own boundary knot plays the part of an eliminated knot with S^-1 = 0 (nothing free) and h = its Hermite data: the first
        // interior knot then needs no special case (E = 0, the coupling moves to the right-hand side through h).
        Inv lprev;
        IP::zero(lprev);
        double hprev[R];
        double zp_ = 0.0, zc = 0.0, zn = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            hprev[i] = x0[i];
#pragma unroll
            for (int c = 0; c < R; ++c) { sa.B11[i][c] = (i == c) ? 1.0 : 0.0; sa.B01[i][c] = 0.0; }
        }
        {
            double rx = 0.0, rz = 0.0, rl = 0.0, rh = 0.0, rw = 0.0;
            double Tn = TT[tclamp(tbase + ksign * (mmax + 1))];
            for (int u = 0; u <= mmax; ++u) {
                const int j = u - off - 1;
                const int sj = mmax - u + 1;  // slot of own knot j; j + 1 <-> sj - 1, j + 2 <-> sj - 2
                // (1)
                const double Tcur = Tn;
                {
                    const double zv = znew(sj - 1, rx, rz, rl, rh, rw);
                    zn = (j >= 0) ? zv : zn;
                }
                __builtin_amdgcn_sched_barrier(0);
                // (2)
                Tn = TT[tclamp(tbase + ksign * (sj - 1))];   // own segment j + 1 = m - (sj - 1)
                if (sj >= 2) ld_xz(sj - 2, rx, rz);
                {
                    const int kk2 = kclamp(kslot(sj - 2)), k2 = 3 * kk2;
                    rl = LO[k2];
                    rh = HI[k2];
                    rw = ZS[(zprev ? max(min(kk2, M - 1), 0) : kk2) * zstride];   // (a lane without a problem has M = 0: stay in bounds)
                }
                __builtin_amdgcn_sched_barrier(0);
                // (3)
                FullBlocks<R> sb;
                sb.build(Tcur);
                if (u >= 2 && j >= 1) {   // (u >= 2: uniform; j >= 1: the halves that have started)
                    const bool pk = (s_pin >> sj) & 1ull;
                    const bool pprev = (j > 1) & (bool)((s_pin >> (sj + 1)) & 1ull);   // (j = 1: the boundary knot, all of it known through h)
                    const bool pnext = (s_pin >> (sj - 1)) & 1ull;  // own knot j+1 <= m: an interior knot (the meeting knot at the latest)
                    double D[R][R], rhs[R];
                    const double znm = pnext ? zn : 0.0, zcm = pk ? zc : 0.0, zpm = pprev ? zp_ : 0.0;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        rhs[i] = -sb.B01[i][0] * znm - sa.B01[0][i] * zpm;   // known positions of pinned neighbours
#pragma unroll
                        for (int c = 0; c < R; ++c) D[i][c] = sa.B11[i][c] + sb.B00(i, c);
                    }
#pragma unroll
                    for (int i = 1; i < R; ++i) {
                        rhs[i] -= D[i][0] * zcm;
                        D[i][0] = pk ? 0.0 : D[i][0];
                    }
                    D[0][0] = pk ? 1.0 : D[0][0];
                    // masked coupling block between (j-1, j) and E = S_{j-1}^-1 Mp from the previous inverse
                    // Schur update D -= Mp' S_{j-1}^-1 Mp through the factors: Y = L^-1 Mp, D -= Y' (D^-1 Y) -- forward substitutions only
                    double Mp[R][R], Yp[R][R], Zp[R][R];
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) Mp[i][c] = ((pprev & (i == 0)) | (pk & (c == 0))) ? 0.0 : sa.B01[i][c];
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
                        for (int q = 0; q < R; ++q) {
#pragma unroll
                            for (int c = 0; c <= i; ++c) D[i][c] -= Yp[q][i] * Zp[q][c];
                            rhs[i] -= Mp[q][i] * hprev[q];
                        }
                    rhs[0] = pk ? zc : rhs[0];   // pinned: row 0 of the system is the identity
                    Inv ldl;
                    ldl.factor(D);
                    ldl.solve(rhs);
                    st_rec(sj, ldl, rhs, zc);  // factors, h, z of knot j
#pragma unroll
                    for (int i = 0; i < R; ++i) hprev[i] = rhs[i];
                    lprev = ldl;
                }
                sa = sb;
                zp_ = zc;
                zc = zn;
            }
        }
        zmode = 0;
        UAVQP_CT(1);

        // ================= meeting knot (own knot m, slot 0): own partial Schur complement, exchange, solve =================
        // sa = blocks of the last own segment m-1, lprev / hprev = factors and h of own knot m-1, zp_ / zc = z of knots m-1 / m.
        double xn[R];  // solution at the meeting knot, own frame
        const bool pc = act && (s_pin & 1ull);
        {
            const bool pprev = (mm > 1) & (bool)((s_pin >> 1) & 1ull);
            double P[R][R], q[R];
            const double zpm = pprev ? zp_ : 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                q[i] = -sa.B01[0][i] * zpm;
#pragma unroll
                for (int c = 0; c < R; ++c) P[i][c] = sa.B11[i][c];
            }
            double Mp[R][R], Ep[R][R];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int c = 0; c < R; ++c) Mp[i][c] = ((pprev & (i == 0)) | (pc & (c == 0))) ? 0.0 : sa.B01[i][c];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                double col[R];
#pragma unroll
                for (int i = 0; i < R; ++i) col[i] = Mp[i][c];
                lprev.solve(col);
#pragma unroll
                for (int i = 0; i < R; ++i) Ep[i][c] = col[i];
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int qq = 0; qq < R; ++qq) {
#pragma unroll
                    for (int c = 0; c <= i; ++c) P[i][c] -= Mp[qq][i] * Ep[qq][c];
                    q[i] -= Mp[qq][i] * hprev[qq];
                }
            // S = P_own + F P_other F,  rhs = q_own + F q_other,  F = diag((-1)^d): the partner's frame is the reversed one
            double S[R][R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
#pragma unroll
                for (int c = 0; c <= i; ++c) {
                    const double o = swap_pair(P[i][c]);
                    S[i][c] = P[i][c] + (((i + c) & 1) ? -o : o);
                }
#pragma unroll
                for (int c = i + 1; c < R; ++c) S[i][c] = 0.0;  // upper triangle is never read
                const double o = swap_pair(q[i]);
                xn[i] = q[i] + ((i & 1) ? -o : o);
            }
            const double zcm = pc ? zc : 0.0;
#pragma unroll
            for (int i = 1; i < R; ++i) {
                xn[i] -= S[i][0] * zcm;
                S[i][0] = pc ? 0.0 : S[i][0];
            }
            S[0][0] = pc ? 1.0 : S[0][0];
            xn[0] = pc ? zc : xn[0];
            Inv ldl;
            ldl.factor(S);
            ldl.solve(xn);
            if (act) st_xz(0, xn, zc);
        }

        UAVQP_CT(2);
        // ================= backward sweep: x_j = h_j - S_j^-1 (M_j x_{j+1}) + the decisions of this iteration ======
        // Trip i computes x of own knot j = m - 1 - i (slot i + 1; j = 0: the boundary knot, j < 0: this half is finished) -- the
        // only loop-carried chain -- and, independent of it and ONE KNOT LATE, evaluates what the previous trips left complete:
        // the multiplier of knot j + 2 (slot i - 1), the first half of knot j + 1's, and the box test of the free position of
        // knot j + 1 (slot i).  A dependent FP64 result takes ~40 cycles: the two strands of a trip interleave, each hides the
        // other's latency.  The record of knot j-1 (inverse, h, z) and its bounds are fetched while knot j is processed.
        // Decisions are gathered in SLOT numbering and translated to original knot numbers afterwards.  The ratio test keeps the
        // step length as a fraction (an / ad, both >= 0) and compares by cross-multiplication: one division per iteration
        // instead of one per knot on the chain.
        constexpr int NONE = 1 << 30;
        unsigned long long np_s = 0ull, nu_s = 0ull;       // block-pivot round (slot order)
        double an = 1.0, ad = 1.0, worst = 0.0;             // ratio test (alpha = an / ad) / worst wrong-signed multiplier
        int block = NONE, rel = NONE;                       // ORIGINAL knot numbers; ties go to the lowest knot
        bool block_upper = false;
        {
            double nx[F], nl, nh, Tn;       // record of the knot processed next
#pragma unroll
            for (int f = 0; f < F; ++f) nx[f] = 0.0;
            if (mmax >= 2) ld_rec(1, nx);
            {
                const int k1 = 3 * kclamp(kslot(1));
                nl = LO[k1];
                nh = HI[k1];
                Tn = TT[tclamp(tbase + ksign)];   // own segment m - 1
            }
            // what the delayed strand works on: x of the last two knots, row-0 pieces of the segment between them, box and z of
            // the later one (starts with the meeting knot)
            double xnn[R], pe11[R], pe01r[R], pe01c[R];
#pragma unroll
            for (int q = 0; q < R; ++q) { xnn[q] = 0.0; pe11[q] = 0.0; pe01r[q] = 0.0; pe01c[q] = 0.0; }
            double plk, phk, pzk = zc;
            {
                const int kc = 3 * kclamp(kslot(0));
                plk = LO[kc];
                phk = HI[kc];
            }
            double lamA = 0.0, magA = 0.0;  // part of a knot's multiplier known one trip before the rest
            for (int i = 0; i <= mmax; ++i) {
                const int j = mm - 1 - i;
                const bool interior = j >= 1;
                double cur[F];
#pragma unroll
                for (int f = 0; f < F; ++f) cur[f] = nx[f];
                const double lk = nl, hk = nh, zk = cur[F_Z], Tcur = Tn;
                __builtin_amdgcn_sched_barrier(0);
                if (i + 2 < mmax) ld_rec(i + 2, nx);   // uniform: slot i + 2 = own knot j - 1
                {
                    const int k1 = 3 * kclamp(kslot(i + 2));
                    nl = LO[k1];
                    nh = HI[k1];
                    Tn = TT[tclamp(tbase + ksign * (i + 2))];   // own segment j - 1 = m - (i + 2)
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- strand 1: x_j (speculative for j <= 0 and in the extra last trip: discarded) ----------------
                // own segment j: inverse powers of its duration and the row-0 pieces of its blocks
                const double itv = fast_rcp(Tcur);
                double ip[2 * R];
                ip[0] = 1.0;
#pragma unroll
                for (int q = 1; q < 2 * R; ++q) ip[q] = ip[q - 1] * itv;
                double e11[R], e01r[R], e01c[R];  // B11_s[0][c], B01_s[0][c], B01_s[c][0]
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const double p = ip[2 * R - 1 - c];
                    e11[c] = p * Tab<R>::W(0, c);
                    e01r[c] = -p * Tab<R>::V(0, c);
                    e01c[c] = -p * Tab<R>::V(c, 0);
                }
                const bool pk = (s_pin >> (i + 1)) & 1ull;      // own knot j
                const bool pj = (s_pin >> i) & 1ull;            // own knot j + 1
                double x[R];
                {
                    double t[R];
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        double acc = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const double mv = -ip[2 * R - 1 - q - c] * Tab<R>::V(q, c);  // B01 of own segment j
                            acc += (((pk & (q == 0)) | (pj & (c == 0))) ? 0.0 : mv) * xn[c];
                        }
                        t[q] = acc;
                    }
                    Inv ldl;
                    {
                        double e[NE];
#pragma unroll
                        for (int q = 0; q < NE; ++q) e[q] = cur[F_I + q];
                        IP::set(ldl, e);
                    }
                    ldl.solve(t);
#pragma unroll
                    for (int q = 0; q < R; ++q) x[q] = interior ? cur[F_X + q] - t[q] : x0[q];
                }
                if (interior) st_x(i + 1, x);
                // ---------------- strand 2 (one knot late): xn = x of knot j + 1, xnn = x of knot j + 2, pe* = the segment between ----
                const bool d_on = (i >= 1) & (j >= -1), d_interior = (i >= 1) & (j >= 0);
                {
                    // multiplier of own knot j + 2: d(cost)/d p (up to the factor 2), row 0 of the unmasked block row; lower bound
                    // active: need lam >= 0, upper: lam <= 0.  For the meeting knot (i = 1) the other half of the row comes from
                    // the partner lane (a scalar: frame-independent).
                    double t1[R], t2[R];
#pragma unroll
                    for (int c = 0; c < R; ++c) { t1[c] = pe01c[c] * xn[c]; t2[c] = pe11[c] * xnn[c]; }
                    double own = t1[0] + t2[0], omag = fabs(t1[0]) + fabs(t2[0]);
#pragma unroll
                    for (int c = 1; c < R; ++c) { own += t1[c] + t2[c]; omag += fabs(t1[c]) + fabs(t2[c]); }
                    const double so = swap_pair(own), sm = swap_pair(omag);
                    lamA = (i == 1) ? so : lamA;
                    magA = (i == 1) ? sm : magA;
                    const int s2 = i >= 1 ? i - 1 : 0;   // slot of own knot j + 2
                    const int kj = kslot(s2);
                    const double lam = lamA + own, mag = magA + omag;
                    const bool p2 = (s_pin >> s2) & 1ull, ej = (s_eq >> s2) & 1ull, uj = (s_up >> s2) & 1ull;
                    const double viol = uj ? lam : -lam;
                    // rounding of lam is a few ulp of mag; 1e-11 let a 4e-4-relative wrong-signed multiplier pass on T^-7-scaled
                    // blocks (tools/soak.py, seed 11)
                    const bool wrong = viol > 1e-13 * mag;
                    const bool cand = d_on & p2 & !ej;
                    const bool keep = cand & !wrong;   // multiplier has the right sign: stays active in a block-pivot round
                    np_s |= (unsigned long long)keep << s2;
                    nu_s |= (unsigned long long)(keep & uj) << s2;
                    const bool take = cand & wrong & ((viol > worst) | ((viol == worst) & (kj < rel)));
                    worst = take ? viol : worst;
                    rel = take ? kj : rel;
                }
                {   // first half of own knot (j + 1)'s multiplier: its right-hand segment (needs x_{j+1} and x_{j+2} only)
                    double t2[R], t3[R];
#pragma unroll
                    for (int c = 0; c < R; ++c) { t2[c] = ((c & 1) ? -pe11[c] : pe11[c]) * xn[c]; t3[c] = pe01r[c] * xnn[c]; }
                    double la = t2[0] + t3[0], ma = fabs(t2[0]) + fabs(t3[0]);
#pragma unroll
                    for (int c = 1; c < R; ++c) { la += t2[c] + t3[c]; ma += fabs(t2[c]) + fabs(t3[c]); }
                    lamA = la;
                    magA = ma;
                }
                {   // free position of own knot j + 1 (slot i): outside its box?
                    const int kk = kslot(i);
                    const double ph = xn[0];
                    const bool pfree = !((s_pin >> i) & 1ull);
                    const bool below = ph < plk - 1e-12 * (1.0 + fabs(plk));
                    const bool above = !below & (ph > phk + 1e-12 * (1.0 + fabs(phk)));
                    const bool v = ((i == 0) ? (act & !pc) : d_interior) & pfree & (below | above);
                    np_s |= (unsigned long long)v << i;
                    nu_s |= (unsigned long long)(v & above) << i;
                    const double num = fabs((above ? phk : plk) - pzk), den = fabs(ph - pzk);   // step to the bound / full step
                    const double lhs = num * ad, rhs_ = an * den;                                 // num / den < an / ad ?
                    const bool take = v & ((lhs < rhs_) | ((lhs == rhs_) & (kk < block)));
                    an = take ? num : an;
                    ad = take ? den : ad;
                    block = take ? kk : block;
                    block_upper = take ? above : block_upper;
                }
                // ---------------- rotate ----------------
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    xnn[q] = xn[q];
                    xn[q] = x[q];
                    pe11[q] = e11[q];
                    pe01r[q] = e01r[q];
                    pe01c[q] = e01c[q];
                }
                plk = lk;
                phk = hk;
                pzk = zk;
            }
        }
        UAVQP_CT(3);
        // ---- combine the decisions of the two halves (identical values in both lanes afterwards)
        unsigned long long npin = eqmask | fromslot(np_s), nupper = fromslot(nu_s);
        double alpha = 1.0;
        {
            npin |= swap_pair_u64(npin);
            nupper |= swap_pair_u64(nupper);
            const double oan = swap_pair(an), oad = swap_pair(ad), ow = swap_pair(worst);
            const int ob = swap_pair_i(block), obu = swap_pair_i(block_upper ? 1 : 0), orl = swap_pair_i(rel);
            const double lhs = oan * ad, rhs_ = an * oad;
            if (lhs < rhs_ || (lhs == rhs_ && ob < block)) { an = oan; ad = oad; block = ob; block_upper = obu != 0; }
            if (ow > worst || (ow == worst && orl < rel)) { worst = ow; rel = orl; }
            alpha = an / ad;   // (an = ad = 1 when nothing blocks)
        }

        bool done = false;
        if (act) {
            if (final_pass) {
                done = true;
            } else if (pdas_left > 0) {
                // ---- block-pivoting warm-up (primal-dual active set): the first iterations change the whole working set at
                // once: every free position outside its box is pinned at the violated bound, every pinned one whose multiplier
                // has the wrong sign is released.  It usually identifies the active set in 2-3 solves (vs one change per solve)
                // but is not monotone, so after pdas_rounds rounds the safe single-pivot method takes over from the clipped
                // (feasible) point.
                ++it;
                if (npin == pin && nupper == upper) {  // KKT point: free positions feasible, all multipliers right
                    done = true;
                } else {
                    --pdas_left;
                    zmode = 1;
                    zpin = npin;
                    pin = npin;
                    upper = nupper;
                    if (it >= a.max_iter) { pin = ~0ull; final_pass = true; }
                }
            } else {
                // ---- safe phase: primal active set, one change per solve
                bool converged = false;
                if (block != NONE) {
                    // partial step to the first blocking bound, which joins the working set
                    zmode = 2;
                    zpin = pin;
                    zblock = block;
                    zblock_upper = block_upper;
                    zalpha = alpha < 0.0 ? 0.0 : alpha;
                    pin |= 1ull << block;
                    if (block_upper) upper |= 1ull << block; else upper &= ~(1ull << block);
                } else if (rel == NONE) {
                    converged = true;
                } else {
                    // full step: free positions move to the subspace minimiser; release the worst wrong-signed multiplier
                    zmode = 3;
                    zpin = pin;
                    pin &= ~(1ull << rel);
                }
                ++it;
                if (converged) done = true;
                else if (it >= a.max_iter) {
                    pin = ~0ull;  // freeze the feasible iterate, re-solve the derivatives only
                    final_pass = true;
                }
            }
        }

        UAVQP_CT(4);
#ifdef UAVQP_CORRIDOR_TIMING
        ct_acc[6] += 1;
#endif
        // ================= a finished pair hands over its Hermite solution and frees its slot =================
        if (__ballot(done) != 0ull) {
            if (!a.fused_emit) {
                for (int s = 0; s < mmax; ++s) {   // slot s = own knot m - s
                    const int j = mm - s;
                    if (done && j >= 1 && (s > 0 || !isR)) {  // the meeting knot is written by the L lane
                        const int kk = korig(j);
                        double xs[R], zs;
                        if (in_lds(s)) {
#pragma unroll
                            for (int q = 0; q < R; ++q) xs[q] = L(s, F_X + q);
                            zs = L(s, F_Z);
                        } else {
#pragma unroll
                            for (int q = 0; q < R; ++q) xs[q] = G(s, F_X + q);
                            zs = G(s, F_Z);
                        }
                        if (bit(pin, kk)) xs[0] = zs;  // pinned positions: exact bound value
                        double* o = a.xsol + (base3 + 3LL * kk) * R;
#pragma unroll
                        for (int q = 0; q < R; ++q) o[q] = (isR && (q & 1)) ? -xs[q] : xs[q];  // back to the original frame
                    }
                }
            } else {
                // The polynomials of the own segments, straight from the sweep state: own segment j joins own knots j and j + 1 -- walking the
                // slots from the meeting knot outwards, the previous slot is the segment's other end; own knot 0 is the boundary knot (x0).  The
                // same segment_coeffs on the same numbers corridor_emit_kernel would read back from xsol (Hermite data in the ORIGINAL frame:
                // the reversed lane flips the odd derivatives and swaps the ends), so the coefficients are bit-identical to the two-kernel
                // path; what is saved is the round trip of the Hermite solution through HBM and a launch (round 4: 55 us of config 3's 530).
                constexpr int NC = 2 * R;
                const bool al16 = (reinterpret_cast<uintptr_t>(a.coeff) & 15u) == 0;
                // Whole wave done at once, uniform batch, every slot on chip (the one-iteration case the dual prelude makes the rule): the
                // coefficients go through LDS -- over the sweep records nobody needs any more, [pair][segment][2 R] -- and leave as linear
                // 16-byte-per-lane stores, whole lines.  (Straight from the lanes, a store instruction is 64 scattered 16-byte pieces: measured
                // +45 us on config 3's solve kernel for 151 MB of output.)
                const bool whole = a.uniform > 0 && al16 && mmax <= NT && __ballot(act && !done) == 0ull;
                if (whole) {
                    const int Mu = a.uniform;
                    double X[NT + 1][R];
#pragma unroll
                    for (int s = 0; s <= NT; ++s) {
                        const int j = mm - s;
#pragma unroll
                        for (int q = 0; q < R; ++q) X[s][q] = x0[q];
                        if (s < NT && s < mmax) {
                            double xs[R];
#pragma unroll
                            for (int q = 0; q < R; ++q) xs[q] = L(s, F_X + q);
                            const double zs = L(s, F_Z);
                            if (bit(pin, korig(j))) xs[0] = zs;
                            if (j >= 1) {
#pragma unroll
                                for (int q = 0; q < R; ++q) X[s][q] = xs[q];
                            }
                        }
#pragma unroll
                        for (int q = 1; q < R; q += 2) X[s][q] = isR ? -X[s][q] : X[s][q];
                    }
                    wave_lds_sync();
                    bool finite = true;
                    double* const stage = s_rec + (size_t)(lane >> 1) * Mu * NC;
#pragma unroll
                    for (int s = 1; s <= NT; ++s) {
                        const int j = mm - s;
                        if (s <= mmax && done && j >= 0) {
                            const int seg = isR ? M - 1 - j : j;
                            double ys[ND], ye[ND], c[NC];
#pragma unroll
                            for (int d = 0; d < ND; ++d) { ys[d] = isR ? X[s - 1][d + 1] : X[s][d + 1]; ye[d] = isR ? X[s][d + 1] : X[s - 1][d + 1]; }
                            const double Tk = TT[seg];
                            segment_coeffs_det<R>(isR ? X[s - 1][0] : X[s][0], ys, isR ? X[s][0] : X[s - 1][0], ye, Tk, fast_rcp(Tk), c);
                            finite = finite && (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
#pragma unroll
                            for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(stage + seg * NC + q) = make_double2(c[q], c[q + 1]);
                        }
                    }
                    if (done && !finite) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                    wave_lds_sync();
                    const int PP = Mu * R;                                   // 16-byte pieces per problem
                    const unsigned inv = 0xFFFFFFFFu / (unsigned)PP + 1u;    // t / PP = (t * inv) >> 32 for t < 2^16
                    const int gi = done ? (int)g : -1;
                    for (int t0 = 0; t0 < 32 * PP; t0 += 64) {
                        const int t = t0 + lane;
                        const int pr = (int)(((unsigned long long)(unsigned)t * inv) >> 32);
                        const int gp = __shfl(gi, 2 * (pr < 32 ? pr : 31), 64);
                        if (pr < 32 && gp >= 0) {
                            const int off = t - pr * PP;
                            const double2 v = *reinterpret_cast<const double2*>(s_rec + 2 * (size_t)t);
                            *reinterpret_cast<double2*>(a.coeff + (size_t)gp * Mu * NC + 2 * off) = v;
                        }
                    }
                    wave_lds_sync();
                } else {
                double xn[R];
#pragma unroll
                for (int q = 0; q < R; ++q) xn[q] = 0.0;
                bool finite = true;
                for (int s = 0; s <= mmax; ++s) {
                    const int j = mm - s;
                    double xo[R];
#pragma unroll
                    for (int q = 0; q < R; ++q) xo[q] = x0[q];
                    if (s < mmax) {
                        const int kk = korig(j);
                        double xs[R], zs;
                        if (in_lds(s)) {
#pragma unroll
                            for (int q = 0; q < R; ++q) xs[q] = L(s, F_X + q);
                            zs = L(s, F_Z);
                        } else {
#pragma unroll
                            for (int q = 0; q < R; ++q) xs[q] = G(s, F_X + q);
                            zs = G(s, F_Z);
                        }
                        if (bit(pin, kk)) xs[0] = zs;  // pinned positions: exact bound value
                        if (j >= 1) {
#pragma unroll
                            for (int q = 0; q < R; ++q) xo[q] = xs[q];
                        }
                    }
#pragma unroll
                    for (int q = 1; q < R; q += 2) xo[q] = isR ? -xo[q] : xo[q];     // back to the original frame
                    if (done && j >= 0 && s >= 1) {
                        const int seg = isR ? M - 1 - j : j;                          // original segment of own segment j
                        double ys[ND], ye[ND], c[NC];
#pragma unroll
                        for (int d = 0; d < ND; ++d) { ys[d] = isR ? xn[d + 1] : xo[d + 1]; ye[d] = isR ? xo[d + 1] : xn[d + 1]; }
                        const double Tk = TT[seg];
                        segment_coeffs_det<R>(isR ? xn[0] : xo[0], ys, isR ? xo[0] : xn[0], ye, Tk, fast_rcp(Tk), c);
                        finite = finite && (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
                        double* o = a.coeff + ((size_t)3 * s0 + (size_t)axis * M + seg) * NC;
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
                }
                if (done && !finite) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
                }
            }
            if (done && !isR) {
                if (final_pass) atomicMin(&a.status[b], (int32_t)UAVQP_MAX_ITER_REACHED);
                if (a.iters) atomicMax(&a.iters[b], (int32_t)it);
                if (a.active) {
                    const unsigned long long valid = ((1ull << M) - 2ull);  // bits 1..M-1 (M >= 2 here)
                    const unsigned long long fin = final_pass ? 0ull : (pin & ~eqmask & valid);
                    a.active[2 * g] = fin;
                    a.active[2 * g + 1] = upper & fin;
                }
            }
            if (done) { g = -1; m = 0; }
        }
        UAVQP_CT(5);
    }
    UAVQP_CT_FLUSH;
}

// ---------------------------------------------------------------------------------------------------
// Hermite solution -> monomial coefficients (ascending powers, segment-local time: the reference's coef_1d_ layout,
// minimum_control.cpp:186).  One lane per (trajectory, axis, segment): lane e writes the e-th 2r-coefficient chunk of
// the output, so consecutive lanes own consecutive 48- / 64-byte chunks (written out through LDS, coalesced).
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void corridor_emit_kernel(CorridorArgs a, long long total_chunks) {
    constexpr int ND = R - 1, NC = 2 * R;
    // the 256 lanes of a block own 256 consecutive chunks = one contiguous piece of the output: the coefficients go through LDS
    // (row stride NC + 1 doubles) and leave as linear 16-byte-per-lane stores instead of NC / 2 stores at a 48- / 64-byte lane stride
    __shared__ __attribute__((aligned(16))) double s_c[256 * NC];
    __shared__ unsigned char s_keep[256];
    const int tid = threadIdx.x;
    const bool al16 = (reinterpret_cast<uintptr_t>(a.coeff) & 15u) == 0;
    for (long long e0 = (long long)blockIdx.x * 256; e0 < total_chunks; e0 += (long long)gridDim.x * 256) {
        const long long e = e0 + tid;
        bool keep = false;
        double c[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) c[q] = 0.0;
        if (e < total_chunks) {
            int b, s0, M;
            if (a.uniform > 0) {
                M = a.uniform;
                b = (int)(e / (3LL * M));
                s0 = b * M;
            } else {
                // chunk e belongs to the trajectory b with 3 seg_offsets[b] <= e < 3 seg_offsets[b+1]
                int lo_b = 0, hi_b = a.n_traj - 1;
                while (lo_b < hi_b) {
                    const int mid = (lo_b + hi_b + 1) >> 1;
                    if (3LL * a.seg_offsets[mid] <= e) lo_b = mid; else hi_b = mid - 1;
                }
                b = lo_b;
                s0 = a.seg_offsets[b];
                M = a.seg_offsets[b + 1] - s0;
            }
            const int st = a.status[b];
            if (!(st == UAVQP_INVALID_INPUT || M < 1) && corridor_takes_part(a.only_i32, a.only_u8, b)) {   // (those are left untouched)
                keep = true;
                const int rem = (int)(e - 3LL * s0), ax = rem / M, k = rem - ax * M;
                const size_t row0 = (size_t)(s0 + b);
                const double* bc = a.bc + (size_t)b * 2 * ND * 3 + ax;
                double p0, p1, ys[ND], ye[ND];
                if (k == 0) {
                    p0 = a.waypoints[3 * row0 + ax];
#pragma unroll
                    for (int d = 0; d < ND; ++d) ys[d] = bc[d * 3];
                } else {
                    const double* x = a.xsol + (3 * (row0 + k) + ax) * R;
                    p0 = x[0];
#pragma unroll
                    for (int d = 0; d < ND; ++d) ys[d] = x[d + 1];
                }
                if (k == M - 1) {
                    p1 = a.waypoints[3 * (row0 + M) + ax];
#pragma unroll
                    for (int d = 0; d < ND; ++d) ye[d] = bc[(ND + d) * 3];
                } else {
                    const double* x = a.xsol + (3 * (row0 + k + 1) + ax) * R;
                    p1 = x[0];
#pragma unroll
                    for (int d = 0; d < ND; ++d) ye[d] = x[d + 1];
                }
                const double Tk = a.times[s0 + k];
                segment_coeffs_det<R>(p0, ys, p1, ye, Tk, fast_rcp(Tk), c);
                if (!((fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY))) atomicMin(&a.status[b], (int32_t)UAVQP_NON_FINITE);
            }
        }
        double* o = a.coeff + (size_t)e0 * NC;
        if (al16) {
            s_keep[tid] = keep ? 1 : 0;
#pragma unroll
            for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(s_c + tid * NC + q) = make_double2(c[q], c[q + 1]);
            __syncthreads();
            const long long left = total_chunks - e0;
            const int n_ch = left < 256 ? (int)left : 256;
            for (int i = tid; i < n_ch * (NC / 2); i += 256) {
                const int ch = i / (NC / 2);
                if (s_keep[ch]) *reinterpret_cast<double2*>(o + 2 * i) = *reinterpret_cast<const double2*>(s_c + 2 * i);
            }
            __syncthreads();
        } else if (keep) {   // the caller passed a view that starts at an odd double
#pragma unroll
            for (int q = 0; q < NC; ++q) o[(size_t)tid * NC + q] = c[q];
        }
    }
}

}  // namespace uavqp
