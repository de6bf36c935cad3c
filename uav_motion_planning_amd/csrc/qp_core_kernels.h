// qp_core_kernels.h -- BatchArgs and the kernels that need no solver header of their own: the window sort of the ragged dealing, the one-lane
// generic solve (any segment count), batched evaluation / length (N1), the exhaustive SE(3) ellipsoid check (N4) and the time re-allocation.
// (Moved out of uavqp.hip in round 6 so that every kernel family can be compiled as its own translation unit: csrc/Makefile.)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/uavqp.h"
#include "qp_device.h"

namespace uavqp {

struct BatchArgs {
    int n_traj;
    int uniform;       // > 0: uniform segment count
    int max_segments;  // ragged upper bound
    const int32_t* seg_offsets;
    const double* waypoints;
    const double* times;
    const double* bc;
    double* coeff;
    int32_t* status;
    double* ws;     // forward-sweep workspace (generic kernel)
    const int32_t* perm;  // ragged dealing (generic kernel, LSORT): lane slot -> trajectory
    const int4* perm4;    // the same with the trajectory's first segment and segment count packed in: {b, s0, M, 0} (pair kernel: one
                          // load instead of three dependent ones at the top of the wave)
    int fused_sort;       // pair kernel, LSORT (round 6): no window_sort_kernel launch -- every wave sorts its window's 512 segment counts itself (qp_generic2.h)
    double* dummy;  // 1 KiB sink for the predicated-off stores of the specialised kernel
#ifdef UAVQP_PHASE_TIMING
    long long* stamps;  // debug: s_memtime stamps of wave 0 (tools/ubench only)
#endif
};

// ---------------------------------------------------------------------------------------------------
// Generic kernel: any segment count (ragged batches), r = 3 or 4.
// Forward block elimination keeps E_k = S_k^-1 A01(k) and h_k = S_k^-1 z_k per interior knot in a
// HBM workspace  ws[wave][k-1][f][lane]  (every access a coalesced 512-byte row of the wave),
// the backward sweep re-reads them and emits segment coefficients as it goes.
//
// NAX = 3: one lane per trajectory carries all three axes (the factorisation is shared).
// NAX = 1: one lane per (trajectory, axis), 21 trajectories per wave: the 3 lanes of a trajectory repeat the (cheap)
//          matrix elimination, each carries one right-hand side and emits one axis.  Three times the waves and a
//          third of the per-lane state -- for batches that do not fill the machine with one lane per trajectory.
//          E_k is stored once per trajectory (by the x lane, in its own slot) and read by all three lanes from that
//          slot: same wave, program order, so the hand-off needs no fence.
// ---------------------------------------------------------------------------------------------------
// LSORT (ragged batches): every window of 16 x IPW consecutive trajectories (IPW = trajectories per wave: 64 or 21) is
// dealt to 16 consecutive single-wave workgroups by descending segment count -- workgroup q of the group takes ranks
// [IPW q, IPW (q + 1)) -- so that the lanes of one wave run sweeps of nearly equal length while the window stays
// contiguous in memory (a GLOBAL sort by M was measured at 157 -> 244 us on config 4: it destroys the locality the
// strided per-lane accesses live on).  The dealing is a permutation `perm` of the batch, written ONCE per window by
// window_sort_kernel (one workgroup per window, LDS counting sort); the solve kernel only reads it.  The order inside a
// bin is whatever the LDS atomics of that one sort give: it decides WHICH lane solves a trajectory, never the result, and
// since every trajectory index is written to exactly one slot of perm, none can be solved twice or dropped.
template <int WIN>
__global__ __launch_bounds__(256) void window_sort_kernel(const int32_t* __restrict__ seg_offsets, int n_traj, int32_t* __restrict__ perm,
                                                          int4* __restrict__ perm4) {
    constexpr int KPT = (WIN + 255) / 256;  // keys per thread
    __shared__ int s_cnt[256];
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = blockIdx.x * WIN;
    int key[KPT], off0[KPT], cnt[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int w = j * 256 + tid, t = base + w;
        int Mt = -1;
        off0[j] = 0;
        cnt[j] = 0;
        if (w < WIN && t < n_traj) {
            off0[j] = seg_offsets[t];
            cnt[j] = seg_offsets[t + 1] - off0[j];
            Mt = cnt[j] < 0 ? 0 : (cnt[j] > 255 ? 255 : cnt[j]);
        }
        key[j] = Mt;
    }
    s_cnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j)
        if (key[j] >= 0) atomicAdd(&s_cnt[255 - key[j]], 1);  // bin 0 = longest
    __syncthreads();
    // exclusive prefix over the 256 bins: wave scan + the totals of the waves in front
    const int c = s_cnt[tid];
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int run = incl - c;
    for (int q = 0; q < wv; ++q) run += s_wave[q];
    s_cnt[tid] = run;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j)
        if (key[j] >= 0) {
            const int pos = atomicAdd(&s_cnt[255 - key[j]], 1);
            perm[base + pos] = base + j * 256 + tid;
            if (perm4) perm4[base + pos] = make_int4(base + j * 256 + tid, off0[j], cnt[j], 0);
        }
}

template <int R, bool LSORT, int NAX>
__global__ __launch_bounds__(64) void solve_generic_kernel(BatchArgs a) {
    constexpr int ND = R - 1, NC = 2 * R, F = ND * ND + NAX * ND;
    constexpr int LPI = 3 / NAX;              // lanes per trajectory
    constexpr int IPW = 64 / LPI;             // trajectories per wave (64 or 21)
    const int lane = threadIdx.x;
    const int ax0 = NAX == 1 ? lane % 3 : 0;
    const int item = lane / LPI;
    const bool lane_used = item < IPW;   // NAX = 1: lane 63 idles
    // workspace [wave][interior knot][field][lane]: a wave's record of one knot is F consecutive 512-byte rows
    const int kmax = a.max_segments > 1 ? a.max_segments - 1 : 1;
    // (no __restrict__: for the x lane ws and wsE are the same address)
    double* ws = a.ws + (size_t)blockIdx.x * kmax * F * 64 + lane;
    const double* wsE = ws - ax0;  // the x lane's slot holds E for the whole trajectory
    constexpr size_t wstride = 64;
    const int n_items = gridDim.x * IPW;  // LSORT: the host rounds the grid to a multiple of 16 (whole windows per round)
    const int n_round = (a.n_traj + n_items - 1) / n_items;
    for (int round = 0; round < n_round; ++round) {
        int b = lane_used ? round * n_items + blockIdx.x * IPW + item : a.n_traj;
        if constexpr (LSORT) {
            if (b < a.n_traj) b = a.perm[b];  // dealt by segment count inside the window (window_sort_kernel)
        }
        if (b >= a.n_traj) continue;
        int s0, M;
        if (a.uniform > 0) {
            M = a.uniform;
            s0 = b * M;
        } else {
            s0 = a.seg_offsets[b];
            M = a.seg_offsets[b + 1] - s0;
        }
        const double* __restrict__ wp = a.waypoints + 3 * (size_t)(s0 + b);
        const double* __restrict__ T = a.times + s0;
        const double* __restrict__ bc = a.bc + (size_t)b * 2 * ND * 3;
        double* __restrict__ out = a.coeff + (size_t)3 * NC * s0;

        bool ok = (M >= 1) && (a.uniform > 0 || M <= a.max_segments);
        if (ok)
            for (int i = 0; i < M; ++i) ok = ok && (T[i] > 0.0) && (T[i] < INFINITY);
        if (!ok) {
            if (a.status && ax0 == 0) a.status[b] = UAVQP_INVALID_INPUT;
            continue;
        }

        double y0[ND][NAX], yM[ND][NAX];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                y0[d][ax] = bc[d * 3 + ax0 + ax];
                yM[d][ax] = bc[(ND + d) * 3 + ax0 + ax];
            }

        // ---------------- forward elimination over interior knots k = 1..M-1 ----------------
        SegBlocks<R> sa;
        sa.build(T[0]);
        double pb[NAX], dpa[NAX];
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) {
            pb[ax] = wp[3 + ax0 + ax];
            dpa[ax] = pb[ax] - wp[ax0 + ax];
        }
        double Eprev[ND][ND], hprev[ND][NAX];
        // software prefetch: the loads of step k+1 are issued before the arithmetic of step k (one lane per
        // trajectory has nothing else to hide an HBM round trip per knot behind)
        double Tn = M > 1 ? T[1] : 1.0, pn[NAX];
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) pn[ax] = M > 1 ? wp[6 + ax0 + ax] : 0.0;
        for (int k = 1; k < M; ++k) {
            const double Tk_ = Tn;
            double pcur[NAX];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) pcur[ax] = pn[ax];
            if (k + 1 < M) {
                Tn = T[k + 1];
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) pn[ax] = wp[3 * (k + 2) + ax0 + ax];
            }
            SegBlocks<R> sb;
            sb.build(Tk_);
            double dpb[NAX];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                const double pc = pcur[ax];
                dpb[ax] = pc - pb[ax];
                pb[ax] = pc;
            }
            double S[ND][ND], z[ND][NAX];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int j = 0; j < ND; ++j) S[i][j] = sa.A11[i][j] + sb.A00(i, j);
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) z[i][ax] = sb.gv(i) * dpb[ax] - sa.gw[i] * dpa[ax];
            }
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int j = 0; j < ND; ++j)
#pragma unroll
                        for (int ax = 0; ax < NAX; ++ax) z[i][ax] -= sa.A01[j][i] * y0[j][ax];
            } else {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int j = 0; j < ND; ++j) {
#pragma unroll
                        for (int c = 0; c < ND; ++c) S[i][c] -= sa.A01[j][i] * Eprev[j][c];
#pragma unroll
                        for (int ax = 0; ax < NAX; ++ax) z[i][ax] -= sa.A01[j][i] * hprev[j][ax];
                    }
            }
            if (k == M - 1) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int j = 0; j < ND; ++j)
#pragma unroll
                        for (int ax = 0; ax < NAX; ++ax) z[i][ax] -= sb.A01[i][j] * yM[j][ax];
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
                for (int i = 0; i < ND; ++i) hprev[i][ax] = col[i];
            }
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                double col[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) col[i] = sb.A01[i][c];
                ldl.solve(col);
#pragma unroll
                for (int i = 0; i < ND; ++i) Eprev[i][c] = col[i];
            }
            double* w = ws + (size_t)(k - 1) * F * wstride;
            if (ax0 == 0) {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int c = 0; c < ND; ++c) w[(size_t)(i * ND + c) * wstride] = Eprev[i][c];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) w[(size_t)(ND * ND + i * NAX + ax) * wstride] = hprev[i][ax];
            sa = sb;
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) dpa[ax] = dpb[ax];
        }

        // ---------------- backward substitution + coefficient emission ----------------
        double ynext[ND][NAX], pend[NAX];
        bool finite = true;
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) {
            pend[ax] = wp[3 * M + ax0 + ax];
#pragma unroll
            for (int d = 0; d < ND; ++d) ynext[d][ax] = yM[d][ax];
        }
        // software prefetch of the sweep state of knot k-1 (and of T, waypoint) while knot k is processed
        double wn[F], Tkn = T[M - 1], pkn[NAX];
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) pkn[ax] = wp[3 * (M - 1) + ax0 + ax];
        auto load_rec = [&](int knot, double (&dst)[F]) {  // interior knot `knot` = 1..M-1
            const size_t off = (size_t)(knot - 1) * F * wstride;
#pragma unroll
            for (int f = 0; f < ND * ND; ++f) dst[f] = wsE[off + (size_t)f * wstride];
#pragma unroll
            for (int f = ND * ND; f < F; ++f) dst[f] = ws[off + (size_t)f * wstride];
        };
        if (M >= 2) load_rec(M - 1, wn);
        for (int k = M - 1; k >= 0; --k) {
            double wc[F];
#pragma unroll
            for (int f = 0; f < F; ++f) wc[f] = wn[f];
            const double Tk = Tkn;
            double pkc[NAX];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) pkc[ax] = pkn[ax];
            if (k >= 1) {
                Tkn = T[k - 1];
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) pkn[ax] = wp[3 * (k - 1) + ax0 + ax];
                if (k >= 2) load_rec(k - 1, wn);
            }
            double y[ND][NAX];
            if (k == 0) {
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) y[d][ax] = y0[d][ax];
            } else {
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int ax = 0; ax < NAX; ++ax) y[i][ax] = wc[ND * ND + i * NAX + ax];
                if (k < M - 1) {
#pragma unroll
                    for (int i = 0; i < ND; ++i)
#pragma unroll
                        for (int c = 0; c < ND; ++c) {
                            const double e = wc[i * ND + c];
#pragma unroll
                            for (int ax = 0; ax < NAX; ++ax) y[i][ax] -= e * ynext[c][ax];
                        }
                }
            }
            const double itk = 1.0 / Tk;
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                const double pk = pkc[ax];
                double ys[ND], ye[ND], c[NC];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    ys[d] = y[d][ax];
                    ye[d] = ynext[d][ax];
                }
                segment_coeffs<R>(pk, ys, pend[ax], ye, Tk, itk, c);
                double* o = out + ((size_t)(ax0 + ax) * M + k) * NC;
#pragma unroll
                for (int j = 0; j < NC; ++j) o[j] = c[j];
                finite = finite && (fabs(c[NC - 1]) < INFINITY) && (fabs(c[R]) < INFINITY);
                pend[ax] = pk;
            }
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int ax = 0; ax < NAX; ++ax) ynext[d][ax] = y[d][ax];
        }
        if constexpr (NAX == 1) {  // AND over the three lanes of the trajectory (they took the same branches)
            const int f0 = finite ? 1 : 0, l0 = lane - ax0;
            finite = (__shfl(f0, l0, 64) & __shfl(f0, l0 + 1, 64) & __shfl(f0, l0 + 2, 64)) != 0;
        }
        if (a.status && ax0 == 0) a.status[b] = finite ? UAVQP_SOLVED : UAVQP_NON_FINITE;
    }
}

// ---------------------------------------------------------------------------------------------------
// N1: batched evaluation on a uniform time grid.  One lane per (trajectory, sample); consecutive lanes are
// consecutive samples of one trajectory, so coefficient reads hit the same few cache lines and the
// output (the dominant traffic: 24 B x K per sample) is fully coalesced.
// ---------------------------------------------------------------------------------------------------
struct EvalArgs {
    int n_traj, uniform, n_samples, what;
    const int32_t* seg_offsets;
    const double* times;
    const double* coeff;
    double t0, dt;
    double* out;
};

template <int R>
__global__ __launch_bounds__(256) void eval_kernel(EvalArgs a) {
    constexpr int NC = 2 * R;
    // the 256 lanes of a block own 256 consecutive (trajectory, sample) rows = one contiguous piece of the output: the rows go
    // through LDS (row stride 9 doubles: conflict-free) and leave as 16-byte-per-lane linear stores instead of nine 8-byte stores
    // at a 72-byte lane stride
    __shared__ __attribute__((aligned(16))) double s_o[256 * 9];
    const long long total = (long long)a.n_traj * a.n_samples;
    const int K = __popc(a.what & 7);
    const int tid = threadIdx.x;
    for (long long g0 = (long long)blockIdx.x * 256; g0 < total; g0 += (long long)gridDim.x * 256) {
        const long long g = g0 + tid;
        double res[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) res[k] = 0.0;
        if (g < total) {
            const int b = (int)(g / a.n_samples), s = (int)(g - (long long)b * a.n_samples);
            int s0, M;
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
            if (M >= 1) {
                const double* __restrict__ T = a.times + s0;
                // segment search exactly as PolyTraj::evaluatePos (poly_traj.hpp:77-88); written without an early exit: the same
                // subtractions in the same order, but the loads of T[i] do not depend on the comparisons
                double t = a.t0 + s * a.dt;
                int idx = 0;
                bool going = true;
                double Tlast = 0.0;
                for (int i = 0; i < M; ++i) {
                    const double Ti = T[i];
                    const bool adv = going & (t > Ti + 1e-4);
                    t = adv ? t - Ti : t;
                    idx += adv ? 1 : 0;
                    going = adv;
                    Tlast = Ti;
                }
                if (idx == M) {
                    --idx;
                    t = Tlast;
                }
                const double* __restrict__ c = a.coeff + (size_t)3 * NC * s0 + (size_t)idx * NC;
                int k = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (!((a.what >> d) & 1)) continue;
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        const double* ca = c + (size_t)ax * NC * M;
                        double acc = 0.0;
#pragma unroll
                        for (int j = NC - 1; j >= d; --j) {  // Horner on the d-th derivative
                            double f = 1.0;
                            for (int q = 0; q < d; ++q) f *= (double)(j - q);
                            acc = fma(acc, t, f * ca[j]);
                        }
                        res[k * 3 + ax] = acc;
                    }
                    ++k;
                }
            }
        }
        const int row = 3 * K;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (k < row) s_o[tid * row + k] = res[k];
        __syncthreads();
        const long long left = total - g0;
        const int n_rows = left < 256 ? (int)left : 256;
        const int n_d = n_rows * row;                          // doubles of this block's piece (even: 256 rows, or handled below)
        double* o = a.out + (size_t)g0 * row;
        const bool al16 = ((reinterpret_cast<uintptr_t>(o)) & 15u) == 0;
        if (al16) {
            for (int i = tid; 2 * i + 1 < n_d; i += 256) *reinterpret_cast<double2*>(o + 2 * i) = *reinterpret_cast<const double2*>(s_o + 2 * i);
            if ((n_d & 1) && tid == 0) o[n_d - 1] = s_o[n_d - 1];
        } else {
            for (int i = tid; i < n_d; i += 256) o[i] = s_o[i];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// N1 (continued): PolyTraj::getTraj + getLength + getMeanVel (traj_utils/poly_traj.hpp:175-207) for a whole batch.
// One wave per trajectory.  Lane 0 repeats the reference's sampling loop to COUNT the samples -- the reference accumulates
// t += dt in floating point and stops at t >= total_time, so for a total time that is a multiple of dt (its own constant
// 1.0 s per segment) the rounding of that accumulation decides whether the last sample exists; the count has to be exact.
// The 64 lanes then evaluate the chords in parallel at t_s = s dt (differs from the accumulated t by ~1e-16 s relative:
// rounding-level differences in the positions) and the wave sums them.
// ---------------------------------------------------------------------------------------------------
struct LengthArgs {
    int n_traj, uniform;
    const int32_t* seg_offsets;
    const double* times;
    const double* coeff;
    double dt;
    double* length;
    double* mean_vel;
    int32_t* n_samples;
};

template <int R>
__global__ __launch_bounds__(64) void traj_length_kernel(LengthArgs a) {
    constexpr int NC = 2 * R;
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < a.n_traj; b += gridDim.x) {
        int s0, M;
        if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        const double* __restrict__ T = a.times + s0;
        const double* __restrict__ c = a.coeff + (size_t)3 * NC * s0;
        double total = 0.0;
        int n = 0;
        if (lane == 0) {
            for (int i = 0; i < M; ++i) total += T[i];                      // PolyTraj::init :64-72
            double t = 0.0;
            while (t < total && n < (1 << 24)) { t += a.dt; ++n; }          // getTraj :180-184 (accumulated t)
        }
        total = __shfl(total, 0, 64);
        n = __shfl(n, 0, 64);
        auto pos = [&](int s, double (&p)[3]) {
            double t = (double)s * a.dt;
            int idx = 0;
            while (idx < M && t > T[idx] + 1e-4) { t -= T[idx]; ++idx; }    // evaluatePos :77-88
            if (idx == M) { --idx; t = T[idx]; }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double* ca = c + ((size_t)ax * M + idx) * NC;
                double v = 0.0;
#pragma unroll
                for (int j = NC - 1; j >= 0; --j) v = fma(v, t, ca[j]);
                p[ax] = v;
            }
        };
        double acc = 0.0;
        if (M >= 1)
            for (int s = lane; s + 1 < n; s += 64) {                        // getLength :189-202: chords between consecutive samples
                double p0[3], p1[3];
                pos(s, p0);
                pos(s + 1, p1);
                const double dx = p1[0] - p0[0], dy = p1[1] - p0[1], dz = p1[2] - p0[2];
                acc += sqrt(dx * dx + dy * dy + dz * dz);
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) {
            if (a.length) a.length[b] = acc;
            if (a.mean_vel) a.mean_vel[b] = acc / total;                    // getMeanVel :204-207
            if (a.n_samples) a.n_samples[b] = n;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// N4: SE(3) ellipsoid collision check.  One lane per (trajectory, sample); obstacle points stream through LDS
// in tiles shared by the 256 samples of a block.
// ---------------------------------------------------------------------------------------------------
struct EllipsoidArgs {
    int n_traj, uniform, n_samples, n_obs;
    const int32_t* seg_offsets;
    const double* times;
    const double* coeff;
    const double* obs;
    double t0, dt, robot_r, robot_h;
    int32_t* first_hit;
    uint8_t* flags;
};

template <int R>
__global__ __launch_bounds__(256) void ellipsoid_kernel(EllipsoidArgs a) {
    constexpr int NC = 2 * R, TILE = 1024;
    __shared__ double s_obs[TILE * 3];
    const long long total = (long long)a.n_traj * a.n_samples;
    const long long n_round = (total + 255) / 256 * 256;  // every thread of a block joins the LDS tile loads
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n_round; g += (long long)gridDim.x * 256) {
        bool live = g < total;
        int b = 0, s = 0;
        double p[3] = {0, 0, 0}, b1[3] = {1, 0, 0}, b2[3] = {0, 1, 0}, b3[3] = {0, 0, 1};
        bool empty = false;  // zero-segment trajectory (flagged invalid by the solver): nothing to sample, reported collision-free
        if (live) {
            b = (int)(g / a.n_samples);
            s = (int)(g - (long long)b * a.n_samples);
            int s0, M;
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
            empty = M < 1;
        }
        if (live && empty) {
            if (a.flags) a.flags[g] = 0;
            live = false;  // still joins the LDS tile loads below
        }
        if (live) {
            int s0, M;
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
            const double* __restrict__ T = a.times + s0;
            double t = a.t0 + s * a.dt;
            int idx = 0;
            while (idx < M && t > T[idx] + 1e-4) { t -= T[idx]; ++idx; }
            if (idx == M) { --idx; t = T[idx]; }
            double acc[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double* ca = a.coeff + (size_t)3 * NC * s0 + ((size_t)ax * M + idx) * NC;
                double pv = 0.0, av = 0.0;
#pragma unroll
                for (int j = NC - 1; j >= 0; --j) pv = fma(pv, t, ca[j]);
#pragma unroll
                for (int j = NC - 1; j >= 2; --j) av = fma(av, t, (double)(j * (j - 1)) * ca[j]);
                p[ax] = pv;
                acc[ax] = av;
            }
            // kino_astar.cpp:724-727
            double n3 = sqrt(acc[0] * acc[0] + acc[1] * acc[1] + (acc[2] + 9.81) * (acc[2] + 9.81));
            b3[0] = acc[0] / n3; b3[1] = acc[1] / n3; b3[2] = (acc[2] + 9.81) / n3;
            double c2[3] = {0.0, b3[2], -b3[1]};  // b3 x (1,0,0)
            double n2 = sqrt(c2[1] * c2[1] + c2[2] * c2[2]);
            b2[0] = 0.0; b2[1] = c2[1] / n2; b2[2] = c2[2] / n2;
            double c1[3] = {b2[1] * b3[2] - b2[2] * b3[1], b2[2] * b3[0] - b2[0] * b3[2], b2[0] * b3[1] - b2[1] * b3[0]};
            double n1 = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
            b1[0] = c1[0] / n1; b1[1] = c1[1] / n1; b1[2] = c1[2] / n1;
        }
        const double rad2 = (a.robot_r + 1e-1) * (a.robot_r + 1e-1);
        const double ir = 1.0 / a.robot_r, ih = 1.0 / a.robot_h;
        bool hit = false;
        for (int o0 = 0; o0 < a.n_obs; o0 += TILE) {
            const int nt = min(TILE, a.n_obs - o0);
            __syncthreads();
            for (int i = threadIdx.x; i < nt * 3; i += 256) s_obs[i] = a.obs[(size_t)o0 * 3 + i];
            __syncthreads();
            if (live && !hit) {
                for (int i = 0; i < nt; ++i) {
                    const double dx = s_obs[3 * i] - p[0], dy = s_obs[3 * i + 1] - p[1], dz = s_obs[3 * i + 2] - p[2];
                    if (dx * dx + dy * dy + dz * dz <= rad2) {  // the reference's radius search (r + 0.1)
                        const double e1 = (b1[0] * dx + b1[1] * dy + b1[2] * dz) * ir;
                        const double e2 = (b2[0] * dx + b2[1] * dy + b2[2] * dz) * ir;
                        const double e3 = (b3[0] * dx + b3[1] * dy + b3[2] * dz) * ih;
                        if (e1 * e1 + e2 * e2 + e3 * e3 <= 1.0) { hit = true; break; }  // |E^-1 d| <= 1
                    }
                }
            }
        }
        if (live) {
            if (a.flags) a.flags[g] = hit ? 1 : 0;
            if (hit) atomicMin(&a.first_hit[b], s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Time re-allocation: one lane per segment; peak |v|, |a| by sampling, stretch-only update of T.
// ---------------------------------------------------------------------------------------------------
struct ReallocArgs {
    int n_traj, uniform, samples;
    const int32_t* seg_offsets;
    double* times;
    const double* coeff;
    double v_max, a_max, max_stretch;
    double dead_band, overshoot;  // uavqp_settings.realloc_dead_band / realloc_overshoot
    int32_t* changed;
    double* scale_acc;            // optional [n_traj]: multiplied by the factor applied (the pipeline's record of how far a trajectory was stretched)
    const int32_t* list;          // optional: only these trajectories, *n_list of them (the pipeline's later rounds: a trajectory the last round
    const int* n_list;            // did not stretch was not re-solved -- its peaks, and so its verdict, are what they were)
};

template <int R>
__global__ __launch_bounds__(64) void realloc_kernel(ReallocArgs a) {
    // The whole trajectory is scaled by ONE factor.  (Stretching single segments diverges: a long segment next to short
    // ones inherits their knot acceleration and overshoots more the longer it gets; under uniform scaling T -> sT speeds
    // drop ~1/s and accelerations ~1/s^2.)  Eight lanes per trajectory: sub-lane j samples segments j, j + 8, ..., the
    // peaks are combined with three xor-shuffles (max is order-independent: same result as a single lane), every lane
    // then scales its own segments.
    constexpr int NC = 2 * R, LPT = 8;
    const int sub = threadIdx.x % LPT;
    const long long n_lanes = (long long)(a.list ? *a.n_list : a.n_traj) * LPT;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_round = (n_lanes + stride - 1) / stride * stride;  // whole waves take part in the shuffles
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n_round; g += stride) {
        const bool live = g < n_lanes;
        const int b = live ? (a.list ? a.list[g / LPT] : (int)(g / LPT)) : 0;
        int s0 = 0, M = 0;
        if (live) {
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        }
        double v2 = 0.0, a2 = 0.0;
        for (int i = sub; i < M; i += LPT) {
            const double T = a.times[s0 + i];
            const double* __restrict__ c = a.coeff + (size_t)3 * NC * s0 + (size_t)i * NC;
            for (int s = 0; s <= a.samples; ++s) {
                const double t = T * (double)s / (double)a.samples;
                double vs = 0.0, as = 0.0;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const double* ca = c + (size_t)ax * NC * M;
                    double v = 0.0, ac = 0.0;
#pragma unroll
                    for (int j = NC - 1; j >= 1; --j) v = fma(v, t, (double)j * ca[j]);
#pragma unroll
                    for (int j = NC - 1; j >= 2; --j) ac = fma(ac, t, (double)(j * (j - 1)) * ca[j]);
                    vs += v * v;
                    as += ac * ac;
                }
                v2 = fmax(v2, vs);
                a2 = fmax(a2, as);
            }
        }
#pragma unroll
        for (int d = 1; d < LPT; d <<= 1) {
            v2 = fmax(v2, __shfl_xor(v2, d, 64));
            a2 = fmax(a2, __shfl_xor(a2, d, 64));
        }
        if (!live) continue;
        const double ratio = fmax(sqrt(v2) / a.v_max, sqrt(sqrt(a2) / a.a_max));
        int ch = 0;
        // dead band (default 1 %) and overshoot (default 2 %) so that the loop settles instead of creeping towards the limit
        if (ratio > a.dead_band && ratio < INFINITY) {
            const double s = fmin(a.overshoot * ratio, a.max_stretch);
            for (int i = sub; i < M; i += LPT) a.times[s0 + i] *= s;
            ch = M;
            if (a.scale_acc && sub == 0) a.scale_acc[b] *= s;
        }
        if (a.changed && sub == 0) a.changed[b] = ch;
    }
}

}  // namespace uavqp
