// qp_generic2.h -- runtime-M equality solve with TWO lanes per trajectory (ragged batches, uninstantiated segment counts).
//
// The one-lane-per-trajectory kernel (solve_generic_kernel) runs 2 (M - 1) dependent block steps per lane and fills only
// half of the SIMDs at BASELINE config 4 (32768 trajectories = 512 waves on 1024 SIMDs).  Here the two-sided ("twisted") block
// elimination of qp_twisted.h is applied with a RUNTIME segment count: lane L eliminates the knots 1..c-1 forward, lane R the
// time-reversed trajectory (derivative d picks up (-1)^d) from knot M-1 down to c+1 with the same instruction stream, the partial
// Schur complements of the meeting knot c = ceil(M/2) are exchanged through DPP (lane ^ 1), both lanes solve it and
// back-substitute / emit their own half.  Half the sequential depth per lane, twice the waves, the three axes still share the
// factorisation (3 right-hand sides per lane).
//
// Memory: a lane that walks its own trajectory with 8-byte loads pays one HBM/L2 round trip per knot (and the compiler's
// s_waitcnt vmcnt(0) at the loop back-edge adds the round trip of the stores in flight): measured 3.6 us per own segment with
// ALL stores compiled out.  So the wave first copies the waypoints and durations of its 32 trajectories into LDS -- 8 lanes per
// trajectory, 16 bytes per lane, every load in flight before the first LDS write (LDS-DMA was tried first: one copy instruction
// per trajectory and run costs ~270 cycles of issue each) -- and the sweeps read LDS.  Durations are validated where the forward
// sweep reads them.  The forward-sweep state (E_j, h_j of the own knots) goes through the HBM workspace
// [wave][own knot][field pair][lane]; the forward loop has no vector-memory load, so its stores are never waited on; the
// backward loop alternates between two register sets for the records, re-loads a set two trips ahead and retires the re-loads
// in the basic block that follows the stores they must not wait for (vmcnt(N) past the stores, not vmcnt(0)).  Coefficients
// leave through LDS as 64-byte (r = 3: 48-byte) chunks, 4 (3) lanes each, spelled as GLOBAL stores.
// Ragged batches are dealt to the lane PAIRS by segment count inside windows of 16 waves x 32 trajectories; the rank ranges are rotated by
// the window index so that the long waves spread over the XCDs.  Round 6: the waves rank their window themselves (BatchArgs::fused_sort, see
// the top of the round loop) -- window_sort_kernel in front (rounds 2-5, uavqp_settings.ragged_window_sort = 2) cost a launch boundary and
// 8 + 16 bytes of permutation per trajectory for a sort of 512 small integers; the ranking costs the kernel ~1.8 us, the step is 0.5-2 us
// shorter inside a hipGraph and with eager launches (tools/g2_fused_ab.sh).  DESIGN.md section 5.2.
#pragma once
#include "qp_core_kernels.h"
#include "qp_wave_utils.h"

// (probe builds, tools/generic2_probe.py: -DG2_NO_WS / -DG2_NO_OUT take the workspace round trip / the coefficient stores out)
#ifdef G2_NO_WS
#define G2_WS_ST(dst, v) do { if ((v).x == 1.2345e300) (dst) = (v); } while (0)
#else
#define G2_WS_ST(dst, v) (dst) = (v)
#endif
#ifdef G2_NO_OUT
#define G2_OUT_ST2(addr, v) do { if ((v).x == 1.2345e300) store16_global((addr), (v)); } while (0)
#else
#define G2_OUT_ST2(addr, v) store16_global((addr), (v))
#endif
#ifdef G2_TIMING   // probe build: s_memtime stamps of wave 0 -> the sink page behind the first KiB (tools/generic2_sections.py)
#define G2_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(a.dummy)[128 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define G2_STAMP(k) do {} while (0)
#endif
#ifdef G2_TIMING
#define G2_ACC_DECL long long g2_acc[6] = {0, 0, 0, 0, 0, 0}; long long g2_t = 0;
#define G2_ACC_START g2_t = __builtin_readcyclecounter()
#define G2_ACC(k) do { const long long n_ = __builtin_readcyclecounter(); g2_acc[k] += n_ - g2_t; g2_t = n_; } while (0)
#define G2_ACC_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 6; ++k_) reinterpret_cast<long long*>(a.dummy)[128 + 9 + k_] = g2_acc[k_]; } while (0)
#else
#define G2_ACC_DECL
#define G2_ACC_START do {} while (0)
#define G2_ACC(k) do {} while (0)
#define G2_ACC_FLUSH do {} while (0)
#endif
#ifndef G2_INV
#define G2_INV SmallLDL
#endif

namespace uavqp {

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside the DPP rows (row_shr 1 / 2 / 4 / 8, zero fill), then the rows' totals across
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_incl_scan_add(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// 16-byte store to an address that went through LDS as an integer: spelled as a GLOBAL store (address space 1) -- from a generic
// pointer the compiler emits flat_store, and a pending FLAT operation makes it wait with vmcnt(0) / lgkmcnt(0) everywhere
__device__ __forceinline__ void store16_global(unsigned long long addr, double2 v) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) v2d* gptr;
    v2d w = {v.x, v.y};
    *reinterpret_cast<gptr>(addr) = w;
}

// LDS doubles per trajectory slot: 3 (Mx + 1) waypoint coordinates, Mx durations, each region rounded up to 16 bytes (the copies
// move 16 bytes per lane and may write 8 bytes past an odd-sized run)
__host__ __device__ inline int generic2_wp_doubles(int max_segments) { return (3 * (max_segments + 1) + 1) & ~1; }
__host__ __device__ inline int generic2_slot_doubles(int max_segments) { return (generic2_wp_doubles(max_segments) + ((max_segments + 1) & ~1)) | 2; }   // = 2 mod 4: slots 2-way over the banks
constexpr int G2_ROW = 26;   // staging row of a lane: 24 coefficients (r = 4) + 2 pad doubles: ds_write_b128 / ds_read_b128 conflict-free
__host__ __device__ inline size_t generic2_lds_bytes(int max_segments) {
    return sizeof(double) * (32 * (size_t)generic2_slot_doubles(max_segments) + 64 * G2_ROW + 64 * 2);
}

template <int R, bool LSORT>
__global__ __launch_bounds__(64) void solve_generic2_kernel(BatchArgs a) {
    constexpr int ND = R - 1, NC = 2 * R, F = ND * ND + 3 * ND;
    constexpr int IPW = 32;  // trajectories per wave
    extern __shared__ __attribute__((aligned(16))) double s_in[];
    const int lane = threadIdx.x;
    const int isR = lane & 1;
    const int item = lane >> 1;
    const int Mx = a.max_segments;
    const int stride = generic2_slot_doubles(Mx), TOFF = generic2_wp_doubles(Mx);
    const double* sl = s_in + item * stride;
    double* s_out = s_in + 32 * stride;                       // [lane][G2_ROW]: the coefficients a lane emitted in this trip
    double* s_meta = s_out + 64 * G2_ROW;                     // [lane]{address of its axis-0 chunk, axis stride in bytes}
    // workspace: [wave][own knot 1..][field][lane]
    const int kown = (Mx + 1) / 2;   // own segments of the longer half; it has kown - 1 eliminated knots
    // (two fields per lane and instruction: 16-byte stores / loads, 1 KiB per wave instruction -- the sweeps pay per vector-memory
    // instruction, ~100 cycles each with four waves per CU in the same phase, not per byte)
    static_assert(F % 2 == 0, "records are moved as double2");
    double2* ws = reinterpret_cast<double2*>(a.ws + (size_t)blockIdx.x * (size_t)(kown > 1 ? kown - 1 : 1) * F * 64) + lane;
    auto W2 = [&](int j, int f2) -> double2& { return ws[((size_t)(j - 1) * (F / 2) + f2) * 64]; };   // own knot j = 1..m-1
    auto load_rec = [&](int j, double (&rec)[F]) {
#pragma unroll
        for (int f2 = 0; f2 < F / 2; ++f2) {
#ifdef G2_NO_WS
            rec[2 * f2] = (double)(j + f2); rec[2 * f2 + 1] = 1.0;
#else
            const double2 t = W2(j, f2);
            rec[2 * f2] = t.x; rec[2 * f2 + 1] = t.y;
#endif
        }
    };

    const int n_items = gridDim.x * IPW;
    const int n_round = (a.n_traj + n_items - 1) / n_items;
    for (int round = 0; round < n_round; ++round) {
        int b = round * n_items + blockIdx.x * IPW + item;
        if constexpr (LSORT) {
            // Which 32 ranks of its window a wave takes is rotated by the window index: workgroup ids go round-robin over the 8
            // XCDs, so with the plain order every window's longest wave (ranks 0..31) would sit at an id = 0 mod 16 -- all the long
            // waves of the batch on XCD 0, next to each other on its CUs.  (The host rounds the grid to a multiple of 16.)
            // (any spreading does: xor with the window index, a rotation by 3 windows, ... all measure 49.7-50.0 us on config 4
            // against 53.7 us for the plain order)
            const int win = blockIdx.x >> 4, q = ((blockIdx.x & 15) + win) & 15;
            b = round * n_items + (win * 16 + q) * IPW + item;
        }
        G2_STAMP(0);
        int s0 = 0, M = 0;
        bool packed = false;
        if constexpr (LSORT) {
            if (a.fused_sort) {
                // ---- the dealing, by the wave itself (round 6; was window_sort_kernel behind its own launch boundary: 4.8 us of a 40 us step).
                // The 16 waves of a window each rank the window's 512 trajectories by (segment count descending, index ascending) -- the SAME
                // deterministic order in every one of them -- and take their own 32 ranks.  Trajectory i = 8 lane + j:
                //   * H[k] = trajectories of the window with count k (LDS adds: sums do not depend on their order), start[k] = those with a larger
                //     count (the window's total minus the inclusive prefix sum over the lanes, lane = count);
                //   * only the counts whose rank range [start, start + H) meets this wave's [32 q, 32 q + 32) matter to it -- one to three of them
                //     on a batch like config 4: for each, an exclusive scan over the lanes of "how many of my eight have it" (DPP) orders its
                //     trajectories by index, rank = start + lanes in front + earlier ones of the own eight.
                // (First version: a [64][64] table of 16-bit counts, every row scanned by its lane -- the whole order, of which a wave needs a
                // sixteenth: 3 us per wave.)  Counts are clamped to 0..63 for the ranking (the host uses this path up to 63 segments); the record
                // keeps the real count.
                const int win = blockIdx.x >> 4, q = ((blockIdx.x & 15) + win) & 15;
                const int wbase = round * n_items + win * 512;
                int* const H = reinterpret_cast<int*>(s_in);                                // [64], on top of the input slots (free until the staging)
                int4* const SLOT = reinterpret_cast<int4*>(s_in + 32);                      // [32] records of this wave's ranks
                int off[9], key[8];
                const int t0 = wbase + lane * 8;
#pragma unroll
                for (int j = 0; j < 9; ++j) off[j] = t0 + j <= a.n_traj ? a.seg_offsets[t0 + j] : 0;
                wave_lds_sync();      // (the previous round's slots and staging rows have been read by everybody)
                H[lane] = 0;
                if (lane < 32) SLOT[lane] = make_int4(a.n_traj, 0, 0, 0);                    // (rank beyond the window's trajectories: nobody)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = off[j + 1] - off[j];
                    key[j] = t0 + j < a.n_traj ? (c < 0 ? 0 : (c > 63 ? 63 : c)) : -1;
                }
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (key[j] >= 0) atomicAdd(&H[key[j]], 1);
                wave_lds_sync();
                const int tot = H[lane];
                const int pre = wave_incl_scan_add(tot);                                      // trajectories with a count <= this lane's (lane = count)
                const int start = __builtin_amdgcn_readlane(pre, 63) - pre;                    // ... with a larger one: the first rank of this count
                const int r0 = 32 * q;
                unsigned long long rel = __ballot(tot > 0 && start < r0 + 32 && start + tot > r0);
                while (rel != 0ull) {            // (wave-uniform)
                    const int k = __ffsll((long long)rel) - 1;
                    rel &= rel - 1ull;
                    const int st_k = __builtin_amdgcn_readlane(start, k);
                    int before[8], cnt = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { before[j] = cnt; cnt += key[j] == k ? 1 : 0; }
                    const int excl = wave_incl_scan_add(cnt) - cnt;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pos = st_k + excl + before[j] - r0;
                        if (key[j] == k && pos >= 0 && pos < 32) SLOT[pos] = make_int4(t0 + j, off[j], off[j + 1] - off[j], 0);
                    }
                }
                wave_lds_sync();
                const int4 rec = SLOT[item];
                b = rec.x; s0 = rec.y; M = rec.z;
                packed = true;
                wave_lds_sync();      // (the slots are read before the staging below writes over them)
            } else if (b < a.n_traj) {
                if (a.perm4) {   // {trajectory, first segment, segment count}: one load (the sort kernel had all three in hand)
                    const int4 rec = a.perm4[b];
                    b = rec.x; s0 = rec.y; M = rec.z;
                    packed = true;
                } else {
                    b = a.perm[b];
                }
            }
        }
        const bool live = b < a.n_traj;   // (both lanes of a pair agree)
        if (live && !packed) {
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        }
        const bool shape_ok = live && (M >= 1) && (M <= Mx);
        const int Mc = shape_ok ? M : 0;

        G2_STAMP(1);
        // ---- the wave's inputs -> LDS (slot i: waypoint rows 0..M, then durations)
        // 8 lanes per trajectory (128 contiguous bytes per load instruction and trajectory), 8 trajectories per pass, 4 passes of
        // up to 8 steps: all 32 loads of a lane are in flight before the first LDS write.  (LDS-DMA -- global_load_lds -- would
        // need one instruction per trajectory and run, and costs ~270 cycles of issue each: measured 17-23 k cycles per wave.)
        // A 16-byte unit that would reach past an odd-sized run is loaded as 8 bytes; the runs are only 8-byte aligned.
        {
            // (the 16-byte loads are spelled as aligned ones: global_load_dwordx4 only needs dword alignment in hardware, while an
            // honest aligned(8) vector type makes the compiler split every load in two)
            const int w = lane & 7;
            double2 stg[4][8];
            int dsto[4][8];    // LDS destination (doubles), -1 = nothing
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int slot = pass * 8 + (lane >> 3);
                const int Ms = __shfl(Mc, 2 * slot), s0s = __shfl(s0, 2 * slot), bs = __shfl(b, 2 * slot);
                const int nwd = 3 * (Ms + 1), nw = (nwd + 1) >> 1, nt = (Ms + 1) >> 1;   // doubles / 16-byte units of the two runs
                const int c = Ms > 0 ? nw + nt : 0;
                const double* gw = a.waypoints + 3 * (size_t)(s0s + bs);
                const double* gt = a.times + s0s;
                auto fetch = [&](int u, double2& val, int& dst) {
                    dst = -1;
                    val = make_double2(0.0, 0.0);
                    if (u < c) {
                        const bool isT = u >= nw;
                        const int e = isT ? 2 * (u - nw) : 2 * u;                 // first double of the unit within its run
                        const double* g = isT ? gt + e : gw + e;
                        const bool tail = e + 1 >= (isT ? Ms : nwd);
                        if (tail) val.x = g[0];
                        else val = *reinterpret_cast<const double2*>(g);
                        dst = slot * stride + (isT ? TOFF : 0) + e;
                    }
                };
#pragma unroll
                for (int st = 0; st < 8; ++st) fetch(st * 8 + w, stg[pass][st], dsto[pass][st]);
            }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    if (dsto[pass][st] >= 0) *reinterpret_cast<double2*>(s_in + dsto[pass][st]) = stg[pass][st];
            // more than 64 units per trajectory (M > 31 -- not the shape this path is tuned for): one round trip per extra step
            if (Mx > 31) {
                for (int pass = 0; pass < 4; ++pass) {
                    const int slot = pass * 8 + (lane >> 3);
                    const int Ms = __shfl(Mc, 2 * slot), s0s = __shfl(s0, 2 * slot), bs = __shfl(b, 2 * slot);
                    const int nwd = 3 * (Ms + 1), nw = (nwd + 1) >> 1, nt = (Ms + 1) >> 1;
                    const int c = Ms > 0 ? nw + nt : 0;
                    const double* gw = a.waypoints + 3 * (size_t)(s0s + bs);
                    const double* gt = a.times + s0s;
                    const int cmax = wave_max_int(c);
                    for (int u0 = 64; u0 < cmax; u0 += 8) {
                        const int u = u0 + w;
                        if (u < c) {
                            const bool isT = u >= nw;
                            const int e = isT ? 2 * (u - nw) : 2 * u;
                            const double* g = isT ? gt + e : gw + e;
                            double* d = s_in + slot * stride + (isT ? TOFF : 0) + e;
                            d[0] = g[0];
                            if (e + 1 < (isT ? Ms : nwd)) d[1] = g[1];
                        }
                    }
                }
            }
        }
        G2_STAMP(2);
        const double* __restrict__ bc = a.bc + (size_t)(live ? b : 0) * 2 * ND * 3;
        double* __restrict__ out = a.coeff + (size_t)3 * NC * s0;
        const int m = isR ? Mc / 2 : (Mc + 1) / 2;   // own segments; the meeting knot is own knot m (for M = 1: the end knot itself)
        // own frame (the reversed lane counts segments and knots from the end); indices clamped: reads beyond the own half
        // return something finite-or-not that is never used
        auto Tof = [&](int j) -> double { const int q = isR ? M - 1 - j : j; return sl[TOFF + (q < 0 ? 0 : (q >= Mx ? Mx - 1 : q))]; };
        auto wrow = [&](int j, double (&p)[3]) {
            const int q = isR ? M - j : j;
            const double* w = sl + 3 * (q < 0 ? 0 : (q > Mx ? Mx : q));
            p[0] = w[0]; p[1] = w[1]; p[2] = w[2];
        };

        // own boundary knot, own frame: y'_0 = F y_M for the reversed lane
        double h_prev[ND][3], E_prev[ND][ND], y0[ND][3], yend[ND][3];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int c = 0; c < ND; ++c) E_prev[i][c] = 0.0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double v = bc[(isR * ND + i) * 3 + ax];
                yend[i][ax] = bc[(ND + i) * 3 + ax];
                h_prev[i][ax] = (isR && ((i & 1) == 0)) ? -v : v;
                y0[i][ax] = h_prev[i][ax];
            }
        }
        wait_vmcnt0();
        wave_lds_sync();
        G2_STAMP(3);

        // ---------------- elimination of the own interior knots j = 1..m-1 ----------------
        SegBlocks<R> sa;
        double pb[3], dpa[3], Tn, pn[3];
        bool okT;
        {
            const double T0 = Tof(0);
            okT = (int)(T0 > 0.0) & (int)(T0 < INFINITY);
            sa.build(T0);
            double p0[3];
            wrow(0, p0);
            wrow(1, pb);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dpa[ax] = pb[ax] - p0[ax];
            Tn = Tof(1);
            wrow(2, pn);
        }
        // One elimination step (own knot j): updates the carried state and leaves the knot's record (E_j, h_j) in rec.
        auto elim_step = [&](const int j, double (&rec)[F]) {
            const double Tj = Tn;
            double pc[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) pc[ax] = pn[ax];
            Tn = Tof(j + 1);
            wrow(j + 2, pn);
            okT = (int)okT & (int)(Tj > 0.0) & (int)(Tj < INFINITY);
            SegBlocks<R> sb;
            sb.build(Tj);
            double dpb[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                dpb[ax] = pc[ax] - pb[ax];
                pb[ax] = pc[ax];
            }
            double S[ND][ND], z[ND][3];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c < ND; ++c) S[i][c] = sa.A11[i][c] + sb.A00(i, c);
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) z[i][ax] = sb.gv(i) * dpb[ax] - sa.gw[i] * dpa[ax];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int q = 0; q < ND; ++q) {
#pragma unroll
                    for (int c = 0; c <= i; ++c) S[i][c] -= sa.A01[q][i] * E_prev[q][c];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) z[i][ax] -= sa.A01[q][i] * h_prev[q][ax];
                }
            G2_INV<ND> inv;
            inv.factor(S);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double col[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) col[i] = z[i][ax];
                inv.solve(col);
#pragma unroll
                for (int i = 0; i < ND; ++i) h_prev[i][ax] = col[i];
            }
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                double col[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) col[i] = sb.A01[i][c];
                inv.solve(col);
#pragma unroll
                for (int i = 0; i < ND; ++i) E_prev[i][c] = col[i];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c < ND; ++c) rec[i * ND + c] = E_prev[i][c];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) rec[ND * ND + i * 3 + ax] = h_prev[i][ax];
            }
            sa = sb;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dpa[ax] = dpb[ax];
        };
        // The records of the LAST NKR own knots (the ones next to the meeting knot: the first the backward sweep needs) stay in
        // registers -- NKR peeled steps behind the loop, NKR peeled trips in front of the backward loop; only the knots before them
        // go through the HBM workspace.  With one wave per SIMD the register file has 512 registers per lane and the compiler parks
        // what does not fit the 256 architectural ones in the accumulator registers (v_accvgpr_write / _read: two instructions per
        // double and direction instead of a 16-byte store, a 16-byte load and their ~1-2 us round trip through L2 / HBM).
        // Config 4 (own eliminated knots per lane 1..11) is HBM-bound on its ACTUAL traffic -- 213 MB per dispatch at 5 TB/s with
        // NKR = 3 (algorithmic: 109 MB) --, so every record kept on chip is time: NKR = 3 / 5 / 6 / 7 -> 43.8 / 39.3 / 38.4 / 38.6 us
        // per launch, 213 / 183 / 171 / 162 MB (r = 4: 7 records = 252 accumulator registers, no scratch; r = 3 records are 10 doubles).
#ifndef G2_NKR
#define G2_NKR (R == 3 ? 11 : 7)
#endif
        constexpr int NKR = G2_NKR;
        const int ne = m > 0 ? m - 1 : 0;                  // own eliminated knots 1..ne
        const int n_ws = ne > NKR ? ne - NKR : 0;          // ... of which 1..n_ws go through the workspace
        // (a plain per-lane loop: lanes that are done are masked off and keep their state in place -- a wave-uniform loop with
        // the body under `if (j < m)` makes the compiler copy all ~50 loop-carried doubles twice per trip)
        for (int j = 1; j <= n_ws; ++j) {
            double rec[F];
            elim_step(j, rec);
#pragma unroll
            for (int f2 = 0; f2 < F / 2; ++f2) G2_WS_ST(W2(j, f2), make_double2(rec[2 * f2], rec[2 * f2 + 1]));
        }
        double RK[NKR][F];   // records of own knots ne, ne - 1, .., ne - NKR + 1
#pragma unroll
        for (int k = 0; k < NKR; ++k)
#pragma unroll
            for (int f = 0; f < F; ++f) RK[k][f] = 0.0;
#pragma unroll
        for (int k = NKR - 1; k >= 0; --k)
            if (ne >= k + 1) elim_step(ne - k, RK[k]);
        G2_STAMP(4);
        // every duration of the trajectory has been seen by one of the two lanes
        bool ok;
        {
            const int oki = (shape_ok && (okT || m == 0)) ? 1 : 0;
            ok = (oki & __builtin_amdgcn_mov_dpp(oki, 0xB1, 0xF, 0xF, true)) != 0;
        }
        if (live && !ok && !isR && a.status) a.status[b] = UAVQP_INVALID_INPUT;
        const int mm = ok ? m : 0;   // an invalid trajectory emits nothing

        // ---------------- meeting knot: own partial Schur complement, exchange, solve ----------------
        // sa = blocks of the last own segment (m-1); E_prev / h_prev belong to own knot m-1 (or the boundary knot).
        double ynext[ND][3];   // solution at the meeting knot, own frame
        {
            double P[ND][ND], zp[ND][3];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c < ND; ++c) P[i][c] = sa.A11[i][c];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) zp[i][ax] = -sa.gw[i] * dpa[ax];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int q = 0; q < ND; ++q) {
#pragma unroll
                    for (int c = 0; c <= i; ++c) P[i][c] -= sa.A01[q][i] * E_prev[q][c];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) zp[i][ax] -= sa.A01[q][i] * h_prev[q][ax];
                }
            double S[ND][ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
#pragma unroll
                for (int c = 0; c <= i; ++c) {
                    const double o = swap_pair(P[i][c]);
                    S[i][c] = P[i][c] + (((i + c) & 1) ? -o : o);   // P_own + F P_other F
                }
#pragma unroll
                for (int c = i + 1; c < ND; ++c) S[i][c] = 0.0;     // upper triangle is never read
            }
            G2_INV<ND> inv;
            inv.factor(S);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double col[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const double o = swap_pair(zp[i][ax]);
                    col[i] = zp[i][ax] + ((i & 1) ? o : -o);       // z_own + F z_other, F_ii = (-1)^(i+1)
                }
                inv.solve(col);
#pragma unroll
                for (int i = 0; i < ND; ++i) ynext[i][ax] = (M == 1) ? yend[i][ax] : col[i];   // M = 1: the end knot is given
            }
        }

        G2_STAMP(5);
        // ---------------- back-substitution + emission of the own segments j = m-1 .. 0 ----------------
        // Every lane runs every trip: a lane that has emitted its last segment (j < 0) goes on with clamped indices and harmless
        // values and its stores go to the sink -- nothing of its state is needed any more, so the loop body is not divergent.
        // Workspace records: two register sets, trips alternate between them, and a set is re-loaded (for the trip after the
        // next) as soon as its trip has consumed it.  vmcnt retires loads and stores in issue order, so waiting for a load also
        // waits for every store issued before it: with the record fetched one trip ahead each trip waited for the coefficient
        // stores of the trip before (measured 5.2 k cycles per trip against 1.9 k of issue); now they have a trip to drain.
        // No register copies between the sets: a copy would have to wait for the loads it reads.
        bool finite = true;
        {
            double recA[F], recB[F], Tc, pa[3], pe[3];   // records of own knots mm-1-i (trip i even: A, odd: B), data of the trip's segment
            {
                const int j1 = mm >= NKR + 2 ? mm - 1 - NKR : 1, j2 = mm >= NKR + 3 ? mm - 2 - NKR : 1;   // trips 0..NKR-1 take their records from registers
                load_rec(j1, recA);
                load_rec(j2, recB);
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("" : "+v"(recA[f]));   // retired before the loop: no wait merged into its top
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("" : "+v"(recB[f]));
                Tc = Tof(mm - 1);
                wrow(mm - 1, pa);
                wrow(mm, pe);
            }
            G2_ACC_DECL
            auto trip = [&](const int i, double (&rec)[F], double (&other)[F], auto touch, auto reload) {
                constexpr bool TOUCH = decltype(touch)::value;
                constexpr bool RELOAD = decltype(reload)::value;   // false: a register-resident record (peeled trips), nothing to fetch
                G2_ACC_START;
                const int j = mm - 1 - i;   // own segment / own knot of this trip
                double Tn2, pan[3];
                Tn2 = Tof(j - 1);
                wrow(j - 1, pan);
                double y[ND][3];
#pragma unroll
                for (int q = 0; q < ND; ++q)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        double acc = rec[ND * ND + q * 3 + ax];
#pragma unroll
                        for (int c = 0; c < ND; ++c) acc -= rec[q * ND + c] * ynext[c][ax];
                        y[q][ax] = (j == 0) ? y0[q][ax] : acc;
                    }
                __builtin_amdgcn_sched_barrier(0);
                G2_ACC(0);
                if constexpr (RELOAD) {   // the record of the trip after the next (own knot j - 2)
                    const int j2 = j >= 3 ? j - 2 : 1;
                    load_rec(j2, rec);
                }
                __builtin_amdgcn_sched_barrier(0);
                const double itj = fast_rcp(Tc);
                const int seg = isR ? M - 1 - j : j;
                // the lane's 3 x 2r coefficients -> its LDS row; the wave then writes them out 64 (r = 3: 48) contiguous bytes --
                // a (trajectory, axis, segment) chunk -- per 4 (3) lanes: a scattered 16-byte store per lane costs the L1 one
                // transaction per lane (measured +22 us on config 4 against +3 us for lane-linear stores of the same bytes)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    double ys[ND], ye[ND], c[NC];
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const double fs = ((d & 1) == 0) ? -1.0 : 1.0;
                        ys[d] = isR ? fs * ynext[d][ax] : y[d][ax];
                        ye[d] = isR ? fs * y[d][ax] : ynext[d][ax];
                    }
                    segment_coeffs<R>(isR ? pe[ax] : pa[ax], ys, isR ? pa[ax] : pe[ax], ye, Tc, itj, c);
#pragma unroll
                    for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(s_out + lane * G2_ROW + ax * NC + q) = make_double2(c[q], c[q + 1]);
                    finite = (int)finite & ((int)(j < 0) | ((int)(fabs(c[NC - 1]) < INFINITY) & (int)(fabs(c[R]) < INFINITY)));
                }
                {
                    const unsigned long long a0 = j >= 0 ? reinterpret_cast<unsigned long long>(out + (size_t)seg * NC) : reinterpret_cast<unsigned long long>(a.dummy);
                    const unsigned long long st = j >= 0 ? (unsigned long long)((size_t)M * NC * sizeof(double)) : 0ull;
                    *reinterpret_cast<ulonglong2*>(s_meta + 2 * lane) = make_ulonglong2(a0, st);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) ynext[d][ax] = y[d][ax];
                Tc = Tn2;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) { pe[ax] = pa[ax]; pa[ax] = pan[ax]; }
                G2_ACC(1);
                wave_lds_sync();
                if constexpr (TOUCH) {
                    // trip B: retire set A's reload (issued in trip A, before A's stores and B's reload) before this trip's stores
#pragma unroll
                    for (int f = 0; f < F; ++f) asm volatile("" : "+v"(other[f]));
                }
                // chunk u = (axis, source lane, 16-byte piece), piece fastest: NC = 8: 4 lanes write the 64 bytes of a
                // (trajectory, axis, segment); NC = 6: 3 lanes write its 48 bytes.  All LDS reads first, then the stores.
                constexpr int PPC = NC / 2, NST = 3 * PPC;       // pieces per chunk, store instructions per trip
                G2_ACC(2);
                ulonglong2 mt[NST];
                double2 v[NST];
                int axs[NST];
#pragma unroll
                for (int sidx = 0; sidx < NST; ++sidx) {
                    const int u = sidx * 64 + lane;
                    const int ax = u / (64 * PPC), rem = u - ax * (64 * PPC), src = rem / PPC, piece = rem - PPC * src;
                    axs[sidx] = ax;
                    mt[sidx] = *reinterpret_cast<const ulonglong2*>(s_meta + 2 * src);
                    v[sidx] = *reinterpret_cast<const double2*>(s_out + src * G2_ROW + ax * NC + 2 * piece);
                    mt[sidx].x += 16ull * piece;
                }
                __builtin_amdgcn_sched_barrier(0);
                G2_ACC(3);
#pragma unroll
                for (int sidx = 0; sidx < NST; ++sidx) G2_OUT_ST2(mt[sidx].x + axs[sidx] * mt[sidx].y, v[sidx]);
                if constexpr (TOUCH && RELOAD) {
                    // ... and its own reload (issued before these stores), so that no load is pending over the loop's back-edge,
                    // where the compiler would wait with vmcnt(0), stores included
#pragma unroll
                    for (int f = 0; f < F; ++f) asm volatile("" : "+v"(rec[f]));
                }
                wave_lds_sync();
                G2_ACC(4);
            };
#pragma unroll
            for (int k = 0; k < NKR; ++k)
                if (__ballot(k < mm) != 0ull) trip(k, RK[k], RK[k], std::false_type{}, std::false_type{});
            for (int i = NKR; __ballot(i < mm) != 0ull; i += 2) {
                trip(i, recA, recB, std::false_type{}, std::true_type{});
                if (__ballot(i + 1 < mm) == 0ull) break;
                trip(i + 1, recB, recA, std::true_type{}, std::true_type{});
            }
            G2_ACC_FLUSH;
        }
        G2_STAMP(6);
#ifdef G2_TIMING
        if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(a.dummy)[128 + 8] = mm;
#endif
        {
            const int f = finite ? 1 : 0;
            const bool fin = (f & __builtin_amdgcn_mov_dpp(f, 0xB1, 0xF, 0xF, true)) != 0;
            if (ok && !isR && a.status) a.status[b] = fin ? UAVQP_SOLVED : UAVQP_NON_FINITE;
        }
    }
}

}  // namespace uavqp
