// qp_wave_utils.h -- single-wave hand-over fences, wait counts, 16-byte stores and the HBM -> LDS DMA helpers every kernel family uses
// (moved out of qp_twisted.h in round 6: each family is its own translation unit and includes what it needs).
#pragma once
#include <hip/hip_runtime.h>

namespace uavqp {

// Single-wave workgroups: LDS operations of one wave execute in issue order, so cross-lane hand-offs
// through LDS need no s_barrier and -- unlike __syncthreads() -- must NOT wait for outstanding global
// stores (vmcnt).  This only pins the compiler's ordering of LDS accesses.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);  // also keep arithmetic of the next unit below: bounds register pressure
}

// s_waitcnt vmcnt(0) through the builtin (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15) so that the
// compiler's own wait-count bookkeeping knows that pending LDS-DMA writes have landed.
__device__ __forceinline__ void wait_vmcnt0() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}

// 16-B coefficient stores.
//   store_pair_wt: write-through (sc0 sc1) for the line-complete chunk-mode stores -- the lines do not stay
//     dirty in L2, so the write-back that otherwise piles up at the kernel boundary overlaps the kernel
//     (measured: 16 k batch 11.5 -> 9.9 us, 64 k 25.0 -> 22.9 us, 1 M 409 -> 399 us);
//   store_pair: plain stores for the latency shape, whose 16-B pieces are partial lines that L2 has to merge
//     first (write-through there: 6.0 -> 7.9 us).  Non-temporal stores change nothing either way.
__device__ __forceinline__ void store_pair(double* dst, double2 v) { *reinterpret_cast<double2*>(dst) = v; }
__device__ __forceinline__ void store_pair_wt(double* dst, double2 v) {
    typedef double wt_v2 __attribute__((ext_vector_type(2)));
    wt_v2 w = {v.x, v.y};
    // (s_nop 1: the store reads its data registers after issue -- two wait states before anything may overwrite them on gfx940+,
    //  and the compiler cannot see through the string)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
}

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// Asynchronous HBM -> LDS copy of ND_ doubles (full tile): one global_load_lds_dwordx4 per 1 KiB.  AUX = 2: non-temporal (the
// latency shapes read every input byte exactly once: measured 5.49 -> 5.24 us on the 4096 batch, tools/ubench/tw).
template <int ND_, int AUX = 0>
__device__ __forceinline__ void dma_tile(const double* __restrict__ g, double* s, int lane) {
    constexpr int NP = ND_ / 2, NL = (NP + 63) / 64;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int p = lane + 64 * k;
        if (p < NP) __builtin_amdgcn_global_load_lds((gas_ptr)(g + 2 * p), (las_ptr)(s + 128 * k), 16, 0, AUX);
    }
}

// Guarded synchronous copy for the partial last tile; entries at or beyond n_valid become `fill`.
template <int ND_>
__device__ __forceinline__ void load_tile_guarded(const double* __restrict__ g, int n_valid, double* s, int lane, double fill) {
    for (int i = lane; i < ND_; i += 64) s[i] = (i < n_valid) ? g[i] : fill;
}

// wave-uniform maximum of a per-lane integer in [0, 256): eight ballots
__device__ __forceinline__ int wave_max_int(int v) {
    int r = 0;
#pragma unroll
    for (int bit = 7; bit >= 0; --bit) {
        const int t = r | (1 << bit);
        if (__ballot(v >= t) != 0ull) r = t;
    }
    return r;
}

}  // namespace uavqp
