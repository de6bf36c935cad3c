// uavqp.hip -- HIP kernels + C ABI (include/uavqp.h) of the batched min-jerk / min-snap QP back-end.
// gfx950 only.  The product path has no CPU fallback: every entry point fails with a negative code
// when no device is usable.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/uavqp.h"
#include "qp_device.h"
#include "uavqp_comm.h"

#include "qp_core_kernels.h"


#include "qp_twisted.h"
#include "qp_generic2.h"
#include "qp_corridor.h"
#include "qp_corridor_dual.h"
#include "qp_rows.h"
#include "qp_rows2.h"
#include "qp_rows_dual.h"
#include "obstacle_grid.h"
// Measured-slower alternatives kept as bit-identical cross-checks: `make experiments` (-DUAVQP_EXPERIMENTS) only
#ifdef UAVQP_EXPERIMENTS
#include "cloud_grid2d.h"
#include "qp_corridor_lane.h"
#endif

// the solver families are compiled as their own translation units (k_*.hip): here their instantiations are only DECLARED
// (-DUAVQP_SINGLE_TU: a build that includes this file whole -- tools/ubench/twisted_phases.hip -- instantiates everything itself)
#ifndef UAVQP_SINGLE_TU
#include "kernel_instances.h"
UAVQP_INSTANCES_TWISTED3
UAVQP_INSTANCES_TWISTED4
UAVQP_INSTANCES_GENERIC
UAVQP_INSTANCES_CORRIDOR
UAVQP_INSTANCES_CORRIDOR_DUAL
UAVQP_INSTANCES_ROWS31
UAVQP_INSTANCES_ROWS32
UAVQP_INSTANCES_ROWS41
UAVQP_INSTANCES_ROWS42
UAVQP_INSTANCES_ROWS_DUAL
UAVQP_INSTANCES_CLOUD
#endif

namespace uavqp {
// Specialised (R, M) instantiations of the register-resident kernel; everything else takes the generic one.
typedef void (*twisted_fn)(BatchArgs);
template <int R, int M>
static twisted_fn twisted_ptr(int tile) {
    if (tile == 4) return &solve_twisted_kernel<R, M, 4, 16>;  // two lane pairs per axis: emission split in two (smallest batches)
    if (tile == 8) return &solve_twisted_kernel<R, M, 8, 8>;  // one lane pair per axis (latency shape)
    return tile == 16 ? &solve_twisted_kernel<R, M, 16> : &solve_twisted_kernel<R, M, 32>;
}
static twisted_fn find_twisted(int r, int M, int tile) {
#define UAVQP_CASE(RR, MM) if (r == RR && M == MM) return twisted_ptr<RR, MM>(tile);
    UAVQP_CASE(4, 2) UAVQP_CASE(4, 3) UAVQP_CASE(4, 4) UAVQP_CASE(4, 5) UAVQP_CASE(4, 6) UAVQP_CASE(4, 7)
    UAVQP_CASE(4, 8) UAVQP_CASE(4, 9) UAVQP_CASE(4, 10) UAVQP_CASE(4, 12)
    UAVQP_CASE(3, 2) UAVQP_CASE(3, 3) UAVQP_CASE(3, 4) UAVQP_CASE(3, 5) UAVQP_CASE(3, 6) UAVQP_CASE(3, 7)
    UAVQP_CASE(3, 8) UAVQP_CASE(3, 10) UAVQP_CASE(3, 12) UAVQP_CASE(3, 16)
#undef UAVQP_CASE
    return nullptr;
}
}  // namespace uavqp

namespace uavqp {
// "everything before me on this stream is done": (seq << 32 | *value) into a word of host-coherent pinned memory (value may be null)
// Release side of the hand-over: the system-scope fence BEFORE the store orders everything this stream wrote before the kernel boundary
// (coefficients, statuses: visible to this kernel) ahead of the word; the one behind it pushes the word itself out.
__global__ void host_word_kernel(const int32_t* __restrict__ value, volatile unsigned long long* slot, unsigned int seq) {
    const unsigned int v = value ? (unsigned int)*value : 0u;
    __threadfence_system();
    *slot = ((unsigned long long)seq << 32) | v;
    __threadfence_system();
}
}  // namespace uavqp

// ===================================================================================================
// C ABI
// ===================================================================================================
using uavqp::BatchArgs;

static thread_local std::string g_last_error;

// The host's side of a word the device writes into host-coherent pinned memory (host_word_kernel, compact_order_kernel): poll until its
// upper half is the sequence number awaited.  Bounded: every 65 536 polls the stream is asked -- a failed stream, an idle stream without
// the word, or 60 s without either end the wait.  (Against hipStreamSynchronize / an event: no packet in the stream, and the host
// reacts within the PCIe latency of the store instead of the runtime's wake-up: 17.8 -> 14.6 us for a one-trajectory call.)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}
static int await_host_word(hipStream_t s, volatile unsigned long long* w, unsigned int want, unsigned int* value, const char* who) {
    unsigned long long word = *w;
    const auto t_begin = std::chrono::steady_clock::now();
    for (long long spin = 0; (unsigned int)(word >> 32) != want; ++spin) {
        if ((spin & 0xFFFF) == 0xFFFF) {
            const hipError_t q = hipStreamQuery(s);
            const char* what = nullptr;
            if (q != hipSuccess && q != hipErrorNotReady) what = ": the stream failed while a word from the device was awaited";
            else if (q == hipSuccess && (unsigned int)(*w >> 32) != want) what = ": a word from the device never arrived";
            else if (std::chrono::steady_clock::now() - t_begin > std::chrono::seconds(60)) {
                // a long queue in front of the call (a user stream, a very large batch) is not an error: stop polling, wait for the stream the
                // ordinary way -- nothing of this call may still be running against the caller's buffers when the call returns
                const hipError_t e = hipStreamSynchronize(s);
                word = *w;
                if (e == hipSuccess && (unsigned int)(word >> 32) == want) break;
                what = e == hipSuccess ? ": a word from the device never arrived" : ": the stream failed while a word from the device was awaited";
            }
            if (what) {
                (void)hipStreamSynchronize(s);   // (error path: no kernel of this call may outlive it)
                g_last_error = std::string(who) + what;
                return UAVQP_ERR_HIP;
            }
        }
        cpu_relax();
        word = *w;
    }
    std::atomic_thread_fence(std::memory_order_acquire);   // acquire side: the page's payload is read only after the word
    if (value) *value = (unsigned int)(word & 0xFFFFFFFFull);
    return UAVQP_OK;
}

struct uavqp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int variant = 0;
    int tile_override = 0;  // fixed tile shape of the specialised kernel (variants 4 / 8 / 16 / 32)
    uavqp_settings settings{};
    int num_cus = 256;
    double* ws = nullptr;
    size_t ws_bytes = 0;
    uavqp::Comm comm;         // RCCL communicator of the multi-GPU entry points (uavqp_comm_create)
    void* rows_warm2 = nullptr;      // rows part of the starting set of the general-rows solve + the "box phase needed" flags (qp_rows_dual.h)
    size_t rows_warm2_bytes = 0;
    void* dbg_queue = nullptr;  // (debug builds) where the last corridor solve kept its work counter
    int dual_trips_extra = 0;   // added to the trip budget 4 n + 16 of the dual preludes (UAVQP_DUAL_TRIPS_EXTRA, may be negative: tools/ experiments)
    int wave_prelude = 2;       // re-solves with G in the cache: 2 = corridor_dual_wave2_kernel (two trajectories per wave), 1 = corridor_dual_wave_kernel (one),
                                // 0 = the batch kernels (UAVQP_WAVE_PRELUDE=0|1|2, UAVQP_NO_WAVE_PRELUDE: A/B runs)
    void* dbg_dual = nullptr;   // (UAVQP_DUAL_DEBUG builds) dump area of corridor_dual_kernel
    void* dbg_guess = nullptr;  // (UAVQP_DUAL_DEBUG builds) the starting sets of the last cold corridor solve
    int32_t* perm = nullptr;  // ragged dealing permutation (window_sort_kernel); behind it the packed {b, s0, M, 0} records
    size_t perm_count = 0;
    uint64_t* rows_warm = nullptr;   // [n_traj][3][2] working set of the box-only phase of uavqp_solve_rows_batch_device
    size_t rows_warm_count = 0;
    double* dummy = nullptr;
    int deal_tickets = 1;            // rows_dual_kernel deals its trajectories by ticket (UAVQP_DEAL_TICKETS=0: round-robin as in rounds 4-5, A/B runs)
    // staging buffers of the host-pointer entry points
    void* d_stage = nullptr;
    size_t stage_bytes = 0;
    // pinned, device-mapped page of the single-axis entry point (uavqp_solve_axis_host): the kernel reads the inputs of the one
    // trajectory from it and writes coefficients and status into it -- no copy calls on that path
    void* h_axis = nullptr;
    void* d_axis = nullptr;
    size_t axis_bytes = 0;
    // corridor pipeline (uavqp_pipeline.h): device scratch (counters, flags, working sets) and the pinned page its counters are read through
    void* d_pipe = nullptr;
    size_t pipe_bytes = 0;
    void* h_pipe = nullptr;
    unsigned int pipe_seq = 0;   // sequence number of the last pipeline round enqueued (uavqp_pipeline.h: tags the count a round reports to the host)
};

#define UAVQP_HIP(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return UAVQP_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

#ifndef UAVQP_SRC_HASH
#define UAVQP_SRC_HASH "unknown"
#endif
#ifdef UAVQP_EXPERIMENTS
#define UAVQP_BUILD_TAG ", experiments"
#else
#define UAVQP_BUILD_TAG ""
#endif
extern "C" const char* uavqp_version(void) { return "uavqp 0.6.0 (gfx950, float64, src " UAVQP_SRC_HASH UAVQP_BUILD_TAG ")"; }

extern "C" void uavqp_default_settings(uavqp_settings* out) {
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    out->struct_size = (int32_t)sizeof(uavqp_settings);
    out->warm_start = 1;        // minimum_control.cpp:160
    out->eps_prim_inf = 1e-3;   // minimum_control.cpp:161
    out->max_iter = 0;          // automatic (8 M + 20); the reference's 1000 ADMM iterations (:162) map to an active-set cap
    out->kernel_variant = 0;
    out->ragged_window_sort = 1;
    out->generic_lanes_per_traj = 0;
    out->generic_waves_per_cu = 0;
    out->corridor_pdas_rounds = 3;
    out->corridor_pdas_rounds_warm = 0;
    out->cloud_window = 1;
    out->corridor_tail_shape = 1;
    out->corridor_initial_guess = 2;
    out->rows_lanes_per_problem = 0;
    out->realloc_dead_band = 1.01;
    out->realloc_overshoot = 1.02;
}

static int apply_variant(uavqp_ctx* ctx, int variant);

extern "C" int uavqp_set_settings(uavqp_ctx* ctx, const uavqp_settings* st) {
    if (!ctx || !st || st->struct_size != (int32_t)sizeof(uavqp_settings)) return UAVQP_ERR_INVALID_ARG;
    if (!(st->eps_prim_inf >= 0.0) || !(st->realloc_dead_band >= 1.0) || !(st->realloc_overshoot >= 1.0) || !(st->realloc_dead_band < INFINITY) ||
        !(st->realloc_overshoot < INFINITY) || st->corridor_pdas_rounds < 0 || st->corridor_pdas_rounds > 64 || st->corridor_pdas_rounds_warm < 0 || st->corridor_pdas_rounds_warm > 64 ||
        (st->generic_lanes_per_traj != 0 && st->generic_lanes_per_traj != 1 && st->generic_lanes_per_traj != 2 && st->generic_lanes_per_traj != 3) || st->generic_waves_per_cu < 0 ||
        st->generic_waves_per_cu > 32 || st->rows_lanes_per_problem < 0 || st->rows_lanes_per_problem > 2 ||
        false)
        return UAVQP_ERR_INVALID_ARG;
#ifndef UAVQP_EXPERIMENTS
    // the experimental paths (cloud_window 2 / 3: cloud_grid2d.h; corridor_prelude_lanes 1: qp_corridor_lane.h) exist in `make experiments` builds only
    if (st->cloud_window == 2 || st->cloud_window == 3 || st->corridor_prelude_lanes == 1) {
        g_last_error = "uavqp_set_settings: cloud_window 2 / 3 and corridor_prelude_lanes 1 select experimental kernels; this library was built without -DUAVQP_EXPERIMENTS";
        return UAVQP_ERR_INVALID_ARG;
    }
#endif
    const int rc = apply_variant(ctx, st->kernel_variant);
    if (rc != UAVQP_OK) return rc;
    ctx->settings = *st;
    ctx->settings.warm_start = st->warm_start ? 1 : 0;
    ctx->settings.ragged_window_sort = st->ragged_window_sort == 2 ? 2 : (st->ragged_window_sort ? 1 : 0);     // (2: the dealing as its own launch, rounds 2-5)
    ctx->settings.corridor_initial_guess = st->corridor_initial_guess < 0 ? 0 : (st->corridor_initial_guess > 2 ? 2 : st->corridor_initial_guess);
    // (cloud_window: any value outside 0..3 means "on" = 1, as before round 5; corridor_prelude_lanes sits in what used to be padding: a value
    //  that is none of 0 / 1 / 8 -- a struct filled field by field without uavqp_default_settings -- is taken as the default, not refused)
    ctx->settings.cloud_window = (st->cloud_window >= 0 && st->cloud_window <= 3) ? st->cloud_window : 1;
    ctx->settings.corridor_prelude_lanes = (st->corridor_prelude_lanes == 1 || st->corridor_prelude_lanes == 8) ? st->corridor_prelude_lanes : 0;
    ctx->settings.corridor_tail_shape = st->corridor_tail_shape ? 1 : 0;
    return UAVQP_OK;
}

extern "C" int uavqp_get_settings(const uavqp_ctx* ctx, uavqp_settings* out) {
    if (!ctx || !out) return UAVQP_ERR_INVALID_ARG;
    *out = ctx->settings;
    return UAVQP_OK;
}
extern "C" const char* uavqp_last_error(void) { return g_last_error.c_str(); }

extern "C" int uavqp_create(uavqp_ctx** out_ctx, int device) {
    if (!out_ctx) return UAVQP_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_last_error = "no HIP device visible (the uavqp product path has no CPU fallback)";
        return UAVQP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        g_last_error = "device index out of range";
        return UAVQP_ERR_INVALID_ARG;
    }
    UAVQP_HIP(hipSetDevice(device));
    uavqp_ctx* ctx = new (std::nothrow) uavqp_ctx();
    if (!ctx) return UAVQP_ERR_ALLOC;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    // A BLOCKING stream: ordered with the legacy default (NULL) stream, which is where a caller that never thinks about
    // streams (and PyTorch by default) fills and reads the buffers it hands over.  Callers that want overlap pass their
    // own stream through uavqp_set_stream.
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamDefault);
    if (e != hipSuccess) {
        g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return UAVQP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    e = hipMalloc((void**)&ctx->dummy, 4096);
    if (e != hipSuccess) {
        g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return UAVQP_ERR_ALLOC;
    }
    // The environment is read HERE, once, as overrides of the defaults (tuning aids); nothing on the
    // launch path calls getenv.
    uavqp_default_settings(&ctx->settings);
    if (const char* e = std::getenv("UAVQP_TILE")) {
        const int t = std::atoi(e);
        if (t == 4 || t == 8 || t == 16 || t == 32) {
            (void)apply_variant(ctx, t);
            ctx->settings.kernel_variant = t;
        }
    }
    if (std::getenv("UAVQP_NO_LSORT")) ctx->settings.ragged_window_sort = 0;
    if (std::getenv("UAVQP_NO_WAVE_PRELUDE")) ctx->wave_prelude = 0;
    if (const char* e = std::getenv("UAVQP_DEAL_TICKETS")) ctx->deal_tickets = std::atoi(e) != 0;
    if (const char* e = std::getenv("UAVQP_DUAL_TRIPS_EXTRA")) ctx->dual_trips_extra = std::atoi(e);
    if (const char* e = std::getenv("UAVQP_WAVE_PRELUDE")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) ctx->wave_prelude = v; }
    if (const char* e = std::getenv("UAVQP_GENERIC_NAX")) ctx->settings.generic_lanes_per_traj = std::atoi(e) == 3 ? 1 : 3;
    if (const char* e = std::getenv("UAVQP_GENERIC_WPC")) {
        const int w = std::atoi(e);
        if (w > 0 && w <= 32) ctx->settings.generic_waves_per_cu = w;
    }
    *out_ctx = ctx;
    return UAVQP_OK;
}

extern "C" int uavqp_destroy(uavqp_ctx* ctx) {
    if (!ctx) return UAVQP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm.comm) (void)uavqp_comm_destroy(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->perm) (void)hipFree(ctx->perm);
    if (ctx->dummy) (void)hipFree(ctx->dummy);
    if (ctx->h_axis) (void)hipHostFree(ctx->h_axis);
    if (ctx->rows_warm) (void)hipFree(ctx->rows_warm);
    if (ctx->rows_warm2) (void)hipFree(ctx->rows_warm2);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->d_pipe) (void)hipFree(ctx->d_pipe);
    if (ctx->h_pipe) (void)hipHostFree(ctx->h_pipe);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return UAVQP_OK;
}

extern "C" int uavqp_set_stream(uavqp_ctx* ctx, void* hip_stream) {
    if (!ctx) return UAVQP_ERR_INVALID_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return UAVQP_OK;
}

extern "C" int uavqp_synchronize(uavqp_ctx* ctx) {
    if (!ctx) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    return UAVQP_OK;
}

static int apply_variant(uavqp_ctx* ctx, int variant) {
    if (variant == 4 || variant == 8 || variant == 16 || variant == 32) {  // specialised kernel with a fixed tile shape
        ctx->variant = 2;
        ctx->tile_override = variant;
        return UAVQP_OK;
    }
    if (variant < 0 || variant > 2) return UAVQP_ERR_INVALID_ARG;
    ctx->variant = variant;
    ctx->tile_override = 0;
    return UAVQP_OK;
}

extern "C" int uavqp_set_variant(uavqp_ctx* ctx, int variant) {
    if (!ctx) return UAVQP_ERR_INVALID_ARG;
    const int rc = apply_variant(ctx, variant);
    if (rc == UAVQP_OK) ctx->settings.kernel_variant = variant;
    return rc;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int ensure_ws(uavqp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return UAVQP_OK;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->ws) UAVQP_HIP(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    UAVQP_HIP(hipMalloc((void**)&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return UAVQP_OK;
}

extern "C" int uavqp_solve_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                        const int32_t* d_seg_offsets, const double* d_waypoints,
                                        const double* d_times, const double* d_bc, double* d_coeff_out,
                                        int32_t* d_status_out) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0) return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_waypoints || !d_times || !d_bc || !d_coeff_out) return UAVQP_ERR_INVALID_ARG;
    if (uniform_segments == 0 && (!d_seg_offsets || max_segments < 1)) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    const int Mmax = uniform_segments > 0 ? uniform_segments : max_segments;

    BatchArgs a;
    a.n_traj = n_traj;
    a.uniform = uniform_segments;
    a.max_segments = Mmax;
    a.seg_offsets = d_seg_offsets;
    a.waypoints = d_waypoints;
    a.times = d_times;
    a.bc = d_bc;
    a.coeff = d_coeff_out;
    a.status = d_status_out;
    a.dummy = ctx->dummy;
    a.fused_sort = 0;
    a.perm = nullptr;
    a.perm4 = nullptr;

    // the specialised kernels move 16 bytes per lane (LDS-DMA loads, dwordx4 stores): every array must be
    // 16-byte aligned (allocations are; a view starting at an odd double is not) -- otherwise the generic
    // kernel, which touches memory 8 bytes at a time, takes the batch
    const bool aligned16 = ((((uintptr_t)d_waypoints) | ((uintptr_t)d_times) | ((uintptr_t)d_bc) | ((uintptr_t)d_coeff_out)) & 15u) == 0;
    if (uniform_segments > 0 && ctx->variant != 1 && aligned16) {
        // Tile shape by batch size (measured on MI355X, 8-segment snap, us per launch over rotating buffers, tools/ubench/tw, round 3 --
        // after the pipelined copy-out of the latency shapes; tiles 4 / 8 / 16 / 32):
        //    1024: 4.00 / 4.44 / 7.4 / -      4096: 5.08 / 5.13 / 8.0 / -     5120: 6.62 / 5.14 / 8.4 / -     8192: 7.38 / 6.02 / 8.8 / -
        //   16384: 11.9 / 9.1 / 10.4 / 10.5   24576: 16.4 / 12.3 / 15.9 / 12.2   32768: - / 16.2 / 16.9 / 15.4   65536: - / 34.4 / 31.6 / 25.4
        // One CU moves only ~10 B/clk, so a small batch is spread over all CUs with fewer trajectories per wave (tile 4: 16 lanes per
        // trajectory, every SIMD busy up to 16 trajectories per CU; tile 8: 8 lanes); a large one wants the full-wave shape that does
        // the least redundant work.  Tile 16 no longer wins anywhere (it stays selectable through uavqp_set_variant).
        int tile = (n_traj <= 16 * ctx->num_cus) ? 4 : ((n_traj <= 96 * ctx->num_cus) ? 8 : 32);
        if (ctx->tile_override) tile = ctx->tile_override;
        uavqp::twisted_fn fn = uavqp::find_twisted(r, uniform_segments, tile);
        if (fn) {
            a.ws = nullptr;
            const int n_tiles = (n_traj + tile - 1) / tile;
            const int max_wg = ctx->num_cus * 4;  // 1 wave / SIMD (register-resident state), persistent over tiles
            int g = n_tiles < max_wg ? n_tiles : max_wg;
            hipLaunchKernelGGL(fn, dim3(g), dim3(64), 0, ctx->stream, a);
            UAVQP_HIP(hipGetLastError());
            return UAVQP_OK;
        }
        if (ctx->variant == 2) {
            g_last_error = "variant 2 requested but no specialised kernel for this (r, segments)";
            return UAVQP_ERR_INVALID_ARG;
        }
    }
    // ragged batches of some size: 256-thread workgroups that deal their chunk to the lanes by segment count
    // Lanes: one per (trajectory, axis) for small batches -- 21 trajectories per wave, a shorter dependent chain per lane
    // (measured, r = 4: M = 14 24 -> 18 us, M = 24 40 -> 29 us up to 4096 trajectories) -- and one per trajectory (3 axes
    // share the factorisation, a third of the E traffic) once the batch is throughput-bound (cross-over at ~8192;
    // 32768 x M = 14: 50 vs 76 us).  Ragged batches of some size deal windows of 16 waves' trajectories by segment count.
    const bool lsort = uniform_segments == 0 && n_traj >= 2048 && ctx->settings.ragged_window_sort;
    int nax = n_traj <= 32 * ctx->num_cus ? 1 : 3;
    if (ctx->settings.generic_lanes_per_traj == 1) nax = 3;       // one lane per trajectory carries all three axes
    else if (ctx->settings.generic_lanes_per_traj == 3) nax = 1;  // one lane per (trajectory, axis)
    // two lanes per trajectory (qp_generic2.h: twisted elimination with a runtime segment count): half the dependent chain
    // per lane and twice the waves
    // (inputs staged in LDS: 32 (4 Mmax + 3) doubles per wave, within the 64 KiB a workgroup gets without opting in)
    const size_t pair_lds = uavqp::generic2_lds_bytes(Mmax);
    // Measured (MI355X, us per launch, lanes per trajectory 1 / 2 / 3-axis-lanes): 32768 ragged M in [4, 24], r = 4: 86 / 54 / 125;
    // r = 3: 72 / 48 / 85; 4096 ragged: 45 / 30 / 36; 1000 ragged: 43 / 26 / 29; 32768 x M = 14: 50 / 41 / 76; x M = 24: 123 / 73 /
    // 178; 262144 x M = 14: 501 / 356 / 716; 32768 x M = 1: 5.4 / 8.5 / 5.9, M = 2: 9.6 / 9.3 / 10.9, M = 3: 13.1 / 13.1 / 15.4
    // (tools/generic2_probe.py) -- the pair kernel everywhere but for the shortest trajectories.
    const bool pair = pair_lds <= 64 * 1024 && aligned16 &&
                      (ctx->settings.generic_lanes_per_traj == 2 || (ctx->settings.generic_lanes_per_traj == 0 && Mmax >= 3));
    if (pair) nax = 3;
    const int ipw = pair ? 32 : (nax == 3 ? 64 : 21);
    const int block = 64;
    int grid = (n_traj + ipw - 1) / ipw;
    // Resident waves: the kernel is bound by the E/h workspace round trip, and the workspace is per resident lane.  With
    // one wave per CU it is ~30 MB (r = 4, M = 14) and stays in L2 while the grid strides over the batch (measured with
    // tools/generic_grid_probe.py: 65536 x M=14: 111 us at 1 wave/CU vs 141 us at 8; 262144: 524 vs 640 us; r = 3, M = 20,
    // 262144: 625 vs 920 us); up to 128 trajectories per CU two waves per CU run the batch in a single round (49 vs 58 us).
    int waves_per_cu = pair ? (int)std::min<size_t>(4, (160 * 1024) / pair_lds) : (nax == 3 ? (n_traj <= 128 * ctx->num_cus ? 2 : 1) : 12);
    if (ctx->settings.generic_waves_per_cu > 0) waves_per_cu = ctx->settings.generic_waves_per_cu;
    const int max_grid = ctx->num_cus * waves_per_cu;
    if (grid > max_grid) grid = max_grid;
    if (lsort) grid = (grid + 15) / 16 * 16;
    const int F = (r - 1) * (r - 1) + nax * (r - 1);
    const size_t ws_bytes = sizeof(double) * (size_t)(Mmax > 1 ? Mmax - 1 : 1) * F * (size_t)grid * block;
    int rc = ensure_ws(ctx, ws_bytes);
    if (rc != UAVQP_OK) return rc;
    a.ws = ctx->ws;
    a.perm = nullptr;
    a.perm4 = nullptr;
    // round 6: the pair kernel's waves rank their window themselves (qp_generic2.h) -- no window_sort_kernel launch in front of the solve, no
    // permutation in HBM; ragged_window_sort = 2 keeps the separate launch (A/B, cross-check test)
    a.fused_sort = (lsort && pair && Mmax <= 63 && ctx->settings.ragged_window_sort != 2) ? 1 : 0;
    if (lsort && !a.fused_sort) {
        // dealing permutation: one counting sort per window of 16 waves' trajectories (see window_sort_kernel)
        if ((size_t)n_traj > ctx->perm_count) {
            UAVQP_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->perm) UAVQP_HIP(hipFree(ctx->perm));
            ctx->perm = nullptr;
            ctx->perm_count = 0;
            // [n_traj] int32 + (16-byte aligned) [n_traj] int4
            UAVQP_HIP(hipMalloc((void**)&ctx->perm, align256(sizeof(int32_t) * (size_t)n_traj) + sizeof(int4) * (size_t)n_traj));
            ctx->perm_count = (size_t)n_traj;
        }
        int4* perm4 = pair ? (int4*)((char*)ctx->perm + align256(sizeof(int32_t) * ctx->perm_count)) : nullptr;
        const int win = 16 * ipw;
        const int n_win = (n_traj + win - 1) / win;
        if (pair) hipLaunchKernelGGL((uavqp::window_sort_kernel<512>), dim3(n_win), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, ctx->perm, perm4);
        else if (nax == 3) hipLaunchKernelGGL((uavqp::window_sort_kernel<1024>), dim3(n_win), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, ctx->perm, perm4);
        else hipLaunchKernelGGL((uavqp::window_sort_kernel<336>), dim3(n_win), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, ctx->perm, perm4);
        a.perm = ctx->perm;
        a.perm4 = perm4;
    }
#define UAVQP_GENERIC(RR)                                                                                                   \
    do {                                                                                                                    \
        if (pair) {                                                                                                         \
            if (lsort) hipLaunchKernelGGL((uavqp::solve_generic2_kernel<RR, true>), dim3(grid), dim3(block), pair_lds, ctx->stream, a);  \
            else hipLaunchKernelGGL((uavqp::solve_generic2_kernel<RR, false>), dim3(grid), dim3(block), pair_lds, ctx->stream, a);       \
        } else if (nax == 3) {                                                                                              \
            if (lsort) hipLaunchKernelGGL((uavqp::solve_generic_kernel<RR, true, 3>), dim3(grid), dim3(block), 0, ctx->stream, a);  \
            else hipLaunchKernelGGL((uavqp::solve_generic_kernel<RR, false, 3>), dim3(grid), dim3(block), 0, ctx->stream, a);       \
        } else {                                                                                                            \
            if (lsort) hipLaunchKernelGGL((uavqp::solve_generic_kernel<RR, true, 1>), dim3(grid), dim3(block), 0, ctx->stream, a);  \
            else hipLaunchKernelGGL((uavqp::solve_generic_kernel<RR, false, 1>), dim3(grid), dim3(block), 0, ctx->stream, a);       \
        }                                                                                                                   \
    } while (0)
    if (r == 3) UAVQP_GENERIC(3);
    else UAVQP_GENERIC(4);
#undef UAVQP_GENERIC
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

static int ensure_stage(uavqp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->stage_bytes) return UAVQP_OK;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage) UAVQP_HIP(hipFree(ctx->d_stage));
    ctx->d_stage = nullptr;
    ctx->stage_bytes = 0;
    UAVQP_HIP(hipMalloc(&ctx->d_stage, bytes));
    ctx->stage_bytes = bytes;
    return UAVQP_OK;
}


// the pinned, device-mapped staging page of the latency paths (single-axis entry point, small host batches)
static constexpr size_t MAPPED_HEAD = 256;   // head of the mapped page: the completion word of the polled host entries
static int ensure_mapped(uavqp_ctx* ctx, size_t need) {
    if (need <= ctx->axis_bytes) return UAVQP_OK;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_axis) UAVQP_HIP(hipHostFree(ctx->h_axis));
    ctx->h_axis = ctx->d_axis = nullptr;
    ctx->axis_bytes = 0;
    const size_t cap = need < 65536 ? 65536 : need * 2;
    UAVQP_HIP(hipHostMalloc(&ctx->h_axis, cap, hipHostMallocMapped | hipHostMallocCoherent));
    UAVQP_HIP(hipHostGetDevicePointer(&ctx->d_axis, ctx->h_axis, 0));
    ctx->axis_bytes = cap;
    return UAVQP_OK;
}

extern "C" int uavqp_solve_batch_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                      const int32_t* seg_offsets, const double* waypoints, const double* times,
                                      const double* bc, double* coeff_out, int32_t* status_out) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0) return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!waypoints || !times || !bc || !coeff_out) return UAVQP_ERR_INVALID_ARG;
    if (uniform_segments == 0 && !seg_offsets) return UAVQP_ERR_INVALID_ARG;
    long long total_seg = 0;
    int Mmax = uniform_segments;
    if (uniform_segments > 0) {
        total_seg = (long long)uniform_segments * n_traj;
    } else {
        if (seg_offsets[0] != 0) return UAVQP_ERR_INVALID_ARG;
        for (int b = 0; b < n_traj; ++b) {
            const int M = seg_offsets[b + 1] - seg_offsets[b];
            if (M < 0) return UAVQP_ERR_INVALID_ARG;
            if (M > Mmax) Mmax = M;
        }
        total_seg = seg_offsets[n_traj];
        if (max_segments > 0 && max_segments < Mmax) Mmax = max_segments;  // larger ones are flagged invalid
        if (Mmax < 1) Mmax = 1;
    }
    UAVQP_HIP(hipSetDevice(ctx->device));
    const size_t b_off = uniform_segments > 0 ? 0 : align256(sizeof(int32_t) * (size_t)(n_traj + 1));
    const size_t b_wp = align256(sizeof(double) * 3 * (size_t)(total_seg + n_traj));
    const size_t b_t = align256(sizeof(double) * (size_t)total_seg);
    const size_t b_bc = align256(sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3);
    const size_t b_out = align256(sizeof(double) * 3 * 2 * r * (size_t)total_seg);
    const size_t b_st = align256(sizeof(int32_t) * (size_t)n_traj);
    // Small batches (a planner's single trajectory, a handful of candidates) are a latency path: the whole batch lives in the
    // pinned page that is mapped into the device, the kernels read and write it over the host link, and the call is the
    // launch(es) plus one stream synchronisation instead of 3-4 H2D copies, a memset and 2 D2H copies (measured, one 7-segment
    // trajectory: 72 -> 3x us per call).
    const size_t b_all = b_off + b_wp + b_t + b_bc + b_out + b_st;
    if (b_all <= 256 * 1024) {
        int rcm = ensure_mapped(ctx, b_all + MAPPED_HEAD);
        if (rcm != UAVQP_OK) return rcm;
        char* hb = (char*)ctx->h_axis + MAPPED_HEAD;   // (the page's first 256 bytes hold the completion word: a FIXED slot no payload ever aliases)
        char* db = (char*)ctx->d_axis + MAPPED_HEAD;
        if (uniform_segments == 0) std::memcpy(hb, seg_offsets, sizeof(int32_t) * (size_t)(n_traj + 1));
        std::memcpy(hb + b_off, waypoints, sizeof(double) * 3 * (size_t)(total_seg + n_traj));
        std::memcpy(hb + b_off + b_wp, times, sizeof(double) * (size_t)total_seg);
        std::memcpy(hb + b_off + b_wp + b_t, bc, sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3);
        std::memset(hb + b_off + b_wp + b_t + b_bc, 0, sizeof(double) * 3 * 2 * r * (size_t)total_seg);   // failed trajectories come back as zeros
        std::memset(hb + b_off + b_wp + b_t + b_bc + b_out, 0, sizeof(int32_t) * (size_t)n_traj);
        rcm = uavqp_solve_batch_device(ctx, r, n_traj, uniform_segments, Mmax, uniform_segments > 0 ? nullptr : (const int32_t*)db,
                                       (const double*)(db + b_off), (const double*)(db + b_off + b_wp), (const double*)(db + b_off + b_wp + b_t),
                                       (double*)(db + b_off + b_wp + b_t + b_bc), (int32_t*)(db + b_off + b_wp + b_t + b_bc + b_out));
        if (rcm != UAVQP_OK) return rcm;
        {   // (no stream synchronisation: see uavqp_solve_axis_host)
            const unsigned int seq = ++ctx->pipe_seq;
            volatile unsigned long long* h_word = (volatile unsigned long long*)ctx->h_axis;
            *h_word = 0ull;
            std::atomic_thread_fence(std::memory_order_release);
            hipLaunchKernelGGL(uavqp::host_word_kernel, dim3(1), dim3(1), 0, ctx->stream, (const int32_t*)nullptr, (volatile unsigned long long*)ctx->d_axis, seq);
            rcm = await_host_word(ctx->stream, h_word, seq, nullptr, "uavqp_solve_batch_host");
            if (rcm != UAVQP_OK) return rcm;
        }
        std::memcpy(coeff_out, hb + b_off + b_wp + b_t + b_bc, sizeof(double) * 3 * 2 * r * (size_t)total_seg);
        if (status_out) std::memcpy(status_out, hb + b_off + b_wp + b_t + b_bc + b_out, sizeof(int32_t) * (size_t)n_traj);
        return UAVQP_OK;
    }
    int rc = ensure_stage(ctx, b_all);
    if (rc != UAVQP_OK) return rc;
    char* base = (char*)ctx->d_stage;
    int32_t* d_off = uniform_segments > 0 ? nullptr : (int32_t*)base;
    double* d_wp = (double*)(base + b_off);
    double* d_t = (double*)(base + b_off + b_wp);
    double* d_bc = (double*)(base + b_off + b_wp + b_t);
    double* d_out = (double*)(base + b_off + b_wp + b_t + b_bc);
    int32_t* d_st = (int32_t*)(base + b_off + b_wp + b_t + b_bc + b_out);
    hipStream_t s = ctx->stream;
    if (d_off) UAVQP_HIP(hipMemcpyAsync(d_off, seg_offsets, sizeof(int32_t) * (size_t)(n_traj + 1), hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_wp, waypoints, sizeof(double) * 3 * (size_t)(total_seg + n_traj), hipMemcpyHostToDevice, s));
    if (total_seg > 0) UAVQP_HIP(hipMemcpyAsync(d_t, times, sizeof(double) * (size_t)total_seg, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_bc, bc, sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3, hipMemcpyHostToDevice, s));
    // the kernels leave failed trajectories unwritten and the staging buffer is reused: clear it, so that a failed
    // trajectory comes back as zeros (never as another batch's coefficients)
    if (total_seg > 0) UAVQP_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * 3 * 2 * r * (size_t)total_seg, s));
    rc = uavqp_solve_batch_device(ctx, r, n_traj, uniform_segments, Mmax, d_off, d_wp, d_t, d_bc, d_out, d_st);
    if (rc != UAVQP_OK) return rc;
    if (total_seg > 0) UAVQP_HIP(hipMemcpyAsync(coeff_out, d_out, sizeof(double) * 3 * 2 * r * (size_t)total_seg, hipMemcpyDeviceToHost, s));
    if (status_out) UAVQP_HIP(hipMemcpyAsync(status_out, d_st, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    UAVQP_HIP(hipStreamSynchronize(s));
    return UAVQP_OK;
}

extern "C" int uavqp_solve_axis_host(uavqp_ctx* ctx, int r, int n_seg, const double* pos_1d, const double* bound_vel,
                                     const double* bound_acc, const double* bound_jerk, const double* time_vec,
                                     double* coef_1d, int32_t* status_out) {
    if (!ctx || (r != 3 && r != 4) || n_seg < 1 || !pos_1d || !bound_vel || !bound_acc || !time_vec || !coef_1d)
        return UAVQP_ERR_INVALID_ARG;
    // One axis of the reference call = a 1-trajectory batch whose other two axes are zero.  The reference's callers solve x, y, z
    // one after the other (test_minimum_jerk.cpp:75,100,125; traj_optimizer.cpp), so this call is a latency path: the batch
    // lives in ONE pinned page that is mapped into the device -- the host writes positions / durations / boundary values
    // there, the kernel reads them over the host link and writes the coefficients and the status next to them, and the call is
    // one kernel launch and one stream synchronisation (the copying entry point underneath it: 3 H2D + memset + 2 D2H).
    const int nd = r - 1, nc = 2 * r;
    const size_t o_wp = 0;
    const size_t o_t = align256(sizeof(double) * 3 * (size_t)(n_seg + 1));
    const size_t o_bc = o_t + align256(sizeof(double) * (size_t)n_seg);
    const size_t o_out = o_bc + align256(sizeof(double) * 2 * 3 * 3);
    const size_t o_st = o_out + align256(sizeof(double) * 3 * (size_t)nc * n_seg);
    const size_t need = o_st + 256 + MAPPED_HEAD;
    UAVQP_HIP(hipSetDevice(ctx->device));
    {
        const int rcm = ensure_mapped(ctx, need);
        if (rcm != UAVQP_OK) return rcm;
    }
    char* hb = (char*)ctx->h_axis + MAPPED_HEAD;   // (completion word: the fixed slot at the head of the page)
    char* db = (char*)ctx->d_axis + MAPPED_HEAD;
    double* wp = (double*)(hb + o_wp);
    for (int i = 0; i <= n_seg; ++i) {
        wp[3 * i] = pos_1d[i];
        wp[3 * i + 1] = 0.0;
        wp[3 * i + 2] = 0.0;
    }
    std::memcpy(hb + o_t, time_vec, sizeof(double) * (size_t)n_seg);
    double* bc = (double*)(hb + o_bc);
    for (int k = 0; k < 2 * 3 * 3; ++k) bc[k] = 0.0;
    for (int e = 0; e < 2; ++e) {
        bc[(e * nd + 0) * 3] = bound_vel[e];
        bc[(e * nd + 1) * 3] = bound_acc[e];
        if (r == 4) bc[(e * nd + 2) * 3] = bound_jerk ? bound_jerk[e] : 0.0;
    }
    volatile int32_t* st = (volatile int32_t*)(hb + o_st);
    *st = 0;
    int rc = uavqp_solve_batch_device(ctx, r, 1, n_seg, n_seg, nullptr, (const double*)(db + o_wp), (const double*)(db + o_t),
                                      (const double*)(db + o_bc), (double*)(db + o_out), (int32_t*)(db + o_st));
    if (rc != UAVQP_OK) return rc;
    // (no stream synchronisation: a one-thread kernel behind the solve stamps a word of the page, the host polls it)
    {
        const unsigned int seq = ++ctx->pipe_seq;
        volatile unsigned long long* h_word = (volatile unsigned long long*)ctx->h_axis;
        *h_word = 0ull;
        std::atomic_thread_fence(std::memory_order_release);
        hipLaunchKernelGGL(uavqp::host_word_kernel, dim3(1), dim3(1), 0, ctx->stream, (const int32_t*)nullptr, (volatile unsigned long long*)ctx->d_axis, seq);
        rc = await_host_word(ctx->stream, h_word, seq, nullptr, "uavqp_solve_axis_host");
        if (rc != UAVQP_OK) return rc;
    }
    const int32_t status = *st;
    if (status == UAVQP_SOLVED) std::memcpy(coef_1d, hb + o_out, sizeof(double) * (size_t)nc * n_seg);   // x axis = the first M rows
    if (status_out) *status_out = status;
    return rc;
}

extern "C" int uavqp_capture_begin(uavqp_ctx* ctx) {
    if (!ctx) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    UAVQP_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return UAVQP_OK;
}

extern "C" int uavqp_capture_end(uavqp_ctx* ctx, void** out_graph_exec) {
    if (!ctx || !out_graph_exec) return UAVQP_ERR_INVALID_ARG;
    *out_graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    UAVQP_HIP(hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        g_last_error = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
        return UAVQP_ERR_HIP;
    }
    *out_graph_exec = (void*)exec;
    return UAVQP_OK;
}

extern "C" int uavqp_graph_launch(uavqp_ctx* ctx, void* graph_exec) {
    if (!ctx || !graph_exec) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, ctx->stream));
    return UAVQP_OK;
}

extern "C" int uavqp_graph_destroy(uavqp_ctx* ctx, void* graph_exec) {
    if (!ctx) return UAVQP_ERR_INVALID_ARG;
    if (graph_exec) UAVQP_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return UAVQP_OK;
}

extern "C" int uavqp_eval_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                       const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                       int what, double* d_out) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || n_samples < 0 || uniform_segments < 0 || (what & 7) == 0)
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0 || n_samples == 0) return UAVQP_OK;
    if (!d_times || !d_coeff || !d_out || (uniform_segments == 0 && !d_seg_offsets)) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    uavqp::EvalArgs a;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.n_samples = n_samples; a.what = what & 7;
    a.seg_offsets = d_seg_offsets; a.times = d_times; a.coeff = d_coeff; a.t0 = t0; a.dt = dt; a.out = d_out;
    const long long total = (long long)n_traj * n_samples;
    long long grid = (total + 255) / 256;
    const long long cap = (long long)ctx->num_cus * 16;
    if (grid > cap) grid = cap;
    if (r == 3)
        hipLaunchKernelGGL(uavqp::eval_kernel<3>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(uavqp::eval_kernel<4>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

extern "C" int uavqp_traj_length_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                        const double* d_times, const double* d_coeff, double dt, double* d_length, double* d_mean_vel,
                                        int32_t* d_n_samples) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0 || !(dt > 0.0) || !(dt < INFINITY)) return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_times || !d_coeff || (uniform_segments == 0 && !d_seg_offsets)) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    uavqp::LengthArgs a;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.seg_offsets = d_seg_offsets; a.times = d_times; a.coeff = d_coeff; a.dt = dt;
    a.length = d_length; a.mean_vel = d_mean_vel; a.n_samples = d_n_samples;
    int grid = n_traj < ctx->num_cus * 32 ? n_traj : ctx->num_cus * 32;
    if (r == 3)
        hipLaunchKernelGGL(uavqp::traj_length_kernel<3>, dim3(grid), dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(uavqp::traj_length_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

// Dealing order of a ragged batch (longest trajectory first; seg_hist / seg_scan / seg_scatter_kernel): order [n_traj] at `storage`,
// histogram and cursors in its last 2048 bytes (storage = align256(4 n_traj) + 2048 bytes).  The order depends on the segment counts
// only: a caller that re-solves the same batch (the config-5 pipeline) makes it once and passes it to corridor_warm_impl.
static size_t length_order_bytes(int n_traj) { return align256(sizeof(int32_t) * (size_t)n_traj) + 2048; }
static int make_length_order(uavqp_ctx* ctx, const int32_t* d_seg_offsets, int n_traj, char* storage, const int32_t** d_order_out) {
    int32_t* d_order = (int32_t*)storage;
    int* d_hist = (int*)(storage + length_order_bytes(n_traj) - 2048);
    int* d_cursor = d_hist + 256;
    UAVQP_HIP(hipMemsetAsync(d_hist, 0, 256 * sizeof(int), ctx->stream));
    int sg = (n_traj + 255) / 256;
    if (sg > ctx->num_cus * 4) sg = ctx->num_cus * 4;
    hipLaunchKernelGGL(uavqp::seg_hist_kernel, dim3(sg), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, d_hist);
    hipLaunchKernelGGL(uavqp::seg_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_hist, d_cursor);
    hipLaunchKernelGGL(uavqp::seg_scatter_kernel, dim3(sg), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, d_cursor, d_order);
    *d_order_out = d_order;
    return UAVQP_OK;
}

// total_segments < 0: unknown -- for a ragged batch the last CSR offset is then read back from the device (4 bytes, one stream
// synchronisation); callers that know it (the host-pointer entries, the rows solver's second phase, the corridor pipeline) pass it
// and the call stays asynchronous.
static int corridor_warm_impl(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                              const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                              const double* d_bc, const double* d_corr_lo, const double* d_corr_hi,
                              double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out,
                              uint64_t* d_active_set, int warm_start, long long total_segments, const int32_t* d_order_ready = nullptr,
                              const int32_t* d_only_i32 = nullptr, const unsigned char* d_only_u8 = nullptr,
                              double* d_gcache = nullptr, const double* d_gscale = nullptr, int gcache_mode = 0,
                              const int32_t* d_compact_ready = nullptr, const int* d_n_active_ready = nullptr) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0) return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_waypoints || !d_times || !d_bc || !d_corr_lo || !d_corr_hi || !d_coeff_out || !d_status_out) return UAVQP_ERR_INVALID_ARG;
    if (uniform_segments == 0 && (!d_seg_offsets || max_segments < 1)) return UAVQP_ERR_INVALID_ARG;
    if (warm_start && !d_active_set) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    const int Mmax = uniform_segments > 0 ? uniform_segments : max_segments;
    uavqp::CorridorArgs a;
    a.active = (unsigned long long*)d_active_set; a.warm = warm_start == 2 ? 2 : (warm_start ? 1 : 0);
    a.n_traj = n_traj; a.uniform = uniform_segments; a.max_segments = Mmax; a.max_iter = ctx->settings.max_iter > 0 ? ctx->settings.max_iter : 8 * Mmax + 20;
    // cold start: 0 = empty set, 1 = closed-form set of the prep kernel, 2 = the set the position-space dual method ends with
    // (qp_corridor_dual.h; trajectories of up to 33 segments, longer batches fall back to 1) -- then no block-pivot rounds: the set
    // only has to be verified
    const int gmode = warm_start ? 0 : ctx->settings.corridor_initial_guess;
    const bool dual = gmode == 2 && Mmax >= 2 && Mmax - 1 <= 32;
    a.pdas_rounds = warm_start ? ctx->settings.corridor_pdas_rounds_warm : (dual ? 0 : ctx->settings.corridor_pdas_rounds);
    a.guess_closed_form = (gmode != 0 && !dual) ? 1 : 0;
    a.fused_emit = 1;      // the solve kernel writes the polynomials itself (corridor_emit_kernel: only behind the rows solvers)
    a.only_i32 = d_only_i32; a.only_u8 = d_only_u8;
    a.gcache = d_gcache; a.gscale = d_gscale; a.gcache_mode = (d_gcache && Mmax - 1 <= 24) ? gcache_mode : 0;
    a.seg_offsets = d_seg_offsets; a.waypoints = d_waypoints; a.times = d_times; a.bc = d_bc;
    a.corr_lo = d_corr_lo; a.corr_hi = d_corr_hi; a.coeff = d_coeff_out; a.status = d_status_out; a.iters = d_iters_out;
    // Persistent single-wave workgroups, one per SIMD (the sweep state of a lane pair lives in LDS: 4 x 40 KiB per CU),
    // 32 problems in flight per wave, refilled from the work counter.
    const long long pairs = 3LL * n_traj;
    long long grid = (pairs + 31) / 32;
    const int own_max_ = ((uniform_segments > 0 ? uniform_segments : max_segments) + 1) / 2;
    // Small batches of long snap problems are bound by the slowest problem's chain of iterations, not by throughput: two waves per CU
    // with ten instead of five own knots per lane on chip make every iteration of a 24-segment problem ~13 % shorter (config 5: five
    // rounds 5.85 -> 5.64 ms) where four waves per CU would not be kept busy anyway; large batches (config 3) lose 50 % that way.
    const bool tail_shape = r == 4 && UAVQP_CORRIDOR_WAVES_PER_CU == 4 && own_max_ > uavqp::corridor_lds_knots(4) && pairs <= 4LL * ctx->num_cus * 2 * 32 &&
                            ctx->settings.corridor_tail_shape != 0;
    const int wpc = tail_shape ? 2 : uavqp::corridor_waves_per_cu();
    const long long max_grid = (long long)ctx->num_cus * wpc;
    if (grid > max_grid) grid = max_grid;
    const int NT = uavqp::corridor_lds_knots(r, wpc);
    const int F = r * (r + 1) / 2 + r + 1;  // must match corridor_solve_kernel's state layout
    const int own_max = (Mmax + 1) / 2;        // own knots of the longer half, meeting knot included
    const int ws_knots = own_max > NT ? own_max - NT : 0;
    long long rows = 0;                        // waypoint rows of the batch
    if (uniform_segments > 0) rows = (long long)n_traj * (uniform_segments + 1);
    else if (total_segments >= 0) rows = total_segments + n_traj;
    else {
        // sum(M_b) + n_traj: the last CSR offset is only known on the device; a 4-byte read-back (synchronous on the ctx stream)
        int32_t last = 0;
        UAVQP_HIP(hipMemcpyAsync(&last, d_seg_offsets + n_traj, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        UAVQP_HIP(hipStreamSynchronize(ctx->stream));
        if (last < 0) return UAVQP_ERR_INVALID_ARG;
        rows = (long long)last + n_traj;
    }
    const size_t b_xsol = align256(sizeof(double) * 3 * (size_t)r * (size_t)rows);
    const size_t b_queue = 256;
    const size_t b_desc = align256(sizeof(unsigned long long) * 3 * (size_t)n_traj);
    // (a caller that hands over a compacted dealing order -- the rows solve's box phase, the pipeline -- has decided the order: no window sort nobody reads)
    const bool deal_by_length = uniform_segments == 0 && n_traj >= 64 && ctx->settings.ragged_window_sort && !d_order_ready && !(d_compact_ready && d_n_active_ready);
    const size_t b_order = deal_by_length ? length_order_bytes(n_traj) : 0;   // order + histogram + cursors
    const size_t b_state = sizeof(double) * (size_t)ws_knots * F * (size_t)grid * 64;
    const bool guess = gmode != 0;     // cold start from a starting set (closed form: prep kernel; dual method: corridor_dual_kernel)
    const size_t b_guess = guess ? align256(sizeof(unsigned long long) * 6 * (size_t)n_traj) : 0;
    const bool masked = d_only_i32 || d_only_u8;
    // compacted dealing order + its length -- reserved whether this solve is masked or not (4 bytes per trajectory): the first masked solve of a
    // pipeline call must not grow the workspace, i.e. stop the stream and re-allocate, in the middle of the loop
    const size_t b_compact = align256(sizeof(int32_t) * (size_t)n_traj) + 256;
    // one lane per trajectory (qp_corridor_lane.h): batches of at most 16 segments per trajectory, G not cached across solves
#ifdef UAVQP_EXPERIMENTS
    const bool lane_prelude = dual && !(d_gcache && gcache_mode != 0) && Mmax - 1 <= uavqp::LANE_NV && ctx->settings.corridor_prelude_lanes == 1;
    long long lgrid = ((long long)n_traj + 63) / 64;
    if (lgrid > (long long)ctx->num_cus * 4) lgrid = (long long)ctx->num_cus * 4;     // (one wave per SIMD: the register file, not the 27 KB of LDS per wave, sets it)
    const size_t b_lane = lane_prelude ? align256(sizeof(double) * (size_t)uavqp::lane_scratch_doubles(r) * (size_t)lgrid) : 0;
#else
    const size_t b_lane = 0;
#endif
    int rc = ensure_ws(ctx, b_xsol + b_queue + b_desc + b_order + b_state + b_guess + b_compact + b_lane);
    if (rc != UAVQP_OK) return rc;
    a.guess = guess ? (unsigned long long*)((char*)ctx->ws + b_xsol + b_queue + b_desc + b_order + b_state) : nullptr;
    a.coeff = d_coeff_out;
    a.xsol = ctx->ws;
    a.queue = (unsigned int*)((char*)ctx->ws + b_xsol);
    ctx->dbg_queue = a.queue;
    a.desc = (unsigned long long*)((char*)ctx->ws + b_xsol + b_queue);
    a.ws = ws_knots > 0 ? (double*)((char*)ctx->ws + b_xsol + b_queue + b_desc + b_order) : nullptr;
    a.ws_knots = ws_knots;
    a.order = d_order_ready;
    if (deal_by_length) {
        rc = make_length_order(ctx, d_seg_offsets, n_traj, (char*)ctx->ws + b_xsol + b_queue + b_desc, &a.order);
        if (rc != UAVQP_OK) return rc;
    }
    a.n_active = nullptr;
    if (masked && d_compact_ready && d_n_active_ready) {
        // (the caller compacted the dealing order of the participating trajectories already: the pipeline does it right behind the
        // re-allocation whose flags are the mask, where the count doubles as "how many did it stretch")
        a.order = d_compact_ready;
        a.n_active = d_n_active_ready;
    } else if (masked) {
        int32_t* d_compact = (int32_t*)((char*)ctx->ws + b_xsol + b_queue + b_desc + b_order + b_state + b_guess);
        int* d_n_active = (int*)((char*)d_compact + align256(sizeof(int32_t) * (size_t)n_traj));
        hipLaunchKernelGGL(uavqp::compact_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, a.order, n_traj, d_only_i32, d_only_u8, d_compact, d_n_active);
        a.order = d_compact;
        a.n_active = d_n_active;
    }
#ifdef UAVQP_DUAL_DEBUG
    {
        static double* s_dbg = nullptr;
        if (!s_dbg) { UAVQP_HIP(hipMalloc(&s_dbg, 64 * 2048 * sizeof(double))); }
        if (gcache_mode != 2) UAVQP_HIP(hipMemsetAsync(s_dbg, 0, 64 * 2048 * sizeof(double), ctx->stream));   // (re-solves of a pipeline call keep what the wave prelude recorded)
        a.dbg = s_dbg;
        ctx->dbg_dual = s_dbg;
        ctx->dbg_guess = a.guess;
    }
#endif
#ifdef UAVQP_CORRIDOR_TIMING
    a.stamps = (long long*)((char*)a.queue + 64);   // debug build: section cycles of wave 0, read back by uavqp_debug_corridor_stamps
#endif
    if (!dual) UAVQP_HIP(hipMemsetAsync(a.queue, 0, sizeof(unsigned int), ctx->stream));     // (with a prelude: its first block resets the work counter)
    // with the dual prelude in front, resetting, validating and describing the problems is its business (one launch less, and the inputs are
    // not read a third time); otherwise corridor_reset_kernel + corridor_prep_kernel
    a.prep_in_dual = dual ? 1 : 0;
    if (!a.prep_in_dual) {
        hipLaunchKernelGGL(uavqp::corridor_reset_kernel, dim3((n_traj + 255) / 256), dim3(256), 0, ctx->stream, d_status_out, d_iters_out, n_traj, d_only_i32, d_only_u8);
        long long pgrid = (pairs + 255) / 256;
        if (pgrid > (long long)ctx->num_cus * 16) pgrid = (long long)ctx->num_cus * 16;
        if (r == 3) hipLaunchKernelGGL(uavqp::corridor_prep_kernel<3>, dim3((unsigned)pgrid), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(uavqp::corridor_prep_kernel<4>, dim3((unsigned)pgrid), dim3(256), 0, ctx->stream, a);
    }
    if (dual && gcache_mode == 2 && d_gcache && d_gscale && ctx->wave_prelude) {
        // G of every trajectory is in the cache of an earlier solve of this outer loop: one trajectory per wave, no chain, no lockstep
        if (ctx->wave_prelude == 2) {   // two trajectories per wave (lanes 0-31 / 32-63)
            long long wgrid = (long long)n_traj < (long long)ctx->num_cus * 8 ? (long long)n_traj : (long long)ctx->num_cus * 8;
            if (wgrid < 1) wgrid = 1;
            if (r == 3) hipLaunchKernelGGL((uavqp::corridor_dual_wave2_kernel<3>), dim3((unsigned)wgrid), dim3(64), 0, ctx->stream, a, ctx->dual_trips_extra);
            else hipLaunchKernelGGL((uavqp::corridor_dual_wave2_kernel<4>), dim3((unsigned)wgrid), dim3(64), 0, ctx->stream, a, ctx->dual_trips_extra);
        } else {
            long long wgrid = (long long)n_traj < (long long)ctx->num_cus * 12 ? (long long)n_traj : (long long)ctx->num_cus * 12;
            if (wgrid < 1) wgrid = 1;
            if (r == 3) hipLaunchKernelGGL((uavqp::corridor_dual_wave_kernel<3>), dim3((unsigned)wgrid), dim3(64), 0, ctx->stream, a, ctx->dual_trips_extra);
            else hipLaunchKernelGGL((uavqp::corridor_dual_wave_kernel<4>), dim3((unsigned)wgrid), dim3(64), 0, ctx->stream, a, ctx->dual_trips_extra);
        }
#ifdef UAVQP_EXPERIMENTS
    } else if (lane_prelude) {
        double* const d_lane = (double*)((char*)ctx->ws + b_xsol + b_queue + b_desc + b_order + b_state + b_guess + b_compact);
        if (r == 3) hipLaunchKernelGGL((uavqp::corridor_dual_lane_kernel<3>), dim3((unsigned)lgrid), dim3(64), 0, ctx->stream, a, d_lane, ctx->dual_trips_extra);
        else hipLaunchKernelGGL((uavqp::corridor_dual_lane_kernel<4>), dim3((unsigned)lgrid), dim3(64), 0, ctx->stream, a, d_lane, ctx->dual_trips_extra);
#endif
    } else if (dual) {
        // groups of 8 lanes (two tableau columns each) for the trajectories of up to 17 segments, whole DPP rows for the longer ones: a batch
        // of mixed lengths gets both launches, each skipping (per wave: the dealing order is by length) what the other one takes
        auto launch_dual = [&](int L, int NRW, int n_lo, int last) {
            const long long nb = ((long long)n_traj + 64 / L - 1) / (64 / L);
            const int lds_b = 8 * (64 / L) * uavqp::corridor_dual_lds_doubles(r, L, NRW);
            int wpc_d = (160 * 1024) / lds_b;
            const int wmax = NRW <= 24 ? 8 : 4;                   // (registers: two waves per SIMD up to 24 tableau rows, one beyond)
            if (wpc_d > wmax) wpc_d = wmax;
            if (wpc_d < 1) wpc_d = 1;
            const long long dgrid = nb < (long long)ctx->num_cus * wpc_d ? nb : (long long)ctx->num_cus * wpc_d;
#define UAVQP_DUAL_LAUNCH(R_, L_, N_) hipLaunchKernelGGL((uavqp::corridor_dual_kernel<R_, L_, N_>), dim3((unsigned)dgrid), dim3(64), 0, ctx->stream, a, n_lo, 0, last)
            if (r == 3) { if (NRW == 16) UAVQP_DUAL_LAUNCH(3, 8, 16); else if (NRW == 24) UAVQP_DUAL_LAUNCH(3, 16, 24); else UAVQP_DUAL_LAUNCH(3, 16, 32); }
            else { if (NRW == 16) UAVQP_DUAL_LAUNCH(4, 8, 16); else if (NRW == 24) UAVQP_DUAL_LAUNCH(4, 16, 24); else UAVQP_DUAL_LAUNCH(4, 16, 32); }
#undef UAVQP_DUAL_LAUNCH
        };
        const int nvar = Mmax - 1;
        const bool mixed = uniform_segments == 0;
        if (mixed && nvar > 16 && nvar <= 24) {
            // both shapes in one launch (corridor_dual_mixed_kernel): blocks [0, split) the 8-lane groups, the rest whole rows
            const int lds_b = 8 * uavqp::corridor_dual_mixed_lds(r);
            int wpc_d = (160 * 1024) / lds_b;
            if (wpc_d > 8) wpc_d = 8;
            const long long cap = (long long)ctx->num_cus * wpc_d;
            long long g8 = ((long long)n_traj + 7) / 8, g16 = ((long long)n_traj + 3) / 4;
            if (g8 > cap / 2) g8 = cap / 2;
            if (g16 > cap - g8) g16 = cap - g8;
            if (r == 3) hipLaunchKernelGGL((uavqp::corridor_dual_mixed_kernel<3>), dim3((unsigned)(g8 + g16)), dim3(64), 0, ctx->stream, a, (int)g8, 0);
            else hipLaunchKernelGGL((uavqp::corridor_dual_mixed_kernel<4>), dim3((unsigned)(g8 + g16)), dim3(64), 0, ctx->stream, a, (int)g8, 0);
        } else {
            // (`last`: the launch that also takes -- as invalid -- whatever is longer than any tableau)
            if (nvar <= 16 || mixed) launch_dual(8, 16, 1, nvar <= 16 ? 1 : 0);
            if (nvar > 16) launch_dual(16, nvar <= 24 ? 24 : 32, mixed ? 17 : 1, 1);
        }
    }
    if (r == 3) {
        if (ws_knots > 0) hipLaunchKernelGGL((uavqp::corridor_solve_kernel<3, true>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL((uavqp::corridor_solve_kernel<3, false>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
    } else {
        if (tail_shape) {
            if (ws_knots > 0) hipLaunchKernelGGL((uavqp::corridor_solve_kernel<4, true, 2>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL((uavqp::corridor_solve_kernel<4, false, 2>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
        } else if (ws_knots > 0) hipLaunchKernelGGL((uavqp::corridor_solve_kernel<4, true>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL((uavqp::corridor_solve_kernel<4, false>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
    }
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

extern "C" int uavqp_solve_corridor_warm_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                                const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                                                const double* d_bc, const double* d_corr_lo, const double* d_corr_hi,
                                                double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out,
                                                uint64_t* d_active_set, int warm_start) {
    return corridor_warm_impl(ctx, r, n_traj, uniform_segments, max_segments, d_seg_offsets, d_waypoints, d_times, d_bc, d_corr_lo, d_corr_hi,
                              d_coeff_out, d_status_out, d_iters_out, d_active_set, warm_start, -1);
}

#ifdef UAVQP_DUAL_DEBUG
// debug build only (tools/corridor_dual_gpu_probe.py): what corridor_dual_kernel computed for the first 64 trajectories of the last
// cold corridor solve, and the starting sets [n_traj][3][2] it handed to the solve kernel
extern "C" int uavqp_debug_corridor_dual(uavqp_ctx* ctx, double* out_64x2048, unsigned long long* out_guess, int n_traj) {
    if (!ctx || !ctx->dbg_dual) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    UAVQP_HIP(hipMemcpy(out_64x2048, ctx->dbg_dual, 64 * 2048 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_guess && ctx->dbg_guess) UAVQP_HIP(hipMemcpy(out_guess, ctx->dbg_guess, sizeof(unsigned long long) * 6 * (size_t)n_traj, hipMemcpyDeviceToHost));
    return UAVQP_OK;
}
#endif

#ifdef UAVQP_CORRIDOR_TIMING
// debug build only (tools/corridor_sections.py): cycles wave 0 of the last corridor solve spent per section
extern "C" int uavqp_debug_corridor_stamps(uavqp_ctx* ctx, long long* out7) {
    if (!ctx || !out7 || !ctx->ws) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    UAVQP_HIP(hipMemcpy(out7, (char*)ctx->dbg_queue + 64, 7 * sizeof(long long), hipMemcpyDeviceToHost));
    return UAVQP_OK;
}
#endif

#ifdef UAVQP_CLOUD_STATS
// probe build only (tools/cloud_phase_probe.py): blocks of the last cloud_grid2d_kernel launch that scanned ring 0 / 1 / 2
extern "C" int uavqp_debug_cloud_phases(uavqp_ctx* ctx, unsigned int* out4) {
    if (!ctx || !out4 || !ctx->dbg_queue) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    UAVQP_HIP(hipMemcpy(out4, ctx->dbg_queue, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return UAVQP_OK;
}
#endif

#ifdef UAVQP_LANE_TIMING
// probe build only (tools/lane_sections.py): cycles block 0 of the last corridor_dual_lane_kernel launch spent per section
extern "C" int uavqp_debug_lane_stamps(uavqp_ctx* ctx, long long* out8) {
    if (!ctx || !out8 || !ctx->dbg_queue) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    UAVQP_HIP(hipMemcpy(out8, (char*)ctx->dbg_queue + 128, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return UAVQP_OK;
}
#endif

#ifdef UAVQP_ROWS2_TIMING
// probe build only (tools/rows_sections.py): cycles wave 0 of the last pair-kernel rows solve spent per section
extern "C" int uavqp_debug_rows2_stamps(uavqp_ctx* ctx, long long* out8) {
    if (!ctx || !out8 || !ctx->dbg_queue) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    UAVQP_HIP(hipMemcpy(out8, (char*)ctx->dbg_queue + 64, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return UAVQP_OK;
}
#endif

#ifdef G2_TIMING
// (probe build) s_memtime stamps of wave 0 of the last ragged pair-kernel launch: tools/generic2_sections.py
extern "C" int uavqp_debug_generic2_stamps(uavqp_ctx* ctx, long long* out9) {
    if (!ctx || !out9) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    UAVQP_HIP(hipMemcpy(out9, (char*)ctx->dummy + 1024, 15 * sizeof(long long), hipMemcpyDeviceToHost));
    return UAVQP_OK;
}
#endif

#ifdef UAVQP_DUAL_DEBUG
static double* g_rows_dbg = nullptr;   // dump area of rows_dual_kernel (tools/rows_dual_gpu_probe.py)
#endif
static int rows_batch_impl(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                           const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                           const double* d_bc, const double* d_corr_lo, const double* d_corr_hi, int rows_per_segment,
                           const double* d_row_tau, const int32_t* d_row_deriv, const double* d_row_lo,
                           const double* d_row_hi, double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out,
                           uint64_t* d_active_out, long long total_segments) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0 || (rows_per_segment != 1 && rows_per_segment != 2))
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_waypoints || !d_times || !d_bc || !d_coeff_out || !d_status_out || !d_row_tau || !d_row_deriv || !d_row_lo || !d_row_hi)
        return UAVQP_ERR_INVALID_ARG;
    if ((d_corr_lo == nullptr) != (d_corr_hi == nullptr)) return UAVQP_ERR_INVALID_ARG;
    if (uniform_segments == 0 && (!d_seg_offsets || max_segments < 1)) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    const int Mmax = uniform_segments > 0 ? uniform_segments : max_segments;
    long long rows = 0;
    if (uniform_segments > 0) rows = (long long)n_traj * (uniform_segments + 1);
    else if (total_segments >= 0) rows = total_segments + n_traj;
    else {   // read once here, handed on to the box phase below
        int32_t last = 0;
        UAVQP_HIP(hipMemcpyAsync(&last, d_seg_offsets + n_traj, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        UAVQP_HIP(hipStreamSynchronize(ctx->stream));
        if (last < 0) return UAVQP_ERR_INVALID_ARG;
        rows = (long long)last + n_traj;
    }
    // Phase 1 (only with knot boxes): the box-only problem on the fast corridor kernel; its working set is where the dual method
    // starts, so that its iterations go into the rows and the few boxes they move instead of re-discovering every active box one
    // constraint per block solve (BASELINE config 3 with K = 2 rows: 34 -> 25 iterations mean, 36 -> 30 ms; uavqp_settings.warm_start = 0
    // starts from the empty set).
    const bool warm = d_corr_lo != nullptr && ctx->settings.warm_start != 0;
    const int K_ = rows_per_segment;
    const int K = rows_per_segment, Bk = r + K;
    const bool pair_kernel = ctx->settings.rows_lanes_per_problem != 1;
    // Round 4: the starting set of boxes AND rows from the position-space dual method on the refined time grid (qp_rows_dual.h: a row is a
    // bound on a component of a knot inserted at its time); trajectories it does not take (rows at tau = 0, too many constraints) go
    // through the box phase as before.  uavqp_settings.corridor_initial_guess != 2 or warm_start = 0 switch it off.
    const bool prelude = ctx->settings.corridor_initial_guess == 2 && ctx->settings.warm_start != 0 && pair_kernel &&
                         Mmax >= 2 && (Mmax - 1) + K_ * Mmax <= 48;
    if (warm || prelude) {
        if ((size_t)n_traj * 6 > ctx->rows_warm_count) {
            UAVQP_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->rows_warm) UAVQP_HIP(hipFree(ctx->rows_warm));
            ctx->rows_warm = nullptr;
            ctx->rows_warm_count = 0;
            UAVQP_HIP(hipMalloc((void**)&ctx->rows_warm, sizeof(uint64_t) * (size_t)n_traj * 6));
            ctx->rows_warm_count = (size_t)n_traj * 6;
        }
    }
    const long long total_seg = rows - n_traj;
    const long long pairs = 3LL * n_traj;
    unsigned long long* d_warm_rows = nullptr;
    unsigned char* d_need_phase1 = nullptr;
    double* d_gfun_pre = nullptr;
    unsigned int* d_n_phase1 = nullptr;
    unsigned int* d_queue = nullptr;
    unsigned long long* d_desc = nullptr;
    int32_t* d_status_box = nullptr;
    double* d_kd = nullptr;
    size_t b_kd = 0;
    long long kd_plane = 0;
    uavqp::RowsArgs a;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.max_segments = Mmax;
    a.max_iter = ctx->settings.max_iter > 0 ? ctx->settings.max_iter : 12 * Mmax * (1 + K) + 30;
    a.eps_prim_inf = ctx->settings.eps_prim_inf;
    a.seg_offsets = d_seg_offsets; a.waypoints = d_waypoints; a.times = d_times; a.bc = d_bc;
    a.corr_lo = d_corr_lo; a.corr_hi = d_corr_hi; a.row_tau = d_row_tau; a.row_deriv = d_row_deriv; a.row_lo = d_row_lo; a.row_hi = d_row_hi;
    a.status = d_status_out; a.iters = d_iters_out; a.active = (unsigned long long*)d_active_out;
    a.xsol = nullptr; a.ws = nullptr; a.queue = nullptr; a.warm = nullptr; a.warm_rows = nullptr;
    uavqp::Rows2Args aa{};
    if (pair_kernel) {
        // Round 6: everything the step keeps across its launches lives in ONE persistent allocation (rows_warm2: the box phase in the middle
        // re-uses ctx->ws), and rows_prep_kernel is the step's FIRST kernel: validation + permanent masks as before, and with them the row
        // functionals (was rows_gfun_kernel: a second pass over the same rows), the initial status / iteration counts (was fill_i32_kernel + a
        // memset) and the clearing of what the prelude only writes for the trajectories it takes (was two memsets).
        const size_t b_wr = prelude ? align256(sizeof(uint64_t) * (size_t)n_traj * 3 * 2 * K_) : 0, b_np = prelude ? align256((size_t)n_traj) : 0;
        const size_t b_ctr = 512;                                                                  // [0, 256): the prelude's counters; [256, 512): the pair kernel's queue
        const size_t b_gf = align256(sizeof(double) * (size_t)total_seg * K_ * 2 * r);             // row functionals: made once, used by the prelude AND the rows kernel
        // chain records (rows_chain_kernel): two halves, knot-major planes of n_traj (padded to whole waves) rows each, knots 1 .. Mmax - 1 <= 31
        const int kd_planes = Mmax - 1 < 1 ? 1 : (Mmax - 1 > 31 ? 31 : Mmax - 1);
        kd_plane = ((long long)n_traj + 63) / 64 * 64 * uavqp::rows_chain_half_mem(r);
        b_kd = prelude ? 2 * align256(sizeof(double) * (size_t)kd_planes * (size_t)kd_plane) : 0;
        const size_t b_desc = align256(sizeof(unsigned long long) * (size_t)pairs * (2 + 2 * K));
        const size_t b_sb = warm ? align256(sizeof(int32_t) * (size_t)n_traj) : 0;                 // the box phase's statuses (not the step's)
        const size_t need = b_wr + b_np + b_ctr + b_gf + b_kd + b_desc + b_sb;
        if (need > ctx->rows_warm2_bytes) {
            UAVQP_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->rows_warm2) UAVQP_HIP(hipFree(ctx->rows_warm2));
            ctx->rows_warm2 = nullptr;
            ctx->rows_warm2_bytes = 0;
            UAVQP_HIP(hipMalloc((void**)&ctx->rows_warm2, need));
            ctx->rows_warm2_bytes = need;
        }
        char* p = (char*)ctx->rows_warm2;
        d_warm_rows = prelude ? (unsigned long long*)p : nullptr; p += b_wr;
        d_need_phase1 = prelude ? (unsigned char*)p : nullptr; p += b_np;
        d_n_phase1 = (unsigned int*)p; d_queue = (unsigned int*)(p + 256); p += b_ctr;
        d_gfun_pre = (double*)p; p += b_gf;
        d_kd = (double*)p; p += b_kd;
        d_desc = (unsigned long long*)p; p += b_desc;
        d_status_box = warm ? (int32_t*)p : nullptr;
        a.queue = d_queue;
        ctx->dbg_queue = a.queue;
        aa.r = a;
        aa.desc = d_desc; aa.gfun = d_gfun_pre; aa.prep_gfun = 1;
        aa.init_warm_box = (warm || prelude) ? (unsigned long long*)ctx->rows_warm : nullptr;
        aa.init_warm_rows = d_warm_rows;
        aa.init_counters = d_n_phase1;                      // 128 words: the prelude's counters and the queue behind them
        // rows_prep_kernel: four waves per block; two trajectories per wave when none has more than 31 segments (a half-wave's lanes), else one
        const int tw = Mmax <= 31 ? 2 : 1;
        long long pgrid = ((long long)n_traj + 4 * tw - 1) / (4 * tw);
        if (pgrid > (long long)ctx->num_cus * 16) pgrid = (long long)ctx->num_cus * 16;
#define UAVQP_PREP_LAUNCH(R_, K_) do { if (tw == 2) hipLaunchKernelGGL((uavqp::rows_prep_kernel<R_, K_, 2>), dim3((unsigned)pgrid), dim3(256), 0, ctx->stream, aa); \
                                       else hipLaunchKernelGGL((uavqp::rows_prep_kernel<R_, K_, 1>), dim3((unsigned)pgrid), dim3(256), 0, ctx->stream, aa); } while (0)
        if (r == 3 && K == 1) UAVQP_PREP_LAUNCH(3, 1);
        else if (r == 3) UAVQP_PREP_LAUNCH(3, 2);
        else if (K == 1) UAVQP_PREP_LAUNCH(4, 1);
        else UAVQP_PREP_LAUNCH(4, 2);
#undef UAVQP_PREP_LAUNCH
    }
    if (prelude) {
        uavqp::RowsDualArgs da{};
        da.r.n_traj = n_traj; da.r.uniform = uniform_segments; da.r.max_segments = Mmax;
        da.r.seg_offsets = d_seg_offsets; da.r.waypoints = d_waypoints; da.r.times = d_times; da.r.bc = d_bc;
        da.r.corr_lo = d_corr_lo; da.r.corr_hi = d_corr_hi; da.r.row_tau = d_row_tau; da.r.row_deriv = d_row_deriv; da.r.row_lo = d_row_lo; da.r.row_hi = d_row_hi;
        da.order = nullptr;
        da.warm_box = (unsigned long long*)ctx->rows_warm; da.warm_rows = d_warm_rows; da.need_phase1 = d_need_phase1; da.gfun = d_gfun_pre;
        da.kdF = d_kd; da.kdB = (const double*)((const char*)d_kd + b_kd / 2); da.kd_plane = kd_plane; da.n_phase1 = d_n_phase1;
        da.ticket = (d_n_phase1 && ctx->deal_tickets) ? d_n_phase1 + 16 : nullptr;      // (the prelude's counter block: zeroed by rows_prep_kernel, the step's first kernel)
        {   // the chain of every trajectory once, one lane each (the prelude's waves would each repeat it in 64 lanes)
            long long cg = ((long long)n_traj + 63) / 64;
            if (cg > (long long)ctx->num_cus * 16) cg = (long long)ctx->num_cus * 16;
            if (r == 3) hipLaunchKernelGGL((uavqp::rows_chain_kernel<3>), dim3((unsigned)cg), dim3(64), 0, ctx->stream, da.r, d_kd, (double*)((char*)d_kd + b_kd / 2), kd_plane);
            else hipLaunchKernelGGL((uavqp::rows_chain_kernel<4>), dim3((unsigned)cg), dim3(64), 0, ctx->stream, da.r, d_kd, (double*)((char*)d_kd + b_kd / 2), kd_plane);
        }
#ifdef UAVQP_DUAL_DEBUG
        {
            if (!g_rows_dbg) { UAVQP_HIP(hipMalloc(&g_rows_dbg, 64 * 2048 * sizeof(double))); }   // (the size uavqp_debug_corridor_dual copies)
            UAVQP_HIP(hipMemsetAsync(g_rows_dbg, 0, 64 * 2048 * sizeof(double), ctx->stream));
            da.dbg = g_rows_dbg;
        }
#endif
        // one trajectory per wave; waves per CU by the LDS of a block (two per SIMD at most: 256 registers)
        int wpc_r = (160 * 1024) / (8 * uavqp::rows_dual_lds_doubles(r));
        if (wpc_r > 8) wpc_r = 8;
        const long long dgrid = (long long)n_traj < (long long)ctx->num_cus * wpc_r ? (long long)n_traj : (long long)ctx->num_cus * wpc_r;
        if (r == 3 && K_ == 1) hipLaunchKernelGGL((uavqp::rows_dual_kernel<3, 1>), dim3((unsigned)dgrid), dim3(64), 0, ctx->stream, da, 0);
        else if (r == 3) hipLaunchKernelGGL((uavqp::rows_dual_kernel<3, 2>), dim3((unsigned)dgrid), dim3(64), 0, ctx->stream, da, 0);
        else if (K_ == 1) hipLaunchKernelGGL((uavqp::rows_dual_kernel<4, 1>), dim3((unsigned)dgrid), dim3(64), 0, ctx->stream, da, 0);
        else hipLaunchKernelGGL((uavqp::rows_dual_kernel<4, 2>), dim3((unsigned)dgrid), dim3(64), 0, ctx->stream, da, 0);
    }
    if (warm && !prelude) {
        // The box phase -- only WITHOUT the prelude.  (Round 6: with it, what the prelude does not take is, by the host's own admission test
        // (Mmax - 1) + K Mmax <= 48, exactly what rows_prep_kernel flags as invalid input -- bad durations, bad rows -- or a single segment, which has no
        // knot box to start from: the three launches that compacted and solved that list -- empty as a rule -- are gone; need_phase1 is still written,
        // nobody reads it.)  The box phase's statuses are its own: the step's statuses started in rows_prep_kernel.
        const int rc1 = corridor_warm_impl(ctx, r, n_traj, uniform_segments, max_segments, d_seg_offsets, d_waypoints, d_times,
                                           d_bc, d_corr_lo, d_corr_hi, d_coeff_out, pair_kernel ? d_status_box : d_status_out, nullptr, ctx->rows_warm, 0, rows - n_traj);
        if (rc1 != UAVQP_OK) return rc1;
    }
#ifdef UAVQP_DUAL_DEBUG
    if (prelude) {   // (after the box phase, which points these at its own dump)
        static unsigned long long* s_box = nullptr;
        static size_t s_box_n = 0;
        if (s_box_n < (size_t)n_traj * 6) { if (s_box) (void)hipFree(s_box); UAVQP_HIP(hipMalloc((void**)&s_box, sizeof(uint64_t) * (size_t)n_traj * 6)); s_box_n = (size_t)n_traj * 6; }
        UAVQP_HIP(hipMemcpyAsync(s_box, ctx->rows_warm, sizeof(uint64_t) * (size_t)n_traj * 6, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->dbg_guess = s_box;
        ctx->dbg_dual = g_rows_dbg;
    }
#endif
    a.warm = (warm || prelude) ? (const unsigned long long*)ctx->rows_warm : nullptr;
    a.warm_rows = prelude ? (const unsigned long long*)d_warm_rows : nullptr;
    int rc;
    if (pair_kernel) {
        // two lanes per problem, sweep state in LDS (qp_rows2.h): 80 KiB per single-wave workgroup = two waves per CU
        const int Fp = Bk * (Bk + 1) / 2 + Bk, NCN = 1 + K;
        const int kown = (Mmax + 1) / 2;                       // own knots of the longer half, meeting knot included = state slots 0..kown-1
        // the general passes keep 80 KiB of sweep records per wave (two waves per CU); the verifying pass 40 KiB (four: one per SIMD) and the rest in
        // the HBM workspace -- each pass has its own count of workspace knots and its own grid
        // (the verifying pass also keeps rows2_reg_knots() slots in registers -- unless the batch needs workspace slots anyway: that instantiation has none)
        const int NT = uavqp::rows2_lds_knots(r, K, false), NTv = uavqp::rows2_lds_knots(r, K, true);
        const int ws_knots = kown > NT ? kown - NT : 0, ws_knots_v = kown > NTv + uavqp::rows2_reg_knots(true) ? kown - NTv : 0;
        long long grid = (pairs + 31) / 32, grid_v = grid;
        const long long max_grid = (long long)ctx->num_cus * uavqp::rows2_waves_per_cu(false), max_grid_v = (long long)ctx->num_cus * uavqp::rows2_waves_per_cu(true);
        if (grid > max_grid) grid = max_grid;
        if (grid_v > max_grid_v) grid_v = max_grid_v;
        const bool deal_by_length = uniform_segments == 0 && n_traj >= 64 && ctx->settings.ragged_window_sort;
        const size_t b_order = deal_by_length ? align256(sizeof(int32_t) * (size_t)n_traj) + 2048 : 0;
        const size_t b_state_g = sizeof(double) * (size_t)ws_knots * Fp * (size_t)grid * 64, b_state_v = prelude ? sizeof(double) * (size_t)ws_knots_v * Fp * (size_t)grid_v * 64 : 0;
        const size_t b_state = align256(b_state_g > b_state_v ? b_state_g : b_state_v);
        const size_t b_lam = align256(sizeof(double) * (size_t)kown * 2 * NCN * (size_t)grid * 64);
        const size_t b_redo = align256(sizeof(unsigned int) * (size_t)pairs);
        rc = ensure_ws(ctx, 256 + b_order + b_state + b_lam + b_redo);
        if (rc != UAVQP_OK) return rc;
        char* p = (char*)ctx->ws + 256;
        aa.order = nullptr;
        if (deal_by_length) {
            int32_t* d_order = (int32_t*)p;
            int* d_hist = (int*)(p + b_order - 2048);
            int* d_cursor = d_hist + 256;
            UAVQP_HIP(hipMemsetAsync(d_hist, 0, 256 * sizeof(int), ctx->stream));
            int sg = (n_traj + 255) / 256;
            if (sg > ctx->num_cus * 4) sg = ctx->num_cus * 4;
            hipLaunchKernelGGL(uavqp::seg_hist_kernel, dim3(sg), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, d_hist);
            hipLaunchKernelGGL(uavqp::seg_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_hist, d_cursor);
            hipLaunchKernelGGL(uavqp::seg_scatter_kernel, dim3(sg), dim3(256), 0, ctx->stream, d_seg_offsets, n_traj, d_cursor, d_order);
            aa.order = d_order;
        }
        p += b_order;
        a.ws = (double*)p; p += b_state;
        aa.lam = (double*)p; p += b_lam;
        aa.redo = (unsigned int*)p;
        aa.ws_knots = ws_knots;
        aa.lam_knots = kown;
        aa.coeff = d_coeff_out;                                 // the pair kernels write the polynomials themselves: no xsol, no emission launch
        aa.prep_gfun = 0; aa.init_warm_box = nullptr; aa.init_warm_rows = nullptr; aa.init_counters = nullptr;
        aa.r = a;
        // first pass: every problem.  With the prelude's starting sets it is the VERIFYING pass (one block solve, no dual state in HBM; what it
        // does not confirm goes to the redo list); without them the general first pass.  Second pass (Goldfarb-Idnani's dependent-constraint
        // route, and every problem the verifying pass handed on): the redo list -- empty as a rule (one wave per CU then; behind the
        // verifying pass the full grid: its list may be long when the prelude did not take the batch)
        const long long grid2 = prelude ? grid : (grid < (long long)ctx->num_cus ? grid : (long long)ctx->num_cus);
        uavqp::Rows2Args av = aa;                               // the verifying pass: its own workspace split
        av.ws_knots = ws_knots_v;
#define UAVQP_ROWS2(RR, KK)                                                                                                                  \
    do {                                                                                                                                     \
        if (prelude) {                                                                                                                      \
            if (ws_knots_v > 0) hipLaunchKernelGGL((uavqp::rows_pair_kernel<RR, KK, true, false, true>), dim3((unsigned)grid_v), dim3(64), 0, ctx->stream, av); \
            else hipLaunchKernelGGL((uavqp::rows_pair_kernel<RR, KK, false, false, true>), dim3((unsigned)grid_v), dim3(64), 0, ctx->stream, av);               \
        } else if (ws_knots > 0) {                                                                                                          \
            hipLaunchKernelGGL((uavqp::rows_pair_kernel<RR, KK, true, false, false>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, aa);  \
        } else {                                                                                                                            \
            hipLaunchKernelGGL((uavqp::rows_pair_kernel<RR, KK, false, false, false>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, aa); \
        }                                                                                                                                   \
        if (ws_knots > 0) hipLaunchKernelGGL((uavqp::rows_pair_kernel<RR, KK, true, true, false>), dim3((unsigned)grid2), dim3(64), 0, ctx->stream, aa); \
        else hipLaunchKernelGGL((uavqp::rows_pair_kernel<RR, KK, false, true, false>), dim3((unsigned)grid2), dim3(64), 0, ctx->stream, aa);           \
    } while (0)
        if (r == 3 && K == 1) UAVQP_ROWS2(3, 1);
        else if (r == 3) UAVQP_ROWS2(3, 2);
        else if (K == 1) UAVQP_ROWS2(4, 1);
        else UAVQP_ROWS2(4, 2);
#undef UAVQP_ROWS2
    } else {
        const size_t b_xsol = align256(sizeof(double) * 3 * (size_t)r * (size_t)rows);
        const int F = Bk * (Bk + 1) / 2 + Bk + 2 * (1 + K);   // must match rows_solve_kernel's state layout
        long long grid = (3LL * n_traj + 63) / 64;
        const long long max_grid = (long long)ctx->num_cus * 4;
        if (grid > max_grid) grid = max_grid;
        const size_t b_state = sizeof(double) * (size_t)Mmax * F * (size_t)grid * 64;
        rc = ensure_ws(ctx, b_xsol + 256 + b_state);
        if (rc != UAVQP_OK) return rc;
        a.xsol = ctx->ws; a.queue = (unsigned int*)((char*)ctx->ws + b_xsol); a.ws = (double*)((char*)ctx->ws + b_xsol + 256);
        UAVQP_HIP(hipMemsetAsync(a.queue, 0, 256, ctx->stream));
        hipLaunchKernelGGL(uavqp::fill_i32_kernel, dim3((n_traj + 255) / 256), dim3(256), 0, ctx->stream, d_status_out, n_traj, (int32_t)UAVQP_SOLVED);
        if (d_iters_out) UAVQP_HIP(hipMemsetAsync(d_iters_out, 0, sizeof(int32_t) * (size_t)n_traj, ctx->stream));
        if (r == 3) {
            if (K == 1) hipLaunchKernelGGL((uavqp::rows_solve_kernel<3, 1>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL((uavqp::rows_solve_kernel<3, 2>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
        } else {
            if (K == 1) hipLaunchKernelGGL((uavqp::rows_solve_kernel<4, 1>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL((uavqp::rows_solve_kernel<4, 2>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a);
        }
        // Hermite solution -> coefficients (the corridor solver's emission kernel; it reads the same fields) -- the one-lane cross-check kernel only
        uavqp::CorridorArgs e{};
        e.n_traj = n_traj; e.uniform = uniform_segments; e.max_segments = Mmax; e.seg_offsets = d_seg_offsets; e.waypoints = d_waypoints;
        e.times = d_times; e.bc = d_bc; e.coeff = d_coeff_out; e.status = d_status_out; e.xsol = ctx->ws;
        const long long chunks = 3LL * (rows - n_traj);
        long long egrid = (chunks + 255) / 256;
        if (egrid > (long long)ctx->num_cus * 16) egrid = (long long)ctx->num_cus * 16;
        if (chunks > 0) {
            if (r == 3) hipLaunchKernelGGL((uavqp::corridor_emit_kernel<3>), dim3((unsigned)egrid), dim3(256), 0, ctx->stream, e, chunks);
            else hipLaunchKernelGGL((uavqp::corridor_emit_kernel<4>), dim3((unsigned)egrid), dim3(256), 0, ctx->stream, e, chunks);
        }
    }
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

extern "C" int uavqp_solve_rows_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                             const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                                             const double* d_bc, const double* d_corr_lo, const double* d_corr_hi, int rows_per_segment,
                                             const double* d_row_tau, const int32_t* d_row_deriv, const double* d_row_lo,
                                             const double* d_row_hi, double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out,
                                             uint64_t* d_active_out) {
    return rows_batch_impl(ctx, r, n_traj, uniform_segments, max_segments, d_seg_offsets, d_waypoints, d_times, d_bc, d_corr_lo, d_corr_hi,
                           rows_per_segment, d_row_tau, d_row_deriv, d_row_lo, d_row_hi, d_coeff_out, d_status_out, d_iters_out, d_active_out, -1);
}

extern "C" int uavqp_solve_corridor_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                                 const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                                                 const double* d_bc, const double* d_corr_lo, const double* d_corr_hi,
                                                 double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out) {
    return uavqp_solve_corridor_warm_device(ctx, r, n_traj, uniform_segments, max_segments, d_seg_offsets, d_waypoints, d_times, d_bc,
                                            d_corr_lo, d_corr_hi, d_coeff_out, d_status_out, d_iters_out, nullptr, 0);
}

extern "C" int uavqp_solve_corridor_batch_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                               const int32_t* seg_offsets, const double* waypoints, const double* times,
                                               const double* bc, const double* corr_lo, const double* corr_hi,
                                               double* coeff_out, int32_t* status_out, int32_t* iters_out) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0) return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!waypoints || !times || !bc || !corr_lo || !corr_hi || !coeff_out) return UAVQP_ERR_INVALID_ARG;
    if (uniform_segments == 0 && !seg_offsets) return UAVQP_ERR_INVALID_ARG;
    long long total_seg = 0;
    int Mmax = uniform_segments;
    if (uniform_segments > 0) total_seg = (long long)uniform_segments * n_traj;
    else {
        if (seg_offsets[0] != 0) return UAVQP_ERR_INVALID_ARG;
        for (int b = 0; b < n_traj; ++b) {
            const int M = seg_offsets[b + 1] - seg_offsets[b];
            if (M < 0) return UAVQP_ERR_INVALID_ARG;
            if (M > Mmax) Mmax = M;
        }
        total_seg = seg_offsets[n_traj];
        if (max_segments > 0 && max_segments < Mmax) Mmax = max_segments;
        if (Mmax < 1) Mmax = 1;
    }
    UAVQP_HIP(hipSetDevice(ctx->device));
    const size_t n_wp = 3 * (size_t)(total_seg + n_traj);
    const size_t b_off = uniform_segments > 0 ? 0 : align256(sizeof(int32_t) * (size_t)(n_traj + 1));
    const size_t b_wp = align256(sizeof(double) * n_wp);
    const size_t b_t = align256(sizeof(double) * (size_t)total_seg);
    const size_t b_bc = align256(sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3);
    const size_t b_out = align256(sizeof(double) * 3 * 2 * r * (size_t)total_seg);
    const size_t b_st = align256(sizeof(int32_t) * (size_t)n_traj);
    int rc = ensure_stage(ctx, b_off + 3 * b_wp + b_t + b_bc + b_out + 2 * b_st);
    if (rc != UAVQP_OK) return rc;
    char* p = (char*)ctx->d_stage;
    int32_t* d_off = uniform_segments > 0 ? nullptr : (int32_t*)p; p += b_off;
    double* d_wp = (double*)p; p += b_wp;
    double* d_lo = (double*)p; p += b_wp;
    double* d_hi = (double*)p; p += b_wp;
    double* d_t = (double*)p; p += b_t;
    double* d_bc = (double*)p; p += b_bc;
    double* d_out = (double*)p; p += b_out;
    int32_t* d_st = (int32_t*)p; p += b_st;
    int32_t* d_it = (int32_t*)p;
    hipStream_t s = ctx->stream;
    if (d_off) UAVQP_HIP(hipMemcpyAsync(d_off, seg_offsets, sizeof(int32_t) * (size_t)(n_traj + 1), hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_wp, waypoints, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_lo, corr_lo, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_hi, corr_hi, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
    if (total_seg > 0) UAVQP_HIP(hipMemcpyAsync(d_t, times, sizeof(double) * (size_t)total_seg, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_bc, bc, sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * 3 * 2 * r * (size_t)total_seg, s));
    rc = corridor_warm_impl(ctx, r, n_traj, uniform_segments, Mmax, d_off, d_wp, d_t, d_bc, d_lo, d_hi, d_out, d_st, d_it, nullptr, 0, total_seg);
    if (rc != UAVQP_OK) return rc;
    if (total_seg > 0) UAVQP_HIP(hipMemcpyAsync(coeff_out, d_out, sizeof(double) * 3 * 2 * r * (size_t)total_seg, hipMemcpyDeviceToHost, s));
    if (status_out) UAVQP_HIP(hipMemcpyAsync(status_out, d_st, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    if (iters_out) UAVQP_HIP(hipMemcpyAsync(iters_out, d_it, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    UAVQP_HIP(hipStreamSynchronize(s));
    return UAVQP_OK;
}

extern "C" int uavqp_solve_rows_batch_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                           const int32_t* seg_offsets, const double* waypoints, const double* times, const double* bc,
                                           const double* corr_lo, const double* corr_hi, int rows_per_segment, const double* row_tau,
                                           const int32_t* row_deriv, const double* row_lo, const double* row_hi, double* coeff_out,
                                           int32_t* status_out, int32_t* iters_out) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0 || (rows_per_segment != 1 && rows_per_segment != 2))
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!waypoints || !times || !bc || !coeff_out || !row_tau || !row_deriv || !row_lo || !row_hi) return UAVQP_ERR_INVALID_ARG;
    if ((corr_lo == nullptr) != (corr_hi == nullptr)) return UAVQP_ERR_INVALID_ARG;
    if (uniform_segments == 0 && !seg_offsets) return UAVQP_ERR_INVALID_ARG;
    long long total_seg = 0;
    int Mmax = uniform_segments;
    if (uniform_segments > 0) total_seg = (long long)uniform_segments * n_traj;
    else {
        if (seg_offsets[0] != 0) return UAVQP_ERR_INVALID_ARG;
        for (int b = 0; b < n_traj; ++b) {
            const int M = seg_offsets[b + 1] - seg_offsets[b];
            if (M < 0) return UAVQP_ERR_INVALID_ARG;
            if (M > Mmax) Mmax = M;
        }
        total_seg = seg_offsets[n_traj];
        if (max_segments > 0 && max_segments < Mmax) Mmax = max_segments;
        if (Mmax < 1) Mmax = 1;
    }
    UAVQP_HIP(hipSetDevice(ctx->device));
    const int K = rows_per_segment;
    const size_t n_wp = 3 * (size_t)(total_seg + n_traj), n_row = (size_t)total_seg * K;
    const size_t b_off = uniform_segments > 0 ? 0 : align256(sizeof(int32_t) * (size_t)(n_traj + 1));
    const size_t b_wp = align256(sizeof(double) * n_wp);
    const size_t b_t = align256(sizeof(double) * (size_t)total_seg);
    const size_t b_bc = align256(sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3);
    const size_t b_rt = align256(sizeof(double) * n_row), b_rd = align256(sizeof(int32_t) * n_row), b_rb = align256(sizeof(double) * 3 * n_row);
    const size_t b_out = align256(sizeof(double) * 3 * 2 * r * (size_t)total_seg);
    const size_t b_st = align256(sizeof(int32_t) * (size_t)n_traj);
    int rc = ensure_stage(ctx, b_off + 3 * b_wp + b_t + b_bc + b_rt + b_rd + 2 * b_rb + b_out + 2 * b_st);
    if (rc != UAVQP_OK) return rc;
    char* p = (char*)ctx->d_stage;
    int32_t* d_off = uniform_segments > 0 ? nullptr : (int32_t*)p; p += b_off;
    double* d_wp = (double*)p; p += b_wp;
    double* d_lo = (double*)p; p += b_wp;
    double* d_hi = (double*)p; p += b_wp;
    double* d_t = (double*)p; p += b_t;
    double* d_bc = (double*)p; p += b_bc;
    double* d_rt = (double*)p; p += b_rt;
    int32_t* d_rd = (int32_t*)p; p += b_rd;
    double* d_rl = (double*)p; p += b_rb;
    double* d_rh = (double*)p; p += b_rb;
    double* d_out = (double*)p; p += b_out;
    int32_t* d_st = (int32_t*)p; p += b_st;
    int32_t* d_it = (int32_t*)p;
    hipStream_t s = ctx->stream;
    if (d_off) UAVQP_HIP(hipMemcpyAsync(d_off, seg_offsets, sizeof(int32_t) * (size_t)(n_traj + 1), hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_wp, waypoints, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
    if (corr_lo) {
        UAVQP_HIP(hipMemcpyAsync(d_lo, corr_lo, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
        UAVQP_HIP(hipMemcpyAsync(d_hi, corr_hi, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
    }
    if (total_seg > 0) {
        UAVQP_HIP(hipMemcpyAsync(d_t, times, sizeof(double) * (size_t)total_seg, hipMemcpyHostToDevice, s));
        UAVQP_HIP(hipMemcpyAsync(d_rt, row_tau, sizeof(double) * n_row, hipMemcpyHostToDevice, s));
        UAVQP_HIP(hipMemcpyAsync(d_rd, row_deriv, sizeof(int32_t) * n_row, hipMemcpyHostToDevice, s));
        UAVQP_HIP(hipMemcpyAsync(d_rl, row_lo, sizeof(double) * 3 * n_row, hipMemcpyHostToDevice, s));
        UAVQP_HIP(hipMemcpyAsync(d_rh, row_hi, sizeof(double) * 3 * n_row, hipMemcpyHostToDevice, s));
    }
    UAVQP_HIP(hipMemcpyAsync(d_bc, bc, sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * 3 * 2 * r * (size_t)total_seg, s));   // failed trajectories come back as zeros
    rc = rows_batch_impl(ctx, r, n_traj, uniform_segments, Mmax, d_off, d_wp, d_t, d_bc, corr_lo ? d_lo : nullptr,
                         corr_lo ? d_hi : nullptr, K, d_rt, d_rd, d_rl, d_rh, d_out, d_st, d_it, nullptr, total_seg);
    if (rc != UAVQP_OK) return rc;
    if (total_seg > 0) UAVQP_HIP(hipMemcpyAsync(coeff_out, d_out, sizeof(double) * 3 * 2 * r * (size_t)total_seg, hipMemcpyDeviceToHost, s));
    if (status_out) UAVQP_HIP(hipMemcpyAsync(status_out, d_st, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    if (iters_out) UAVQP_HIP(hipMemcpyAsync(iters_out, d_it, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    UAVQP_HIP(hipStreamSynchronize(s));
    return UAVQP_OK;
}

static int time_reallocate_impl(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                double* d_times, const double* d_coeff, double v_max, double a_max,
                                int samples_per_seg, double max_stretch, int32_t* d_changed_out, double* d_scale_acc,
                                const int32_t* d_list = nullptr, const int* d_n_list = nullptr);
extern "C" int uavqp_time_reallocate_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                            double* d_times, const double* d_coeff, double v_max, double a_max,
                                            int samples_per_seg, double max_stretch, int32_t* d_changed_out) {
    return time_reallocate_impl(ctx, r, n_traj, uniform_segments, d_seg_offsets, d_times, d_coeff, v_max, a_max, samples_per_seg, max_stretch, d_changed_out, nullptr);
}
static int time_reallocate_impl(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                double* d_times, const double* d_coeff, double v_max, double a_max,
                                int samples_per_seg, double max_stretch, int32_t* d_changed_out, double* d_scale_acc,
                                const int32_t* d_list, const int* d_n_list) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0 || !(v_max > 0.0) || !(a_max > 0.0) ||
        samples_per_seg < 1 || !(max_stretch > 1.0))
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_times || !d_coeff || (uniform_segments == 0 && !d_seg_offsets)) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    uavqp::ReallocArgs a;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.samples = samples_per_seg;
    a.seg_offsets = d_seg_offsets; a.times = d_times; a.coeff = d_coeff; a.v_max = v_max; a.a_max = a_max;
    a.max_stretch = max_stretch; a.changed = d_changed_out; a.scale_acc = d_scale_acc;
    a.list = d_list; a.n_list = d_list ? d_n_list : nullptr;
    a.dead_band = ctx->settings.realloc_dead_band; a.overshoot = ctx->settings.realloc_overshoot;
    long long grid_ll = ((long long)n_traj * 8 + 63) / 64;  // 8 lanes per trajectory
    if (grid_ll > (long long)ctx->num_cus * 32) grid_ll = (long long)ctx->num_cus * 32;
    const int grid = (int)grid_ll;
    if (r == 3)
        hipLaunchKernelGGL(uavqp::realloc_kernel<3>, dim3(grid), dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(uavqp::realloc_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

extern "C" int uavqp_ellipsoid_check_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                            const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                            const double* d_obstacles, int n_obs, double robot_r, double robot_h,
                                            int32_t* d_first_hit, uint8_t* d_flags) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || n_samples < 0 || uniform_segments < 0 || n_obs < 0 || !(robot_r > 0.0) || !(robot_h > 0.0))
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_times || !d_coeff || !d_first_hit || (n_obs > 0 && !d_obstacles) || (uniform_segments == 0 && !d_seg_offsets))
        return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(uavqp::fill_i32_kernel, dim3((n_traj + 255) / 256), dim3(256), 0, ctx->stream, d_first_hit, n_traj, (int32_t)n_samples);
    if (n_samples == 0) return UAVQP_OK;
    uavqp::EllipsoidArgs a;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.n_samples = n_samples; a.n_obs = n_obs;
    a.seg_offsets = d_seg_offsets; a.times = d_times; a.coeff = d_coeff; a.obs = d_obstacles;
    a.t0 = t0; a.dt = dt; a.robot_r = robot_r; a.robot_h = robot_h; a.first_hit = d_first_hit; a.flags = d_flags;
    const long long total = (long long)n_traj * n_samples;
    long long grid = (total + 255) / 256;
    if (grid > (long long)ctx->num_cus * 16) grid = (long long)ctx->num_cus * 16;
    if (r == 3)
        hipLaunchKernelGGL(uavqp::ellipsoid_kernel<3>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(uavqp::ellipsoid_kernel<4>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

// ---- obstacle grid (uniform cells over the cloud) + grid-accelerated ellipsoid check ------------------------------
struct uavqp_grid {
    int device = 0;
    int n_obs = 0;
    uavqp::GridView view{};
    int32_t* d_cell_start = nullptr;
    double* d_pts = nullptr;
};

extern "C" int uavqp_obstacle_grid_destroy(uavqp_ctx* ctx, uavqp_grid* grid) {
    if (!grid) return UAVQP_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    (void)hipSetDevice(grid->device);
    if (grid->d_cell_start) (void)hipFree(grid->d_cell_start);
    if (grid->d_pts) (void)hipFree(grid->d_pts);
    delete grid;
    return UAVQP_OK;
}

extern "C" int uavqp_obstacle_grid_build_device(uavqp_ctx* ctx, const double* d_obstacles, int n_obs, double cell_size,
                                                uavqp_grid** out_grid) {
    if (!ctx || !out_grid || n_obs < 0 || (n_obs > 0 && !d_obstacles) || !(cell_size > 0.0) || !(cell_size < INFINITY))
        return UAVQP_ERR_INVALID_ARG;
    *out_grid = nullptr;
    UAVQP_HIP(hipSetDevice(ctx->device));
    uavqp_grid* g = new (std::nothrow) uavqp_grid();
    if (!g) return UAVQP_ERR_ALLOC;
    g->device = ctx->device;
    g->n_obs = n_obs;
    hipStream_t s = ctx->stream;
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    int rc = UAVQP_OK;
    double* d_part = nullptr;
    int32_t* d_cell_of = nullptr;
    int32_t* d_cursor = nullptr;
    std::vector<int32_t> h_counts;
#define UAVQP_GRID_HIP(expr)                                                                     \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            rc = UAVQP_ERR_HIP;                                                                  \
            goto fail;                                                                           \
        }                                                                                        \
    } while (0)
    {
        if (n_obs > 0) {
            const int nb = std::min(256, (n_obs + 255) / 256);
            UAVQP_GRID_HIP(hipMalloc((void**)&d_part, sizeof(double) * 6 * nb));
            hipLaunchKernelGGL(uavqp::grid_bounds_kernel, dim3(nb), dim3(256), 0, s, d_obstacles, n_obs, d_part);
            std::vector<double> part(6 * (size_t)nb);
            UAVQP_GRID_HIP(hipMemcpyAsync(part.data(), d_part, sizeof(double) * 6 * nb, hipMemcpyDeviceToHost, s));
            UAVQP_GRID_HIP(hipStreamSynchronize(s));
            for (int ax = 0; ax < 3; ++ax) { mn[ax] = INFINITY; mx[ax] = -INFINITY; }
            for (int b = 0; b < nb; ++b)
                for (int ax = 0; ax < 3; ++ax) {
                    mn[ax] = std::fmin(mn[ax], part[6 * b + ax]);
                    mx[ax] = std::fmax(mx[ax], part[6 * b + 3 + ax]);
                }
            for (int ax = 0; ax < 3; ++ax)
                if (!(mn[ax] > -INFINITY) || !(mx[ax] < INFINITY) || mn[ax] != mn[ax]) {
                    g_last_error = "obstacle cloud contains non-finite coordinates";
                    rc = UAVQP_ERR_INVALID_ARG;
                    goto fail;
                }
        }
        // cells: at most 2^22; a cloud too large for the requested cell size gets proportionally larger cells
        double cell = cell_size;
        long long dims[3];
        for (;;) {
            long long n = 1;
            for (int ax = 0; ax < 3; ++ax) {
                dims[ax] = (long long)std::floor((mx[ax] - mn[ax]) / cell) + 1;
                n *= dims[ax];
            }
            if (n <= (1ll << 22)) break;
            cell *= 2.0;
        }
        const int ncell = (int)(dims[0] * dims[1] * dims[2]);
        for (int ax = 0; ax < 3; ++ax) { g->view.org[ax] = mn[ax]; g->view.dim[ax] = (int)dims[ax]; }
        g->view.inv_cell = 1.0 / cell;
        UAVQP_GRID_HIP(hipMalloc((void**)&g->d_cell_start, sizeof(int32_t) * (size_t)(ncell + 1)));
        UAVQP_GRID_HIP(hipMalloc((void**)&g->d_pts, sizeof(double) * 3 * (size_t)(n_obs > 0 ? n_obs : 1)));
        UAVQP_GRID_HIP(hipMemsetAsync(g->d_cell_start, 0, sizeof(int32_t) * (size_t)(ncell + 1), s));
        if (n_obs > 0) {
            UAVQP_GRID_HIP(hipMalloc((void**)&d_cell_of, sizeof(int32_t) * (size_t)n_obs));
            UAVQP_GRID_HIP(hipMalloc((void**)&d_cursor, sizeof(int32_t) * (size_t)(ncell + 1)));
            // counts land in cell_start[c + 1] so that the host prefix leaves an exclusive scan in place
            hipLaunchKernelGGL(uavqp::grid_count_kernel, dim3((n_obs + 255) / 256), dim3(256), 0, s, g->view, d_obstacles, n_obs,
                               d_cell_of, g->d_cell_start + 1);
            h_counts.resize((size_t)ncell + 1);
            UAVQP_GRID_HIP(hipMemcpyAsync(h_counts.data(), g->d_cell_start, sizeof(int32_t) * (size_t)(ncell + 1), hipMemcpyDeviceToHost, s));
            UAVQP_GRID_HIP(hipStreamSynchronize(s));
            for (int c = 0; c < ncell; ++c) h_counts[c + 1] += h_counts[c];
            UAVQP_GRID_HIP(hipMemcpyAsync(g->d_cell_start, h_counts.data(), sizeof(int32_t) * (size_t)(ncell + 1), hipMemcpyHostToDevice, s));
            UAVQP_GRID_HIP(hipMemcpyAsync(d_cursor, h_counts.data(), sizeof(int32_t) * (size_t)(ncell + 1), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(uavqp::grid_scatter_kernel, dim3((n_obs + 255) / 256), dim3(256), 0, s, d_obstacles, n_obs, d_cell_of,
                               d_cursor, g->d_pts);
            UAVQP_GRID_HIP(hipGetLastError());
            UAVQP_GRID_HIP(hipStreamSynchronize(s));
        }
        g->view.cell_start = g->d_cell_start;
        g->view.pts = g->d_pts;
    }
    if (d_part) (void)hipFree(d_part);
    if (d_cell_of) (void)hipFree(d_cell_of);
    if (d_cursor) (void)hipFree(d_cursor);
    *out_grid = g;
    return UAVQP_OK;
fail:
    if (d_part) (void)hipFree(d_part);
    if (d_cell_of) (void)hipFree(d_cell_of);
    if (d_cursor) (void)hipFree(d_cursor);
    (void)uavqp_obstacle_grid_destroy(nullptr, g);
    return rc;
#undef UAVQP_GRID_HIP
}

// (d_tmax_bits: the pipeline's form -- dt = [the double whose bits are *d_tmax_bits] / (n_samples - 1), formed by the kernel, and
// d_first_hit already holds n_samples: the host neither waits for the longest duration nor launches the fill)
static int ellipsoid_check_grid_impl(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                     const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                     const unsigned long long* d_tmax_bits, const uavqp_grid* grid, double robot_r, double robot_h,
                                     int32_t* d_first_hit, uint8_t* d_flags);
extern "C" int uavqp_ellipsoid_check_grid_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                                 const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                                 const uavqp_grid* grid, double robot_r, double robot_h, int32_t* d_first_hit,
                                                 uint8_t* d_flags) {
    return ellipsoid_check_grid_impl(ctx, r, n_traj, uniform_segments, d_seg_offsets, d_times, d_coeff, n_samples, t0, dt, nullptr, grid, robot_r,
                                     robot_h, d_first_hit, d_flags);
}
static int ellipsoid_check_grid_impl(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                     const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                     const unsigned long long* d_tmax_bits, const uavqp_grid* grid, double robot_r, double robot_h,
                                     int32_t* d_first_hit, uint8_t* d_flags) {
    if (!ctx || !grid || (r != 3 && r != 4) || n_traj < 0 || n_samples < 0 || uniform_segments < 0 || !(robot_r > 0.0) || !(robot_h > 0.0))
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0) return UAVQP_OK;
    if (!d_times || !d_coeff || !d_first_hit || (uniform_segments == 0 && !d_seg_offsets) || grid->device != ctx->device)
        return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    if (!d_tmax_bits) hipLaunchKernelGGL(uavqp::fill_i32_kernel, dim3((n_traj + 255) / 256), dim3(256), 0, ctx->stream, d_first_hit, n_traj, (int32_t)n_samples);
    if (n_samples == 0) return UAVQP_OK;
    uavqp::EllipsoidGridArgs a;
    a.tmax_bits = d_tmax_bits;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.n_samples = n_samples;
    a.seg_offsets = d_seg_offsets; a.times = d_times; a.coeff = d_coeff; a.grid = grid->view;
    a.t0 = t0; a.dt = dt; a.robot_r = robot_r; a.robot_h = robot_h; a.first_hit = d_first_hit; a.flags = d_flags;
    const long long total = (long long)n_traj * n_samples;
    long long grid_dim = (total + 63) / 64;                    // one wave per workgroup (the wave pools its candidates through LDS)
    if (grid_dim > (long long)ctx->num_cus * 64) grid_dim = (long long)ctx->num_cus * 64;
    if (r == 3)
        hipLaunchKernelGGL(uavqp::ellipsoid_grid_kernel<3>, dim3((unsigned)grid_dim), dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(uavqp::ellipsoid_grid_kernel<4>, dim3((unsigned)grid_dim), dim3(64), 0, ctx->stream, a);
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

extern "C" int uavqp_corridor_from_cloud_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                                int n_rows, const double* d_waypoints, const double* d_times, const double* d_coeff,
                                                const double* d_obstacles, int n_obs, double robot_r, double robot_h, double h_max,
                                                double* d_corr_lo, double* d_corr_hi, double* d_clearance) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || n_rows < 0 || uniform_segments < 0 || n_obs < 0 || !(robot_r > 0.0) ||
        !(robot_h > 0.0) || !(h_max >= 0.0))
        return UAVQP_ERR_INVALID_ARG;
    if (n_traj == 0 || n_rows == 0) return UAVQP_OK;
    if (!d_waypoints || !d_corr_lo || !d_corr_hi || (n_obs > 0 && !d_obstacles) || (uniform_segments == 0 && !d_seg_offsets) ||
        (d_coeff && !d_times) || (uniform_segments > 0 && (long long)n_traj * (uniform_segments + 1) != n_rows))
        return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    uavqp::CloudCorridorArgs a;
    a.n_traj = n_traj; a.uniform = uniform_segments; a.n_rows = n_rows; a.n_obs = n_obs;
    a.seg_offsets = d_seg_offsets; a.waypoints = d_waypoints; a.times = d_times; a.coeff = d_coeff; a.obs = d_obstacles;
    a.robot_r = robot_r; a.robot_h = robot_h; a.h_max = h_max; a.lo = d_corr_lo; a.hi = d_corr_hi; a.clearance = d_clearance;
    long long grid = ((long long)n_rows + 255) / 256;
    if (grid > (long long)ctx->num_cus * 16) grid = (long long)ctx->num_cus * 16;
    a.sort = nullptr; a.row_perm = nullptr; a.row_start = nullptr; a.pt_start = nullptr; a.pts_sorted = nullptr; a.reach = 0.0;
    // Large clouds, boxes only: rows and points sorted along the cloud's longest axis, every block scans the points within `reach`
    // of its rows (obstacle_grid.h, cloud_window_kernel) -- identical boxes, fewer pairs.  The clearance output needs the exhaustive min.
    const bool windowed = !d_clearance && n_obs >= 4096 && n_rows >= 4096 && ctx->settings.cloud_window != 0;
#ifdef UAVQP_EXPERIMENTS
    if (windowed && ctx->settings.cloud_window >= 3) {
        // Round 5, two passes on the 2-D cell grid (cloud_grid2d.h): the rows the bounding box does not cull scan the ring next to them; the ones whose
        // clearance so far does not bound what farther points could change are re-sorted by the radius it implies and scan exactly that -- identical boxes.
        const double rmax = robot_r > robot_h ? robot_r : robot_h, rmin = robot_r > robot_h ? robot_h : robot_r;
        a.reach = rmax * (1.0 + 3.0 * h_max / rmin) * (1.0 + 1e-9) + 1e-9;
        const size_t b_cg = 256, b_h = align256(sizeof(int32_t) * (uavqp::CLOUD2D_MAX_CELLS + 2));
        const size_t b_h2 = align256(sizeof(int32_t) * ((size_t)uavqp::CLOUD2D_BUCKETS * uavqp::CLOUD2D_MAX_CELLS + 2));
        const size_t b_pts = align256(sizeof(double) * 3 * (size_t)n_obs), b_perm = align256(sizeof(int32_t) * (size_t)n_rows), b_m = align256(sizeof(double) * (size_t)n_rows);
        const int rc = ensure_ws(ctx, b_cg + 4 * b_h + 2 * b_h2 + b_pts + 4 * b_perm + b_m);
        if (rc != UAVQP_OK) return rc;
        char* p = (char*)ctx->ws;
        uavqp::Cloud2D* d_cg = (uavqp::Cloud2D*)p; p += b_cg;
        int32_t* d_pstart = (int32_t*)p; p += b_h;
        int32_t* d_pcur = (int32_t*)p; p += b_h;
        int32_t* d_rstart = (int32_t*)p; p += b_h;
        int32_t* d_rcur = (int32_t*)p; p += b_h;
        int32_t* d_hist2 = (int32_t*)p; p += b_h2;
        int32_t* d_cur2 = (int32_t*)p; p += b_h2;
        double* d_pts = (double*)p; p += b_pts;
        int32_t* d_perm = (int32_t*)p; p += b_perm;
        int32_t* d_rbin = (int32_t*)p; p += b_perm;
        int32_t* d_rkey = (int32_t*)p; p += b_perm;
        int32_t* d_perm2 = (int32_t*)p; p += b_perm;
        double* d_rm = (double*)p;
        long long sgrid = ((long long)n_obs + n_rows + 511) / 512;
        if (sgrid > (long long)ctx->num_cus * 2) sgrid = (long long)ctx->num_cus * 2;
        UAVQP_HIP(hipMemsetAsync(d_hist2, 0, b_h2, ctx->stream));
        hipLaunchKernelGGL(uavqp::cloud2d_setup_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_obstacles, n_obs, a.reach / 8.0, d_cg, d_pstart, d_rstart);
        if (r == 3) hipLaunchKernelGGL(uavqp::cloud2d_hist_cull_kernel<3>, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, a, (const uavqp::Cloud2D*)d_cg, d_pstart, d_rstart, d_rbin);
        else hipLaunchKernelGGL(uavqp::cloud2d_hist_cull_kernel<4>, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, a, (const uavqp::Cloud2D*)d_cg, d_pstart, d_rstart, d_rbin);
        hipLaunchKernelGGL(uavqp::cloud2d_scan_n_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uavqp::Cloud2D*)d_cg, 1, 1, d_pstart, d_pcur, d_rstart, d_rcur);
        hipLaunchKernelGGL(uavqp::cloud2d_scatter_bin_kernel, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, d_obstacles, n_obs, (const int32_t*)d_rbin, n_rows,
                           (const uavqp::Cloud2D*)d_cg, d_pcur, d_rcur, d_pts, d_perm);
        a.row_perm = d_perm; a.pt_start = d_pstart; a.pts_sorted = d_pts;
        uavqp::Cloud2PArgs pa;
        pa.c = a; pa.grid = d_cg; pa.row_start = d_rstart; pa.row_key = d_rkey; pa.row_m = d_rm; pa.hist2 = d_hist2; pa.perm2 = d_perm2;
        if (r == 3) hipLaunchKernelGGL(uavqp::cloud2d_pass1_kernel<3>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pa);
        else hipLaunchKernelGGL(uavqp::cloud2d_pass1_kernel<4>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pa);
        hipLaunchKernelGGL(uavqp::cloud2d_scan_n_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uavqp::Cloud2D*)d_cg, 0, uavqp::CLOUD2D_BUCKETS, (int32_t*)nullptr,
                           (int32_t*)nullptr, d_hist2, d_cur2);
        long long g2 = ((long long)n_rows + 255) / 256;
        if (g2 > (long long)ctx->num_cus * 8) g2 = (long long)ctx->num_cus * 8;
        hipLaunchKernelGGL(uavqp::cloud2d_scatter2_kernel, dim3((unsigned)g2), dim3(256), 0, ctx->stream, (const int32_t*)d_rkey, n_rows, d_cur2, d_perm2);
        if (r == 3) hipLaunchKernelGGL(uavqp::cloud2d_pass2_kernel<3>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pa);
        else hipLaunchKernelGGL(uavqp::cloud2d_pass2_kernel<4>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pa);
    } else if (windowed && ctx->settings.cloud_window == 2) {
        // Round 5: points and rows sorted by the cell of a 2-D grid; a block scans the cells around its rows nearest first and stops as soon
        // as the clearance found bounds what farther points could still change (cloud_grid2d.h) -- identical boxes, a fraction of the pairs.
        const double rmax = robot_r > robot_h ? robot_r : robot_h, rmin = robot_r > robot_h ? robot_h : robot_r;
        a.reach = rmax * (1.0 + 3.0 * h_max / rmin) * (1.0 + 1e-9) + 1e-9;
        const size_t b_cg = 256, b_h = align256(sizeof(int32_t) * (uavqp::CLOUD2D_MAX_CELLS + 1));
        const size_t b_pts = align256(sizeof(double) * 3 * (size_t)n_obs), b_perm = align256(sizeof(int32_t) * (size_t)n_rows);
        const int rc = ensure_ws(ctx, b_cg + 4 * b_h + b_pts + b_perm);
        if (rc != UAVQP_OK) return rc;
        char* p = (char*)ctx->ws;
        uavqp::Cloud2D* d_cg = (uavqp::Cloud2D*)p;
        unsigned int* d_phase = (unsigned int*)(p + 128); p += b_cg;
        int32_t* d_pstart = (int32_t*)p; p += b_h;
        int32_t* d_pcur = (int32_t*)p; p += b_h;
        int32_t* d_rstart = (int32_t*)p; p += b_h;
        int32_t* d_rcur = (int32_t*)p; p += b_h;
        double* d_pts = (double*)p; p += b_pts;
        int32_t* d_perm = (int32_t*)p;
        long long sgrid = ((long long)n_obs + n_rows + 511) / 512;
        if (sgrid > (long long)ctx->num_cus * 2) sgrid = (long long)ctx->num_cus * 2;
#ifdef UAVQP_CLOUD_STATS
        UAVQP_HIP(hipMemsetAsync(d_phase, 0, 16, ctx->stream));
#endif
        hipLaunchKernelGGL(uavqp::cloud2d_setup_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_obstacles, n_obs, a.reach / 8.0, d_cg, d_pstart, d_rstart);
        hipLaunchKernelGGL(uavqp::cloud2d_hist_kernel, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, d_obstacles, n_obs, d_waypoints, n_rows,
                           (const uavqp::Cloud2D*)d_cg, d_pstart, d_rstart);
        hipLaunchKernelGGL(uavqp::cloud2d_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uavqp::Cloud2D*)d_cg, d_pstart, d_pcur, d_rstart, d_rcur);
        hipLaunchKernelGGL(uavqp::cloud2d_scatter_kernel, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, d_obstacles, n_obs, d_waypoints, n_rows,
                           (const uavqp::Cloud2D*)d_cg, d_pcur, d_rcur, d_pts, d_perm);
        a.row_perm = d_perm; a.pt_start = d_pstart; a.pts_sorted = d_pts;
        uavqp::Cloud2DArgs ga;
        ga.c = a; ga.grid = d_cg; ga.phase_count = nullptr;
#ifdef UAVQP_CLOUD_STATS
        ga.phase_count = d_phase;      // probe build (tools/cloud_phase_probe.py): blocks per ring
        ctx->dbg_queue = d_phase;
#else
        (void)d_phase;
#endif
        if (r == 3)
            hipLaunchKernelGGL(uavqp::cloud_grid2d_kernel<3>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ga);
        else
            hipLaunchKernelGGL(uavqp::cloud_grid2d_kernel<4>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ga);
    } else
#endif
    if (windowed) {
        const double rmax = robot_r > robot_h ? robot_r : robot_h, rmin = robot_r > robot_h ? robot_h : robot_r;
        a.reach = rmax * (1.0 + 3.0 * h_max / rmin) * (1.0 + 1e-9) + 1e-9;
        const size_t b_cs = 256, b_ph = align256(sizeof(int32_t) * (uavqp::CLOUD_PT_BINS + 1)), b_rh = align256(sizeof(int32_t) * (uavqp::CLOUD_ROW_BINS + 2));
        const size_t b_pts = align256(sizeof(double) * 3 * (size_t)n_obs), b_perm = align256(sizeof(int32_t) * (size_t)n_rows);
        const int rc = ensure_ws(ctx, b_cs + 2 * b_ph + 2 * b_rh + b_pts + 2 * b_perm);
        if (rc != UAVQP_OK) return rc;
        char* p = (char*)ctx->ws;
        uavqp::CloudSort* d_cs = (uavqp::CloudSort*)p; p += b_cs;
        int32_t* d_pstart = (int32_t*)p; p += b_ph;
        int32_t* d_pcur = (int32_t*)p; p += b_ph;
        int32_t* d_rstart = (int32_t*)p; p += b_rh;
        int32_t* d_rcur = (int32_t*)p; p += b_rh;
        double* d_pts = (double*)p; p += b_pts;
        int32_t* d_perm = (int32_t*)p; p += b_perm;
        int32_t* d_rbin = (int32_t*)p;
        long long sgrid = ((long long)n_obs + n_rows + 511) / 512;   // every block: one contiguous slice of the keys, bins counted in LDS
        if (sgrid > (long long)ctx->num_cus * 2) sgrid = (long long)ctx->num_cus * 2;
        hipLaunchKernelGGL(uavqp::cloud_sort_setup_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_obstacles, n_obs, a.reach, d_cs, d_pstart, d_rstart);
        // (the histogram pass also computes every row's attitude: rows the cloud's bounding box proves capped go to an extra bin and are never scanned)
        if (r == 3) hipLaunchKernelGGL(uavqp::cloud_sort_hist_kernel<3>, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, a, (const uavqp::CloudSort*)d_cs, d_pstart, d_rstart, d_rbin);
        else hipLaunchKernelGGL(uavqp::cloud_sort_hist_kernel<4>, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, a, (const uavqp::CloudSort*)d_cs, d_pstart, d_rstart, d_rbin);
        hipLaunchKernelGGL(uavqp::cloud_sort_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_pstart, d_pcur, d_rstart, d_rcur);
        hipLaunchKernelGGL(uavqp::cloud_sort_scatter_kernel, dim3((unsigned)sgrid), dim3(256), 0, ctx->stream, d_obstacles, n_obs, (const int32_t*)d_rbin, n_rows,
                           (const uavqp::CloudSort*)d_cs, d_pcur, d_rcur, d_pts, d_perm);
        a.sort = d_cs; a.row_perm = d_perm; a.row_start = d_rstart; a.pt_start = d_pstart; a.pts_sorted = d_pts;
        long long wgrid = ((long long)n_rows + 127) / 128;        // 128 rows per block: two waves share a row and split the points (four: no further gain)
        if (wgrid > (long long)ctx->num_cus * 24) wgrid = (long long)ctx->num_cus * 24;
        if (r == 3)
            hipLaunchKernelGGL((uavqp::cloud_window_kernel<3, 2>), dim3((unsigned)wgrid), dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((uavqp::cloud_window_kernel<4, 2>), dim3((unsigned)wgrid), dim3(256), 0, ctx->stream, a);
    } else if (r == 3)
        hipLaunchKernelGGL(uavqp::cloud_corridor_kernel<3>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(uavqp::cloud_corridor_kernel<4>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    UAVQP_HIP(hipGetLastError());
    return UAVQP_OK;
}

// ===================================================================================================
// BASELINE config 5 as one call
// ===================================================================================================
#include "uavqp_pipeline.h"

// ===================================================================================================
// N3: quadrotor_msgs/PolynomialTrajectory packer (host only)
// ===================================================================================================
extern "C" int uavqp_pack_polynomial_trajectory(int r, int n_seg, const double* coeff_traj, const double* times, double* coef_x,
                                                double* coef_y, double* coef_z, double* time_out, uint32_t* order_out,
                                                uint32_t* num_order_out, uint32_t* num_segment_out) {
    if ((r != 3 && r != 4) || n_seg < 1 || !coeff_traj || !times || !coef_x || !coef_y || !coef_z || !time_out) return UAVQP_ERR_INVALID_ARG;
    for (int i = 0; i < n_seg; ++i)
        if (!(times[i] > 0.0) || !(times[i] < INFINITY)) return UAVQP_ERR_INVALID_ARG;
    const size_t per_axis = (size_t)n_seg * 2 * r;
    // each axis slice of the solver output IS the message's per-axis array: coef[i * (num_order + 1) + j] multiplies t^j in segment i
    // (poly_traj_server.cpp:68-78 reads exactly that index)
    std::memcpy(coef_x, coeff_traj, sizeof(double) * per_axis);
    std::memcpy(coef_y, coeff_traj + per_axis, sizeof(double) * per_axis);
    std::memcpy(coef_z, coeff_traj + 2 * per_axis, sizeof(double) * per_axis);
    std::memcpy(time_out, times, sizeof(double) * (size_t)n_seg);
    if (order_out)
        for (int i = 0; i < n_seg; ++i) order_out[i] = (uint32_t)(2 * r - 1);
    if (num_order_out) *num_order_out = (uint32_t)(2 * r - 1);
    if (num_segment_out) *num_segment_out = (uint32_t)n_seg;
    return UAVQP_OK;
}

// ===================================================================================================
// Multi-GPU entry points (uavqp_comm.h)
// ===================================================================================================
extern "C" int uavqp_shard_bounds(int n_traj, int world, int32_t* bounds) {
    if (n_traj < 0 || world < 1 || !bounds) return UAVQP_ERR_INVALID_ARG;
    for (int g = 0; g <= world; ++g) bounds[g] = (int32_t)(((long long)n_traj * g) / world);
    return UAVQP_OK;
}

extern "C" int uavqp_shard_bounds_ragged(const int32_t* seg_offsets, int n_traj, int world, int32_t* bounds) {
    if (n_traj < 0 || world < 1 || !bounds || !seg_offsets) return UAVQP_ERR_INVALID_ARG;
    for (int b = 0; b < n_traj; ++b)
        if (seg_offsets[b + 1] < seg_offsets[b]) return UAVQP_ERR_INVALID_ARG;
    const long long total = (long long)seg_offsets[n_traj] - seg_offsets[0];
    bounds[0] = 0;
    for (int g = 1; g < world; ++g) {
        // first trajectory whose offset reaches g / world of the segments (exact integer comparison: off * world >= total * g)
        const long long want = total * g;
        const int32_t* lo = seg_offsets + bounds[g - 1];
        const int32_t* hi = seg_offsets + n_traj;
        const int32_t* it = std::lower_bound(lo, hi, want, [&](int32_t off, long long w) { return ((long long)off - seg_offsets[0]) * world < w; });
        bounds[g] = (int32_t)(it - seg_offsets);
    }
    bounds[world] = n_traj;
    return UAVQP_OK;
}

#define UAVQP_RCCL(expr)                                                                             \
    do {                                                                                             \
        ncclResult_t r_ = (expr);                                                                    \
        if (r_ != ncclSuccess) {                                                                     \
            g_last_error = std::string(#expr) + ": " + uavqp::rccl().GetErrorString(r_);             \
            return UAVQP_ERR_RCCL;                                                                   \
        }                                                                                            \
    } while (0)

static int rccl_ready() {
    if (!uavqp::rccl().load()) {
        g_last_error = uavqp::rccl().error;
        return UAVQP_ERR_RCCL;
    }
    return UAVQP_OK;
}

extern "C" int uavqp_comm_unique_id(void* id_out) {
    static_assert(UAVQP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "token size follows RCCL");
    if (!id_out) return UAVQP_ERR_INVALID_ARG;
    int rc = rccl_ready();
    if (rc != UAVQP_OK) return rc;
    ncclUniqueId id;
    UAVQP_RCCL(uavqp::rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return UAVQP_OK;
}

extern "C" int uavqp_comm_create(uavqp_ctx* ctx, int rank, int world, const void* unique_id) {
    if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world) return UAVQP_ERR_INVALID_ARG;
    if (ctx->comm.comm) { g_last_error = "this ctx already owns a communicator"; return UAVQP_ERR_INVALID_ARG; }
    int rc = rccl_ready();
    if (rc != UAVQP_OK) return rc;
    UAVQP_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    UAVQP_RCCL(uavqp::rccl().CommInitRank(&comm, world, id, rank));
    ctx->comm.comm = comm;
    ctx->comm.rank = rank;
    ctx->comm.world = world;
    return UAVQP_OK;
}

extern "C" int uavqp_comm_info(const uavqp_ctx* ctx, int32_t* rank_out, int32_t* world_out) {
    if (!ctx || (!rank_out && !world_out)) return UAVQP_ERR_INVALID_ARG;
    if (!ctx->comm.comm) { g_last_error = "no communicator: call uavqp_comm_create first"; return UAVQP_ERR_INVALID_ARG; }
    // asked of RCCL, not remembered: ncclCommUserRank / ncclCommCount of the communicator the ctx owns
    int rk = -1, wd = -1;
    UAVQP_RCCL(uavqp::rccl().CommUserRank(ctx->comm.comm, &rk));
    UAVQP_RCCL(uavqp::rccl().CommCount(ctx->comm.comm, &wd));
    if (rank_out) *rank_out = rk;
    if (world_out) *world_out = wd;
    return UAVQP_OK;
}

extern "C" int uavqp_comm_destroy(uavqp_ctx* ctx) {
    if (!ctx) return UAVQP_ERR_INVALID_ARG;
    if (!ctx->comm.comm) return UAVQP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ncclComm_t c = ctx->comm.comm;
    ctx->comm = uavqp::Comm();
    UAVQP_RCCL(uavqp::rccl().CommDestroy(c));
    return UAVQP_OK;
}

template <typename T>
static int allgather_shards(uavqp_ctx* ctx, const T* d_local, const int64_t* counts, T* d_full, ncclDataType_t dt) {
    if (!ctx || !counts || !d_full) return UAVQP_ERR_INVALID_ARG;
    if (!ctx->comm.comm) { g_last_error = "no communicator: call uavqp_comm_create first"; return UAVQP_ERR_INVALID_ARG; }
    const int world = ctx->comm.world, rank = ctx->comm.rank;
    bool equal = true;
    int64_t total = 0;
    for (int g = 0; g < world; ++g) {
        if (counts[g] < 0) return UAVQP_ERR_INVALID_ARG;
        equal = equal && counts[g] == counts[0];
        total += counts[g];
    }
    if (total == 0) return UAVQP_OK;
    if (counts[rank] > 0 && !d_local) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    uavqp::RcclApi& R = uavqp::rccl();
    if (equal) {
        UAVQP_RCCL(R.AllGather(d_local, d_full, (size_t)counts[0], dt, ctx->comm.comm, ctx->stream));
        return UAVQP_OK;
    }
    // all-gather-v: every rank sends its shard to every peer directly, one hop per peer on the xGMI full mesh
    std::vector<int64_t> off(world + 1, 0);
    for (int g = 0; g < world; ++g) off[g + 1] = off[g] + counts[g];
    UAVQP_RCCL(R.GroupStart());
    ncclResult_t first_err = ncclSuccess;
    const char* what = "";
    for (int g = 0; g < world && first_err == ncclSuccess; ++g) {
        if (g == rank) continue;
        if (counts[rank] > 0) {
            first_err = R.Send(d_local, (size_t)counts[rank], dt, g, ctx->comm.comm, ctx->stream);
            what = "ncclSend";
        }
        if (first_err == ncclSuccess && counts[g] > 0) {
            first_err = R.Recv(d_full + off[g], (size_t)counts[g], dt, g, ctx->comm.comm, ctx->stream);
            what = "ncclRecv";
        }
    }
    // the group is closed on EVERY path: an open group would swallow the next collective of this thread
    const ncclResult_t end_err = R.GroupEnd();
    if (first_err != ncclSuccess) {
        g_last_error = std::string(what) + ": " + R.GetErrorString(first_err);
        return UAVQP_ERR_RCCL;
    }
    if (end_err != ncclSuccess) {
        g_last_error = std::string("ncclGroupEnd: ") + R.GetErrorString(end_err);
        return UAVQP_ERR_RCCL;
    }
    if (counts[rank] > 0 && d_local != d_full + off[rank])
        UAVQP_HIP(hipMemcpyAsync(d_full + off[rank], d_local, sizeof(T) * (size_t)counts[rank], hipMemcpyDeviceToDevice, ctx->stream));
    return UAVQP_OK;
}

extern "C" int uavqp_allgather_coeffs(uavqp_ctx* ctx, const double* d_local, const int64_t* counts, double* d_full) {
    return allgather_shards<double>(ctx, d_local, counts, d_full, ncclFloat64);
}

extern "C" int uavqp_allgather_status(uavqp_ctx* ctx, const int32_t* d_local, const int64_t* counts, int32_t* d_full) {
    return allgather_shards<int32_t>(ctx, d_local, counts, d_full, ncclInt32);
}
