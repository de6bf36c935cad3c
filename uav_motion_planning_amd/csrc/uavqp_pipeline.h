// uavqp_pipeline.h -- BASELINE config 5 as ONE C-ABI call (included at the end of uavqp.hip): host-side sequencing of the entry
// points of include/uavqp.h plus the few small kernels the sequencing needs (counters for the loop control, box repair).
//
//   plain solve -> corridor boxes from the obstacle cloud (robot ellipsoid of KinoAstar::isCollisionFree, kino_astar.cpp:721-758,
//   attitude of that solve) -> at most max_rounds x (corridor-constrained solve, working set carried from round to round + time
//   re-allocation) -> SE(3) collision check of the result against a uniform grid over the cloud -> REPAIR: the boxes only bound the
//   knots and carry the attitude of the first solve, so the check of the final polynomials is the arbiter: the boxes of every
//   trajectory it flags are halved towards the searcher's waypoints (last round: collapsed onto them = the reference's equality
//   problem) and the batch is re-solved, at most repair_rounds times.
//
// The reference has no such loop (constant 1.0 s per segment, test_minimum_jerk.cpp:65-71; every row an equality,
// minimum_control.cpp:98-125): nothing to mirror, parity is per inner solve (SURVEY.md section 8-a').
// Loop control is data dependent (did any duration change? does anything still collide?).  Per outer round the host needs ONE number:
// the compaction kernel writes it, tagged with the round's sequence number, into a word of pinned host-coherent memory that the host
// polls -- no copy command, no event, no stream stop between rounds.  What the check finds goes into a 64-byte counter block that is
// copied to the pinned page with the call's one stream synchronisation; everything else stays in device buffers.
#pragma once

namespace uavqp {

struct PipeCounters {
    unsigned int changed;        // trajectories whose durations the last re-allocation stretched
    unsigned int hit;            // trajectories the check flags
    unsigned int hit_blocked;    // ... of those: with an interior waypoint the cloud leaves no room around (degenerate box)
    unsigned int hit_repairable; // ... the rest (what a repair round works on)
    unsigned int unsolved;       // trajectories whose status is not UAVQP_SOLVED
    unsigned int pad_[3];
    unsigned long long tmax_bits;  // max over trajectories of the total duration (bits of a non-negative double: ordered like integers)
    unsigned long long pad2_[3];
};
static_assert(sizeof(PipeCounters) == 64, "counter block is one 64-byte record");

__global__ void pipe_zero_kernel(PipeCounters* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c = PipeCounters{};
}

// head of a pipeline call: the stretch record starts at 1 (optional), the counter block at zero (nothing touches it before the check)
__global__ __launch_bounds__(256) void pipe_begin_kernel(double* __restrict__ scale, int n, PipeCounters* c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (scale && i < n) scale[i] = 1.0;
    if (i == 0) *c = PipeCounters{};
}

// changed[b] > 0 / status[b] != SOLVED counts (either pointer may be null)
__global__ __launch_bounds__(256) void pipe_count_kernel(const int32_t* __restrict__ changed, const int32_t* __restrict__ status, int n, PipeCounters* c) {
    int nc = 0, nu = 0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) {
        if (changed && changed[b] > 0) ++nc;
        if (status && status[b] != UAVQP_SOLVED) ++nu;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nc += __shfl_xor(nc, d, 64);
        nu += __shfl_xor(nu, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nc) atomicAdd(&c->changed, (unsigned)nc);
        if (nu) atomicAdd(&c->unsolved, (unsigned)nu);
    }
}

// What the check needs per trajectory, in one launch, eight lanes per trajectory (sub-lane j takes segments j, j + 8, ...; three
// xor-shuffles combine): the total duration -> max over the batch (all trajectories are sampled on ONE grid: dt = max total /
// (samples - 1), which the check kernel reads from the counter block -- the host does not wait for it), first_hit = "none", and on the
// first pass roomy[b] (null afterwards: a repair has shrunk the boxes since).  Was: three kernels, a copy and a stream synchronisation.
__global__ __launch_bounds__(256) void pipe_check_prep_kernel(int n, int uniform, const int32_t* __restrict__ seg_offsets, const double* __restrict__ times,
                                                              const double* __restrict__ lo, const double* __restrict__ hi, uint8_t* __restrict__ roomy,
                                                              int32_t* __restrict__ first_hit, int n_samples, PipeCounters* c) {
    constexpr int LPT = 8;
    const int sub = threadIdx.x % LPT;
    const long long n_lanes = (long long)n * LPT, stride = (long long)gridDim.x * 256;
    const long long n_round = (n_lanes + stride - 1) / stride * stride;
    double mx = 0.0;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n_round; g += stride) {
        const bool live = g < n_lanes;
        const int b = live ? (int)(g / LPT) : 0;
        int s0 = 0, M = 0;
        if (live) {
            if (uniform > 0) { M = uniform; s0 = b * M; } else { s0 = seg_offsets[b]; M = seg_offsets[b + 1] - s0; }
        }
        // (the partial sums of the eight sub-lanes are added in a fixed tree: the same bits on every run; dt only places the samples)
        double t = 0.0;
        for (int i = sub; i < M; i += LPT) t += times[s0 + i];
        int ok = 1;
        if (roomy) {
            const size_t row0 = (size_t)s0 + b;
            for (int i = sub; i < M; i += LPT) {
                if (i == 0) continue;            // interior waypoint rows only
                const double* l = lo + 3 * (row0 + i);
                const double* h = hi + 3 * (row0 + i);
                const double w = fmin(fmin(h[0] - l[0], h[1] - l[1]), h[2] - l[2]);
                ok &= (w > 0.0) ? 1 : 0;
            }
        }
#pragma unroll
        for (int d = 1; d < LPT; d <<= 1) {
            t += __shfl_xor(t, d, 64);
            ok &= __shfl_xor(ok, d, 64);
        }
        if (live && sub == 0) {
            if (roomy) roomy[b] = (uint8_t)ok;
            first_hit[b] = n_samples;
            if (t > mx && t < INFINITY) mx = t;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
    // one atomic per workgroup (2048 waves on one address cost more than the rest of the kernel)
    __shared__ double s_mx[4];
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3]));
        if (mx > 0.0) atomicMax(&c->tmax_bits, (unsigned long long)__double_as_longlong(mx));
    }
}

// flag[b] = trajectory b collides (first_hit < n_samples) AND is roomy; counts for the host -- with the number of trajectories whose
// status is not UAVQP_SOLVED (status may be null): when no repair follows, this block is the call's summary
__global__ __launch_bounds__(256) void pipe_hits_kernel(int n, const int32_t* __restrict__ first_hit, int n_samples, const uint8_t* __restrict__ roomy,
                                                        uint8_t* __restrict__ flag, const int32_t* __restrict__ status, PipeCounters* c) {
    int nh = 0, nb = 0, nr = 0, nu = 0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) {
        const bool hit = first_hit[b] < n_samples;
        const bool rm = roomy[b] != 0;
        flag[b] = (hit && rm) ? 1 : 0;
        nh += hit;
        nb += hit && !rm;
        nr += hit && rm;
        if (status && status[b] != UAVQP_SOLVED) ++nu;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nh += __shfl_xor(nh, d, 64);
        nb += __shfl_xor(nb, d, 64);
        nr += __shfl_xor(nr, d, 64);
        nu += __shfl_xor(nu, d, 64);
    }
    __shared__ int s_n[4][4];
    if ((threadIdx.x & 63) == 0) {
        int* q = s_n[threadIdx.x >> 6];
        q[0] = nh; q[1] = nb; q[2] = nr; q[3] = nu;
    }
    __syncthreads();
    if (threadIdx.x < 4) {      // one atomic per counter and workgroup
        const int k = threadIdx.x, v = s_n[0][k] + s_n[1][k] + s_n[2][k] + s_n[3][k];
        unsigned int* dst = k == 0 ? &c->hit : k == 1 ? &c->hit_blocked : k == 2 ? &c->hit_repairable : &c->unsolved;
        if (v) atomicAdd(dst, (unsigned)v);
    }
}

// boxes of the flagged trajectories move towards their waypoints: lo <- w - shrink (w - lo), hi <- w + shrink (hi - w)
// (shrink = 0.5 halves them, 0 collapses them onto the waypoint: the reference's equality row).  One wave per trajectory.
__global__ __launch_bounds__(64) void pipe_shrink_kernel(int n, int uniform, const int32_t* __restrict__ seg_offsets, const double* __restrict__ wp,
                                                         double* __restrict__ lo, double* __restrict__ hi, const uint8_t* __restrict__ flag, double shrink) {
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        if (!flag[b]) continue;
        int s0, M;
        if (uniform > 0) { M = uniform; s0 = b * M; } else { s0 = seg_offsets[b]; M = seg_offsets[b + 1] - s0; }
        const size_t e0 = 3 * ((size_t)s0 + b);
        for (int i = threadIdx.x; i < 3 * (M + 1); i += 64) {
            const double w = wp[e0 + i];
            lo[e0 + i] = w - shrink * (w - lo[e0 + i]);
            hi[e0 + i] = w + shrink * (hi[e0 + i] - w);
        }
    }
}

}  // namespace uavqp

static int ensure_pipe_ws(uavqp_ctx* ctx, size_t bytes) {
    // counter block [0] + the ring of per-round words [at 256 bytes]; host-coherent: the device writes a round's word while the stream runs on
    if (!ctx->h_pipe) {
        UAVQP_HIP(hipHostMalloc(&ctx->h_pipe, 1024, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(ctx->h_pipe, 0, 1024);
    }
    if (bytes <= ctx->pipe_bytes) return UAVQP_OK;
    UAVQP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_pipe) UAVQP_HIP(hipFree(ctx->d_pipe));
    ctx->d_pipe = nullptr;
    ctx->pipe_bytes = 0;
    UAVQP_HIP(hipMalloc(&ctx->d_pipe, bytes));
    ctx->pipe_bytes = bytes;
    return UAVQP_OK;
}

extern "C" void uavqp_default_pipeline_params(uavqp_pipeline_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(uavqp_pipeline_params);
    p->robot_r = 0.4;        // test_kino_astar_searching.launch:56
    p->robot_h = 0.1;        // :57
    p->h_max = 0.8;          // corridor half-widths h ~ U(0.3, 0.8) m
    p->v_max = 7.0;          // max_velocity, test_kino_astar_searching.launch:49
    p->a_max = 10.0;         // max_accelration, :50
    p->max_rounds = 5;       // BASELINE config 5: "an outer loop of <= 5 time re-allocations"
    p->samples_per_seg = 16;
    p->max_stretch = 2.0;
    p->check_samples = 100;
    p->repair_rounds = 2;
    p->check_robot_r = 0.0;  // <= 0: the ellipsoid the boxes were built with
    p->check_robot_h = 0.0;
}

extern "C" int uavqp_corridor_pipeline_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments, int total_segments,
                                              const int32_t* d_seg_offsets, const double* d_waypoints, double* d_times, const double* d_bc,
                                              const double* d_obstacles, int n_obs, const uavqp_grid* grid,
                                              const uavqp_pipeline_params* params, double* d_coeff_out, int32_t* d_status_out,
                                              double* d_corr_lo, double* d_corr_hi, int32_t* d_first_hit, uavqp_pipeline_result* result) {
    if (!ctx || !params || params->struct_size != (int32_t)sizeof(uavqp_pipeline_params) || (r != 3 && r != 4) || n_traj < 0 ||
        uniform_segments < 0 || n_obs < 0 || total_segments < 0)
        return UAVQP_ERR_INVALID_ARG;
    const uavqp_pipeline_params P = *params;
    if (!(P.robot_r > 0.0) || !(P.robot_h > 0.0) || !(P.h_max >= 0.0) || !(P.v_max > 0.0) || !(P.a_max > 0.0) || P.max_rounds < 1 ||
        P.samples_per_seg < 1 || !(P.max_stretch > 1.0) || P.check_samples < 0 || P.check_samples == 1 || P.repair_rounds < 0)
        return UAVQP_ERR_INVALID_ARG;
    if (result) *result = uavqp_pipeline_result{};
    if (n_traj == 0) return UAVQP_OK;
    if (!d_waypoints || !d_times || !d_bc || !d_coeff_out || !d_status_out || !d_corr_lo || !d_corr_hi || (n_obs > 0 && !d_obstacles) ||
        (uniform_segments == 0 && (!d_seg_offsets || max_segments < 1)) ||
        (uniform_segments > 0 && (long long)uniform_segments * n_traj != total_segments))
        return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int n = n_traj, uni = uniform_segments, mx = uni > 0 ? uni : max_segments;
    // every workspace below is sized from the caller's total_segments: it is checked against the last CSR offset on the device
    // before anything that depends on it is launched.  The word travels through the host-coherent page (await_word below) while the
    // kernels that only look at seg_offsets[0..n] -- the dealing order, the counter reset -- already run.
    if (!ctx->h_pipe) { const int rc0 = ensure_pipe_ws(ctx, 256); if (rc0 != UAVQP_OK) return rc0; }
    volatile unsigned long long* const h_words = (volatile unsigned long long*)((char*)ctx->h_pipe + 256);   // [0..3] round counts, [4] this check
    unsigned long long* d_words = nullptr;
    UAVQP_HIP(hipHostGetDevicePointer((void**)&d_words, (void*)h_words, 0));
    auto await_word = [&](int slot, unsigned int want, unsigned int* value) -> int {
        return await_host_word(s, h_words + slot, want, value, "uavqp_corridor_pipeline_device");
    };
    unsigned int probe_seq = 0u;
    if (uni == 0) hipLaunchKernelGGL(uavqp::host_word_kernel, dim3(1), dim3(1), 0, s, (const int32_t*)(d_seg_offsets + n_traj), (volatile unsigned long long*)(d_words + 4), probe_seq = ++ctx->pipe_seq);
    const int n_rows = total_segments + n_traj;
    const bool checking = P.check_samples > 0;

    // scratch: [counters 64 B][iters n][changed n][first_hit n][active n x 6 x 8][roomy n][flag n][dealing order of the corridor solves]
    const size_t o_it = 256, o_ch = o_it + align256(sizeof(int32_t) * (size_t)n), o_fh = o_ch + align256(sizeof(int32_t) * (size_t)n);
    const size_t o_as = o_fh + align256(sizeof(int32_t) * (size_t)n), o_rm = o_as + align256(sizeof(uint64_t) * 6 * (size_t)n);
    const size_t o_fl = o_rm + align256((size_t)n), o_or = o_fl + align256((size_t)n);
    const bool deal_by_length = uni == 0 && n >= 64 && ctx->settings.ragged_window_sort;
    // G of the dual prelude across the rounds (the re-allocation multiplies all durations of a trajectory by one factor: G only rescales)
    const bool g_across = ctx->settings.corridor_initial_guess == 2 && mx - 1 <= 24 && mx >= 2;
    const size_t o_cp = o_or + (deal_by_length ? length_order_bytes(n) : 0);          // compacted dealing order of the next re-solve + its count
    const int cblocks = (n + uavqp::COMPACT_BLOCK - 1) / uavqp::COMPACT_BLOCK;           // workgroups of a compaction (more than one: two launches)
    if (cblocks > 1024) { g_last_error = "uavqp_corridor_pipeline_device: more than 16 777 216 trajectories in one call"; return UAVQP_ERR_INVALID_ARG; }
    // (two lists: round k compacts round k - 1's into the other one; then their two counts and the per-workgroup counts of a large compaction)
    const size_t o_sc = o_cp + 2 * align256(sizeof(int32_t) * (size_t)n) + 256 + align256(sizeof(int) * (size_t)cblocks);
    const size_t o_gc = o_sc + (g_across ? align256(sizeof(double) * (size_t)n) : 0);
    const size_t need = o_gc + (g_across ? align256(sizeof(double) * (size_t)n * uavqp::corridor_gcache_stride) : 0);
    int rc = ensure_pipe_ws(ctx, need);
    if (rc != UAVQP_OK) return rc;
    char* base = (char*)ctx->d_pipe;
    uavqp::PipeCounters* d_cnt = (uavqp::PipeCounters*)base;
    int32_t* d_iters = (int32_t*)(base + o_it);
    int32_t* d_changed = (int32_t*)(base + o_ch);
    int32_t* d_fh = d_first_hit ? d_first_hit : (int32_t*)(base + o_fh);
    uint64_t* d_active = (uint64_t*)(base + o_as);
    int32_t* const d_cp2[2] = {(int32_t*)(base + o_cp), (int32_t*)(base + o_cp + align256(sizeof(int32_t) * (size_t)n))};
    int* const d_na2[2] = {(int*)(base + o_cp + 2 * align256(sizeof(int32_t) * (size_t)n)), (int*)(base + o_cp + 2 * align256(sizeof(int32_t) * (size_t)n)) + 32};
    int* const d_cblk = (int*)(base + o_cp + 2 * align256(sizeof(int32_t) * (size_t)n) + 256);
    int cur_list = 0;                                                                  // the list the next re-solve takes
    double* d_scale = g_across ? (double*)(base + o_sc) : nullptr;     // factor by which every trajectory was stretched since its G was stored
    double* d_gcache = g_across ? (double*)(base + o_gc) : nullptr;
    int solves_done = 0;
    uint8_t* d_roomy = (uint8_t*)(base + o_rm);
    uint8_t* d_flag = (uint8_t*)(base + o_fl);
    uavqp::PipeCounters* h_cnt = (uavqp::PipeCounters*)ctx->h_pipe;
    int cgrid = (n + 255) / 256;
    if (cgrid > ctx->num_cus * 8) cgrid = ctx->num_cus * 8;

    auto read_counters = [&]() -> int {   // device counter block -> pinned page, then the one synchronisation of this round
        UAVQP_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(uavqp::PipeCounters), hipMemcpyDeviceToHost, s));
        UAVQP_HIP(hipStreamSynchronize(s));
        return UAVQP_OK;
    };
    // (warm = 2: d_coeff_out still holds the previous round's polynomials -- their knot positions are the re-solve's starting point)
    // every corridor solve of the call deals the same trajectories: their order by segment count is made once
    const int32_t* d_order = nullptr;
    if (deal_by_length) {
        rc = make_length_order(ctx, d_seg_offsets, n, base + o_or, &d_order);
        if (rc != UAVQP_OK) return rc;
    }
    // Re-solves take only the trajectories whose problem changed since their last solve -- durations stretched by the re-allocation
    // (d_changed) or boxes shrunk by a repair (d_flag): the others would reproduce their coefficients bit for bit.  With the default
    // cold start (uavqp_settings.corridor_initial_guess = 2: the working set of the position-space dual method, verified by one block
    // solve) a re-solve is a cold solve; with the other settings it is warm-started from the previous round as before.
    const bool cold_rounds = ctx->settings.corridor_initial_guess == 2 && (uni > 0 ? uni : mx) - 1 <= 32;
    auto corridor_solve = [&](int warm, const int32_t* only_i32 = nullptr, const unsigned char* only_u8 = nullptr, bool precompacted = false) {
        const int32_t* d_cp = d_cp2[cur_list];
        const int* d_na = d_na2[cur_list];
        // first solve: every trajectory takes part, its G is stored; later solves load it and rescale by the stretch since then
        const int gmode = (g_across && cold_rounds) ? (solves_done == 0 ? 1 : 2) : 0;
        ++solves_done;
        return corridor_warm_impl(ctx, r, n, uni, mx, d_seg_offsets, d_waypoints, d_times, d_bc, d_corr_lo, d_corr_hi, d_coeff_out,
                                  d_status_out, d_iters, d_active, cold_rounds ? 0 : warm, total_segments, d_order, only_i32, only_u8,
                                  d_gcache, d_scale, gmode, precompacted ? d_cp : nullptr, precompacted ? d_na : nullptr);
    };
    auto reallocate = [&]() -> int {      // + count of the trajectories it stretched
        int rc_ = time_reallocate_impl(ctx, r, n, uni, d_seg_offsets, d_times, d_coeff_out, P.v_max, P.a_max, P.samples_per_seg,
                                       P.max_stretch, d_changed, d_scale);
        if (rc_ != UAVQP_OK) return rc_;
        hipLaunchKernelGGL(uavqp::pipe_zero_kernel, dim3(1), dim3(64), 0, s, d_cnt);
        hipLaunchKernelGGL(uavqp::pipe_count_kernel, dim3(cgrid), dim3(256), 0, s, (const int32_t*)d_changed, (const int32_t*)nullptr, n, d_cnt);
        return read_counters();
    };

    hipLaunchKernelGGL(uavqp::pipe_begin_kernel, dim3(d_scale ? (n + 255) / 256 : 1), dim3(256), 0, s, d_scale, n, d_cnt);
    if (uni == 0) {
        unsigned int last_offset = 0u;
        rc = await_word(4, probe_seq, &last_offset);
        if (rc != UAVQP_OK) return rc;
        if ((int32_t)last_offset != total_segments) {
            g_last_error = "uavqp_corridor_pipeline_device: total_segments does not match seg_offsets[n_traj]";
            (void)hipStreamSynchronize(s);
            return UAVQP_ERR_INVALID_ARG;
        }
    }
    // 1. the reference's equality problem, 2. boxes from the cloud with the attitude of that solve
    rc = uavqp_solve_batch_device(ctx, r, n, uni, mx, d_seg_offsets, d_waypoints, d_times, d_bc, d_coeff_out, d_status_out);
    if (rc != UAVQP_OK) return rc;
    rc = uavqp_corridor_from_cloud_device(ctx, r, n, uni, d_seg_offsets, n_rows, d_waypoints, d_times, d_coeff_out, d_obstacles, n_obs,
                                          P.robot_r, P.robot_h, P.h_max, d_corr_lo, d_corr_hi, nullptr);
    if (rc != UAVQP_OK) return rc;
    // 3. outer loop.  The host only needs a round's count to know whether ANOTHER round is due, so it does not stop the stream for it:
    // the count of round k arrives in its own word of the host-coherent page (below), and round 1 -- later rounds while the last examined
    // one still stretched many trajectories (> n / 64) -- is enqueued before the previous count is looked at: if that one then reports
    // zero, the extra round re-solves nobody and stretches nobody (same bytes everywhere), at the price of a few empty launches.  Was
    // (round 3): copy + stream synchronisation + 25-30 us of idle device per round.
    int rounds = 0, still = 0;
    bool cap_solve_enqueued = false;
    {
        // A round's count reaches the host through a word of pinned, host-coherent memory that the compaction kernel writes itself, tagged
        // with the round's sequence number: the host polls the word (was: a 4-byte copy + an event per round -- a copy kernel and ~6 us of
        // idle device behind the event packet, per round).
        unsigned long long* const d_ring = d_words;
        unsigned int seq_of[4] = {0u, 0u, 0u, 0u};
        auto enqueue_round = [&](int rnd) -> int {
            int rc_ = corridor_solve(rnd > 0 ? 2 : 0, rnd > 0 ? (const int32_t*)d_changed : nullptr, nullptr, rnd > 0);
            if (rc_ != UAVQP_OK) return rc_;
            // rounds after the first look only at the trajectories that round re-solved (the list of the previous compaction): the others
            // have the coefficients and durations the previous re-allocation already accepted (their flag in d_changed is 0 and stays 0)
            const int32_t* d_prev = rnd > 0 ? d_cp2[cur_list] : nullptr;
            const int* d_nprev = rnd > 0 ? d_na2[cur_list] : nullptr;
            rc_ = time_reallocate_impl(ctx, r, n, uni, d_seg_offsets, d_times, d_coeff_out, P.v_max, P.a_max, P.samples_per_seg,
                                       P.max_stretch, d_changed, d_scale, d_prev, d_nprev);
            if (rc_ != UAVQP_OK) return rc_;
            // the dealing order of the trajectories it stretched, for the re-solve of the next round -- and their number, which is the
            // round's counter (was: a zeroing kernel, a counting kernel, and the compaction at the head of the next solve)
            const int nxt = rnd > 0 ? cur_list ^ 1 : cur_list;
            seq_of[rnd & 3] = ++ctx->pipe_seq;
            if (cblocks == 1)
                hipLaunchKernelGGL(uavqp::compact_order_kernel, dim3(1), dim3(1024), 0, s, rnd > 0 ? d_prev : d_order, n, (const int32_t*)d_changed,
                                   (const unsigned char*)nullptr, d_cp2[nxt], d_na2[nxt], (const unsigned int*)nullptr, d_nprev,
                                   (volatile unsigned long long*)&d_ring[rnd & 3], seq_of[rnd & 3]);
            else
                for (int phase = 0; phase < 2; ++phase)
                    hipLaunchKernelGGL(uavqp::compact_order_blocks_kernel, dim3(cblocks), dim3(1024), 0, s, rnd > 0 ? d_prev : d_order, n,
                                       (const int32_t*)d_changed, (const unsigned char*)nullptr, d_cp2[nxt], d_na2[nxt], d_nprev, d_cblk, phase,
                                       (volatile unsigned long long*)&d_ring[rnd & 3], seq_of[rnd & 3]);
            cur_list = nxt;
            if (rnd + 1 == P.max_rounds) {
                // the cap: the trajectories this last re-allocation stretched need one more solve so that their coefficients belong to
                // d_times -- enqueued behind the compaction without waiting for its count (an empty list solves nobody)
                rc_ = corridor_solve(2, (const int32_t*)d_changed, nullptr, true);
                if (rc_ != UAVQP_OK) return rc_;
                cap_solve_enqueued = true;
            }
            return UAVQP_OK;
        };
        int enq = 0, exam = 0;
        unsigned int last = 0;
        for (;;) {
            // (round 1 is enqueued unseen too: a batch that comes here for a re-allocation normally has something to stretch in round 0)
            while (enq < P.max_rounds && (enq <= exam || (enq == exam + 1 && (exam == 0 || (long long)last * 64 > (long long)n)))) {
                rc = enqueue_round(enq);
                if (rc != UAVQP_OK) return rc;
                ++enq;
            }
            if (exam >= enq) break;
            rc = await_word(exam & 3, seq_of[exam & 3], &last);
            if (rc != UAVQP_OK) return rc;
            ++exam;
            ++rounds;
            still = (int)last;
            if (still == 0 || exam >= P.max_rounds) break;   // the last solve already belongs to the final durations / cap reached
        }
        // (a speculative round may still be in flight: everything that follows is ordered behind it on the same stream)
    }
    if (still != 0 && !cap_solve_enqueued) {
        // cap reached with durations changed by the last re-allocation: one more solve so that the coefficients match d_times
        rc = corridor_solve(2, (const int32_t*)d_changed, nullptr, true);
        if (rc != UAVQP_OK) return rc;
    }
    // 4. check + repair
    int repairs = 0, before = -1, blocked = 0, after = 0, summary_unsolved = -1;
    double check_dt = 0.0;
    uavqp_grid* own_grid = nullptr;
    if (checking) {
        const double chk_r = P.check_robot_r > 0.0 ? P.check_robot_r : P.robot_r, chk_h = P.check_robot_h > 0.0 ? P.check_robot_h : P.robot_h;
        if (!grid) {
            rc = uavqp_obstacle_grid_build_device(ctx, d_obstacles, n_obs, chk_r + 0.1, &own_grid);
            if (rc != UAVQP_OK) return rc;
            grid = own_grid;
        }
        int pgrid = (int)(((long long)n * 8 + 255) / 256);
        if (pgrid > ctx->num_cus * 2) pgrid = ctx->num_cus * 2;
        for (int pass = 0;; ++pass) {
            // (the counter block is zero: pipe_begin_kernel, nothing since -- after a repair the zeroing kernel below)
            if (pass > 0) hipLaunchKernelGGL(uavqp::pipe_zero_kernel, dim3(1), dim3(64), 0, s, d_cnt);
            hipLaunchKernelGGL(uavqp::pipe_check_prep_kernel, dim3(pgrid), dim3(256), 0, s, n, uni, d_seg_offsets, (const double*)d_times,
                               (const double*)d_corr_lo, (const double*)d_corr_hi, pass == 0 ? d_roomy : (uint8_t*)nullptr, d_fh, P.check_samples, d_cnt);
            rc = ellipsoid_check_grid_impl(ctx, r, n, uni, d_seg_offsets, d_times, d_coeff_out, P.check_samples, 0.0, 0.0, &d_cnt->tmax_bits, grid,
                                           chk_r, chk_h, d_fh, nullptr);
            if (rc != UAVQP_OK) break;
            hipLaunchKernelGGL(uavqp::pipe_hits_kernel, dim3(cgrid), dim3(256), 0, s, n, (const int32_t*)d_fh, P.check_samples, (const uint8_t*)d_roomy, d_flag,
                               (const int32_t*)d_status_out, d_cnt);
            rc = read_counters();
            if (rc != UAVQP_OK) break;
            double tmax;
            std::memcpy(&tmax, &h_cnt->tmax_bits, sizeof(double));
            check_dt = tmax / (double)(P.check_samples - 1);          // the quotient the check kernel formed from the same bits
            summary_unsolved = (int)h_cnt->unsolved;
            after = (int)h_cnt->hit;
            if (before < 0) {
                before = (int)h_cnt->hit;
                blocked = (int)h_cnt->hit_blocked;
            }
            if (h_cnt->hit_repairable == 0 || repairs >= P.repair_rounds) break;
            // halve the boxes of the flagged trajectories towards their waypoints (last round: the waypoint equalities), re-solve
            // warm-started, re-allocate once (the durations only ever stretch) and solve again if that changed anything
            summary_unsolved = -1;                                     // statuses change below: the summary is counted again
            const double shrink = (repairs + 1 == P.repair_rounds) ? 0.0 : 0.5;
            hipLaunchKernelGGL(uavqp::pipe_shrink_kernel, dim3(n < ctx->num_cus * 32 ? n : ctx->num_cus * 32), dim3(64), 0, s, n, uni, d_seg_offsets,
                               d_waypoints, d_corr_lo, d_corr_hi, (const uint8_t*)d_flag, shrink);
            rc = corridor_solve(2, nullptr, (const unsigned char*)d_flag);
            if (rc != UAVQP_OK) break;
            rc = reallocate();
            if (rc != UAVQP_OK) break;
            still = (int)h_cnt->changed;
            if (still > 0) {
                rc = corridor_solve(2, (const int32_t*)d_changed);
                if (rc != UAVQP_OK) break;
            }
            ++repairs;
        }
        if (own_grid) (void)uavqp_obstacle_grid_destroy(ctx, own_grid);
        if (rc != UAVQP_OK) return rc;
    }
    // 5. summary (the last check pass counted the statuses along with its hits unless a repair re-solved something after it)
    if (summary_unsolved < 0) {
        hipLaunchKernelGGL(uavqp::pipe_zero_kernel, dim3(1), dim3(64), 0, s, d_cnt);
        hipLaunchKernelGGL(uavqp::pipe_count_kernel, dim3(cgrid), dim3(256), 0, s, (const int32_t*)nullptr, (const int32_t*)d_status_out, n, d_cnt);
        rc = read_counters();
        if (rc != UAVQP_OK) return rc;
        summary_unsolved = (int)h_cnt->unsolved;
    }
    UAVQP_HIP(hipGetLastError());
    if (result) {
        result->rounds = rounds;
        result->repairs = repairs;
        result->still_stretching = still;
        result->colliding_before_repair = before < 0 ? 0 : before;
        result->colliding_with_blocked_waypoints = blocked;
        result->colliding_after = after;
        result->unsolved = (int32_t)summary_unsolved;
        result->check_dt = check_dt;
    }
    return UAVQP_OK;
}

extern "C" int uavqp_corridor_pipeline_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                            const int32_t* seg_offsets, const double* waypoints, double* times, const double* bc,
                                            const double* obstacles, int n_obs, const uavqp_pipeline_params* params, double* coeff_out,
                                            int32_t* status_out, double* corr_lo, double* corr_hi, int32_t* first_hit,
                                            uavqp_pipeline_result* result) {
    if (!ctx || (r != 3 && r != 4) || n_traj < 0 || uniform_segments < 0 || n_obs < 0) return UAVQP_ERR_INVALID_ARG;
    if (result) *result = uavqp_pipeline_result{};
    if (n_traj == 0) return UAVQP_OK;
    if (!waypoints || !times || !bc || !coeff_out || (n_obs > 0 && !obstacles) || (uniform_segments == 0 && !seg_offsets)) return UAVQP_ERR_INVALID_ARG;
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
    if (total_seg > 0x7fffffffLL - n_traj) return UAVQP_ERR_INVALID_ARG;
    UAVQP_HIP(hipSetDevice(ctx->device));
    const size_t n_wp = 3 * (size_t)(total_seg + n_traj);
    const size_t b_off = uniform_segments > 0 ? 0 : align256(sizeof(int32_t) * (size_t)(n_traj + 1));
    const size_t b_wp = align256(sizeof(double) * n_wp);
    const size_t b_t = align256(sizeof(double) * (size_t)total_seg);
    const size_t b_bc = align256(sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3);
    const size_t b_obs = align256(sizeof(double) * 3 * (size_t)(n_obs > 0 ? n_obs : 1));
    const size_t b_out = align256(sizeof(double) * 3 * 2 * r * (size_t)total_seg);
    const size_t b_st = align256(sizeof(int32_t) * (size_t)n_traj);
    int rc = ensure_stage(ctx, b_off + 3 * b_wp + b_t + b_bc + b_obs + b_out + 2 * b_st);
    if (rc != UAVQP_OK) return rc;
    char* p = (char*)ctx->d_stage;
    int32_t* d_off = uniform_segments > 0 ? nullptr : (int32_t*)p; p += b_off;
    double* d_wp = (double*)p; p += b_wp;
    double* d_lo = (double*)p; p += b_wp;
    double* d_hi = (double*)p; p += b_wp;
    double* d_t = (double*)p; p += b_t;
    double* d_bc = (double*)p; p += b_bc;
    double* d_obs = (double*)p; p += b_obs;
    double* d_out = (double*)p; p += b_out;
    int32_t* d_st = (int32_t*)p; p += b_st;
    int32_t* d_fh = (int32_t*)p;
    hipStream_t s = ctx->stream;
    if (d_off) UAVQP_HIP(hipMemcpyAsync(d_off, seg_offsets, sizeof(int32_t) * (size_t)(n_traj + 1), hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_wp, waypoints, sizeof(double) * n_wp, hipMemcpyHostToDevice, s));
    if (total_seg > 0) UAVQP_HIP(hipMemcpyAsync(d_t, times, sizeof(double) * (size_t)total_seg, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemcpyAsync(d_bc, bc, sizeof(double) * (size_t)n_traj * 2 * (r - 1) * 3, hipMemcpyHostToDevice, s));
    if (n_obs > 0) UAVQP_HIP(hipMemcpyAsync(d_obs, obstacles, sizeof(double) * 3 * (size_t)n_obs, hipMemcpyHostToDevice, s));
    UAVQP_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * 3 * 2 * r * (size_t)total_seg, s));   // an invalid trajectory comes back as zeros
    rc = uavqp_corridor_pipeline_device(ctx, r, n_traj, uniform_segments, Mmax, (int)total_seg, d_off, d_wp, d_t, d_bc, d_obs, n_obs, nullptr, params,
                                        d_out, d_st, d_lo, d_hi, d_fh, result);
    if (rc != UAVQP_OK) return rc;
    if (total_seg > 0) {
        UAVQP_HIP(hipMemcpyAsync(coeff_out, d_out, sizeof(double) * 3 * 2 * r * (size_t)total_seg, hipMemcpyDeviceToHost, s));
        UAVQP_HIP(hipMemcpyAsync(times, d_t, sizeof(double) * (size_t)total_seg, hipMemcpyDeviceToHost, s));   // stretched by the re-allocation
    }
    if (status_out) UAVQP_HIP(hipMemcpyAsync(status_out, d_st, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    if (corr_lo) UAVQP_HIP(hipMemcpyAsync(corr_lo, d_lo, sizeof(double) * n_wp, hipMemcpyDeviceToHost, s));
    if (corr_hi) UAVQP_HIP(hipMemcpyAsync(corr_hi, d_hi, sizeof(double) * n_wp, hipMemcpyDeviceToHost, s));
    if (first_hit && params && params->check_samples > 0) UAVQP_HIP(hipMemcpyAsync(first_hit, d_fh, sizeof(int32_t) * (size_t)n_traj, hipMemcpyDeviceToHost, s));
    UAVQP_HIP(hipStreamSynchronize(s));
    return UAVQP_OK;
}
