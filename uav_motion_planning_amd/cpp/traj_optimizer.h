// TrajOptimizer -- batch facade with the method names BASELINE.json's north star asks for
// (setWaypoints / setTimeAllocation / solve / getPolyCoeff).  The reference has no such class
// (its only optimiser is traj_optimization::MinimumControl); this is the batch-oriented
// entry a planner would call once per replanning cycle for all candidate trajectories and all 3 axes.
// Header-only, no Eigen: plain pointers over the C ABI (include/uavqp.h).
#ifndef UAVQP_TRAJ_OPTIMIZER_H_
#define UAVQP_TRAJ_OPTIMIZER_H_

#include <algorithm>
#include <cstdint>
#include <iostream>
#include <vector>

#include "../../include/uavqp.h"

// The sharded (multi-GPU) entry keeps its buffers on the device and therefore needs the HIP runtime API for allocation and the
// host <-> device copies at its two ends; everything else here is plain C ABI.  A planner that runs one process (or thread) per GPU
// has it anyway.
#if defined(__has_include)
#if __has_include(<hip/hip_runtime_api.h>)
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#define UAVQP_TRAJ_OPTIMIZER_HAS_HIP 1
#endif
#endif

namespace traj_optimization {

class TrajOptimizer {
  public:
    explicit TrajOptimizer(int order = 4, int device = 0) : order_(order), device_(device) {}
    ~TrajOptimizer() { if (ctx_) uavqp_destroy(ctx_); }
    TrajOptimizer(const TrajOptimizer&) = delete;
    TrajOptimizer& operator=(const TrajOptimizer&) = delete;

    // xyz: [sum_b n_waypoints_b][3]; wp_offsets[n_traj+1]: first waypoint row of each trajectory
    // (what a list of A* / kino-A* / RRT* paths flattens to).
    void setWaypoints(const double* xyz, const int32_t* wp_offsets, int n_traj) {
        n_traj_ = n_traj;
        seg_offsets_.resize(n_traj + 1);
        for (int b = 0; b <= n_traj; ++b) seg_offsets_[b] = wp_offsets[b] - b;  // M_b = n_waypoints_b - 1
        wp_.assign(xyz, xyz + 3 * static_cast<size_t>(wp_offsets[n_traj]));
    }
    // T: [sum_b M_b] segment durations in trajectory order.
    void setTimeAllocation(const double* T) { T_.assign(T, T + (n_traj_ > 0 ? seg_offsets_[n_traj_] : 0)); }
    // bc: [n_traj][2][order-1][3] = [start|end][vel, acc(, jerk)][xyz]; default all zero.
    void setBoundary(const double* bc) { bc_.assign(bc, bc + static_cast<size_t>(n_traj_) * 2 * (order_ - 1) * 3); }
    // Optional corridor (north-star extension): boxes lo <= p <= hi, [sum_b n_waypoints_b][3] like xyz, replace the
    // interior-waypoint equalities; first/last rows of a trajectory are ignored.  nullptr restores the equalities.
    void setCorridor(const double* lo, const double* hi) {
        lo_.clear();
        hi_.clear();
        if (lo && hi) {
            lo_.assign(lo, lo + wp_.size());
            hi_.assign(hi, hi + wp_.size());
        }
    }

    // Optional general inequality rows (north-star extension; what the reference's solver interface accepts,
    // minimum_control.cpp:146-147,164-180): per segment and slot j < rows_per_segment (1 or 2) a row
    // lo <= p^(deriv)(tau * T_segment) <= hi per axis.  tau / deriv: [sum_b M_b][K], lo / hi: [sum_b M_b][K][3]; deriv < 0 = unused
    // slot.  Needs setTimeAllocation first (sizes).  rows_per_segment = 0 removes them.
    void setRows(int rows_per_segment, const double* tau, const int32_t* deriv, const double* lo, const double* hi) {
        rows_k_ = rows_per_segment;
        if (rows_k_ <= 0 || !tau || !deriv || !lo || !hi) { rows_k_ = 0; return; }   // (null arrays: no rows, never a wild read)
        const size_t n = T_.size() * static_cast<size_t>(rows_k_);
        row_tau_.assign(tau, tau + n);
        row_deriv_.assign(deriv, deriv + n);
        row_lo_.assign(lo, lo + 3 * n);
        row_hi_.assign(hi, hi + 3 * n);
    }

    // Solver settings (include/uavqp.h: the reference's warm_start / eps_prim_inf / max_iter, minimum_control.cpp:160-162, and the
    // library's own knobs).  Applied to the context now, or when it is created.
    bool setSettings(const uavqp_settings& st) {
        settings_ = st;
        have_settings_ = true;
        return !ctx_ || uavqp_set_settings(ctx_, &settings_) == UAVQP_OK;
    }

    bool solve() {
        if (n_traj_ <= 0 || T_.size() != static_cast<size_t>(seg_offsets_[n_traj_])) return false;
        if (!rowsMatchBatch() || !corridorMatchesBatch()) {   // rows / boxes installed for another batch size: refuse, never read past them
            std::cout << "solver solve failed! (rows or corridor arrays do not match the current batch)" << std::endl;
            return false;
        }
        if (!ensureContext()) return false;
        if (bc_.empty()) bc_.assign(static_cast<size_t>(n_traj_) * 2 * (order_ - 1) * 3, 0.0);
        coef_.assign(static_cast<size_t>(3) * 2 * order_ * seg_offsets_[n_traj_], 0.0);
        status_.assign(n_traj_, 0);
        int rc;
        if (rows_k_ > 0)
            rc = uavqp_solve_rows_batch_host(ctx_, order_, n_traj_, 0, 0, seg_offsets_.data(), wp_.data(), T_.data(), bc_.data(),
                                             lo_.empty() ? nullptr : lo_.data(), lo_.empty() ? nullptr : hi_.data(), rows_k_, row_tau_.data(),
                                             row_deriv_.data(), row_lo_.data(), row_hi_.data(), coef_.data(), status_.data(), nullptr);
        else if (lo_.empty())
            rc = uavqp_solve_batch_host(ctx_, order_, n_traj_, 0, 0, seg_offsets_.data(), wp_.data(), T_.data(), bc_.data(), coef_.data(), status_.data());
        else
            rc = uavqp_solve_corridor_batch_host(ctx_, order_, n_traj_, 0, 0, seg_offsets_.data(), wp_.data(), T_.data(), bc_.data(),
                                                 lo_.data(), hi_.data(), coef_.data(), status_.data(), nullptr);
        if (rc != UAVQP_OK) {
            std::cout << "solver solve failed! (" << uavqp_last_error() << ")" << std::endl;
            return false;
        }
        for (int32_t s : status_) if (s != UAVQP_SOLVED) return false;
        return true;
    }
    // Flat coefficients: trajectory b starts at 3*2r*segOffset(b), layout [axis][segment][2r], ascending powers.
    const double* getPolyCoeff() const { return coef_.data(); }
    const double* getPolyCoeff(int traj, int axis) const {
        const int M = seg_offsets_[traj + 1] - seg_offsets_[traj];
        return coef_.data() + static_cast<size_t>(3) * 2 * order_ * seg_offsets_[traj] + static_cast<size_t>(axis) * 2 * order_ * M;
    }
    int segOffset(int traj) const { return seg_offsets_[traj]; }
    const std::vector<int32_t>& status() const { return status_; }

    // BASELINE config 5 as one call (include/uavqp.h uavqp_corridor_pipeline_host): plain solve -> corridor boxes from the obstacle cloud
    // (SE(3) robot ellipsoid of KinoAstar::isCollisionFree) -> <= max_rounds x (corridor solve + time re-allocation) -> collision check
    // -> repair.  obstacles [n_obs][3].  On return: getPolyCoeff() = the final polynomials, timeAllocation() = the stretched durations,
    // corridorLo() / corridorHi() = the boxes of the final solve, firstHit() = first colliding sample per trajectory (check_samples =
    // free), pipelineResult() = the summary.  true iff the call succeeded and every trajectory is SOLVED (collisions are reported, not
    // turned into failure: a planner decides what to do with a colliding candidate).
    bool solvePipeline(const double* obstacles, int n_obs, const uavqp_pipeline_params* params = nullptr) {
        if (n_traj_ <= 0 || T_.size() != static_cast<size_t>(seg_offsets_[n_traj_])) return false;
        if (!ensureContext()) return false;
        if (bc_.empty()) bc_.assign(static_cast<size_t>(n_traj_) * 2 * (order_ - 1) * 3, 0.0);
        uavqp_pipeline_params pp;
        if (params) pp = *params; else uavqp_default_pipeline_params(&pp);
        coef_.assign(static_cast<size_t>(3) * 2 * order_ * seg_offsets_[n_traj_], 0.0);
        status_.assign(n_traj_, 0);
        // (the pipeline's boxes go to their own members: lo_ / hi_ are what setCorridor() installed and what a later plain solve() keys on -- ADVICE r3)
        pipe_lo_.assign(wp_.size(), 0.0);
        pipe_hi_.assign(wp_.size(), 0.0);
        first_hit_.assign(n_traj_, pp.check_samples);
        pipe_result_ = uavqp_pipeline_result{};
        const int rc = uavqp_corridor_pipeline_host(ctx_, order_, n_traj_, 0, 0, seg_offsets_.data(), wp_.data(), T_.data(), bc_.data(), obstacles, n_obs,
                                                    &pp, coef_.data(), status_.data(), pipe_lo_.data(), pipe_hi_.data(), first_hit_.data(), &pipe_result_);
        if (rc != UAVQP_OK) {
            std::cout << "solver solve failed! (" << uavqp_last_error() << ")" << std::endl;
            return false;
        }
        return pipe_result_.unsolved == 0;
    }
    const std::vector<double>& timeAllocation() const { return T_; }
    const std::vector<double>& corridorLo() const { return pipe_lo_; }   // boxes of the last solvePipeline() (setCorridor()'s own are untouched)
    const std::vector<double>& corridorHi() const { return pipe_hi_; }
    const std::vector<int32_t>& firstHit() const { return first_hit_; }
    const uavqp_pipeline_result& pipelineResult() const { return pipe_result_; }

    // ---- multi-GPU: one process (or thread) per GPU, every rank holds the whole batch description, solves its contiguous shard
    // (balanced by segment count) on its device and ends with ALL coefficients: uavqp_shard_bounds_ragged, uavqp_comm_create,
    // uavqp_solve_*_batch_device on views, uavqp_allgather_coeffs / _status (RCCL over xGMI).  Rank 0 draws the id and ships it to
    // the others with whatever the program has (MPI_Bcast, a socket, a file).
    static bool uniqueId(unsigned char* id_out /* UAVQP_UNIQUE_ID_BYTES */) { return uavqp_comm_unique_id(id_out) == UAVQP_OK; }
    bool initDistributed(int rank, int world, const unsigned char* unique_id) {
        if (!ensureContext()) return false;
        rank_ = rank;
        world_ = world;
        if (uavqp_comm_create(ctx_, rank, world, unique_id) != UAVQP_OK) {
            std::cout << "solver init failed! (" << uavqp_last_error() << ")" << std::endl;
            return false;
        }
        return true;
    }
#ifdef UAVQP_TRAJ_OPTIMIZER_HAS_HIP
    bool solveSharded() {
        if (n_traj_ <= 0 || world_ < 1 || !ctx_ || T_.size() != static_cast<size_t>(seg_offsets_[n_traj_])) return false;
        if (!rowsMatchBatch() || !corridorMatchesBatch()) {
            std::cout << "solver solve failed! (rows or corridor arrays do not match the current batch)" << std::endl;
            return false;
        }
        if (bc_.empty()) bc_.assign(static_cast<size_t>(n_traj_) * 2 * (order_ - 1) * 3, 0.0);
        std::vector<int32_t> bounds(world_ + 1);
        if (uavqp_shard_bounds_ragged(seg_offsets_.data(), n_traj_, world_, bounds.data()) != UAVQP_OK) return false;
        const int b0 = bounds[rank_], b1 = bounds[rank_ + 1], nl = b1 - b0;
        const int32_t s0 = seg_offsets_[b0], s1 = seg_offsets_[b1];
        const size_t nc = static_cast<size_t>(3) * 2 * order_, tot = static_cast<size_t>(seg_offsets_[n_traj_]);
        std::vector<int32_t> so_l(nl + 1);
        int mmax = 1;
        for (int b = 0; b <= nl; ++b) so_l[b] = seg_offsets_[b0 + b] - s0;
        for (int b = 0; b < nl; ++b) mmax = std::max(mmax, so_l[b + 1] - so_l[b]);
        std::vector<int64_t> c_counts(world_), t_counts(world_);
        size_t c_off = 0;
        for (int g = 0; g < world_; ++g) {
            c_counts[g] = static_cast<int64_t>(nc) * (seg_offsets_[bounds[g + 1]] - seg_offsets_[bounds[g]]);
            t_counts[g] = bounds[g + 1] - bounds[g];
            if (g < rank_) c_off += static_cast<size_t>(c_counts[g]);
        }
        const size_t n_wp = static_cast<size_t>(3) * ((s1 - s0) + nl), n_t = static_cast<size_t>(s1 - s0), n_bc = static_cast<size_t>(nl) * 2 * (order_ - 1) * 3;
        int32_t *d_so = nullptr, *d_st = nullptr, *d_rd = nullptr;
        double *d_wp = nullptr, *d_T = nullptr, *d_bc = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_coef = nullptr, *d_rt = nullptr, *d_rl = nullptr, *d_rh = nullptr;
        bool ok = hipSetDevice(device_) == hipSuccess;
        auto up = [&](void** d, const void* h, size_t bytes) {
            ok = ok && hipMalloc(d, bytes ? bytes : 8) == hipSuccess && (bytes == 0 || hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess);
        };
        up(reinterpret_cast<void**>(&d_so), so_l.data(), sizeof(int32_t) * (nl + 1));
        up(reinterpret_cast<void**>(&d_wp), wp_.data() + static_cast<size_t>(3) * (s0 + b0), sizeof(double) * n_wp);
        up(reinterpret_cast<void**>(&d_T), T_.data() + s0, sizeof(double) * n_t);
        up(reinterpret_cast<void**>(&d_bc), bc_.data() + static_cast<size_t>(b0) * 2 * (order_ - 1) * 3, sizeof(double) * n_bc);
        if (!lo_.empty()) {
            up(reinterpret_cast<void**>(&d_lo), lo_.data() + static_cast<size_t>(3) * (s0 + b0), sizeof(double) * n_wp);
            up(reinterpret_cast<void**>(&d_hi), hi_.data() + static_cast<size_t>(3) * (s0 + b0), sizeof(double) * n_wp);
        }
        if (rows_k_ > 0) {   // the shard's general inequality rows (per segment, rows_k_ slots)
            const size_t k = static_cast<size_t>(rows_k_);
            up(reinterpret_cast<void**>(&d_rt), row_tau_.data() + k * s0, sizeof(double) * k * n_t);
            up(reinterpret_cast<void**>(&d_rd), row_deriv_.data() + k * s0, sizeof(int32_t) * k * n_t);
            up(reinterpret_cast<void**>(&d_rl), row_lo_.data() + 3 * k * s0, sizeof(double) * 3 * k * n_t);
            up(reinterpret_cast<void**>(&d_rh), row_hi_.data() + 3 * k * s0, sizeof(double) * 3 * k * n_t);
        }
        ok = ok && hipMalloc(reinterpret_cast<void**>(&d_coef), sizeof(double) * nc * tot) == hipSuccess &&
             hipMalloc(reinterpret_cast<void**>(&d_st), sizeof(int32_t) * n_traj_) == hipSuccess;
        // the device kernels leave a trajectory they flag UAVQP_INVALID_INPUT untouched: zeros, as the host entry points give
        ok = ok && hipMemset(d_coef, 0, sizeof(double) * nc * tot) == hipSuccess && hipMemset(d_st, 0, sizeof(int32_t) * n_traj_) == hipSuccess;
        int rc = ok ? UAVQP_OK : UAVQP_ERR_ALLOC;
        if (ok && nl > 0) {   // this rank's shard, written straight into its slice of the full buffers
            if (rows_k_ > 0)
                rc = uavqp_solve_rows_batch_device(ctx_, order_, nl, 0, mmax, d_so, d_wp, d_T, d_bc, d_lo, d_hi, rows_k_, d_rt, d_rd, d_rl, d_rh,
                                                   d_coef + c_off, d_st + b0, nullptr, nullptr);
            else if (lo_.empty())
                rc = uavqp_solve_batch_device(ctx_, order_, nl, 0, mmax, d_so, d_wp, d_T, d_bc, d_coef + c_off, d_st + b0);
            else
                rc = uavqp_solve_corridor_batch_device(ctx_, order_, nl, 0, mmax, d_so, d_wp, d_T, d_bc, d_lo, d_hi, d_coef + c_off, d_st + b0, nullptr);
        }
        if (rc == UAVQP_OK) rc = uavqp_allgather_coeffs(ctx_, d_coef + c_off, c_counts.data(), d_coef);
        if (rc == UAVQP_OK) rc = uavqp_allgather_status(ctx_, d_st + b0, t_counts.data(), d_st);
        if (rc == UAVQP_OK) rc = uavqp_synchronize(ctx_);
        coef_.assign(nc * tot, 0.0);
        status_.assign(n_traj_, 0);
        if (rc == UAVQP_OK) {
            ok = hipMemcpy(coef_.data(), d_coef, sizeof(double) * nc * tot, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(status_.data(), d_st, sizeof(int32_t) * n_traj_, hipMemcpyDeviceToHost) == hipSuccess;
        }
        for (void* p : {static_cast<void*>(d_so), static_cast<void*>(d_st), static_cast<void*>(d_wp), static_cast<void*>(d_T), static_cast<void*>(d_bc),
                        static_cast<void*>(d_lo), static_cast<void*>(d_hi), static_cast<void*>(d_coef), static_cast<void*>(d_rt), static_cast<void*>(d_rd),
                        static_cast<void*>(d_rl), static_cast<void*>(d_rh)})
            if (p) (void)hipFree(p);
        if (rc != UAVQP_OK || !ok) {
            std::cout << "solver solve failed! (" << uavqp_last_error() << ")" << std::endl;
            return false;
        }
        for (int32_t s : status_) if (s != UAVQP_SOLVED) return false;
        return true;
    }
#endif

  private:
    bool rowsMatchBatch() const {
        if (rows_k_ <= 0) return true;
        const size_t n = T_.size() * static_cast<size_t>(rows_k_);
        return row_tau_.size() == n && row_deriv_.size() == n && row_lo_.size() == 3 * n && row_hi_.size() == 3 * n;
    }
    bool corridorMatchesBatch() const { return lo_.empty() || (lo_.size() == wp_.size() && hi_.size() == wp_.size()); }
    bool ensureContext() {
        if (ctx_) return true;
        if (uavqp_create(&ctx_, device_) != UAVQP_OK) {
            std::cout << "solver init failed! (" << uavqp_last_error() << ")" << std::endl;
            ctx_ = nullptr;
            return false;
        }
        if (have_settings_ && uavqp_set_settings(ctx_, &settings_) != UAVQP_OK) {
            std::cout << "solver init failed! (" << uavqp_last_error() << ")" << std::endl;
            return false;
        }
        return true;
    }

    int order_, device_, n_traj_ = 0;
    int rank_ = 0, world_ = 1;
    bool have_settings_ = false;
    uavqp_settings settings_{};
    uavqp_ctx* ctx_ = nullptr;
    std::vector<int32_t> seg_offsets_, status_;
    std::vector<double> wp_, T_, bc_, coef_, lo_, hi_, pipe_lo_, pipe_hi_, row_tau_, row_lo_, row_hi_;
    std::vector<int32_t> row_deriv_, first_hit_;
    int rows_k_ = 0;
    uavqp_pipeline_result pipe_result_{};
};

}  // namespace traj_optimization
#endif
