// The C++ batch facade's multi-GPU path on one GPU (world size 1: RCCL communicator of one rank) against its own host-pointer
// solve: same coefficients, same statuses; with a corridor too.  Exercises uavqp_comm_unique_id / uavqp_comm_create /
// uavqp_shard_bounds_ragged / uavqp_solve_*_batch_device on views / uavqp_allgather_coeffs / _status from C++, as a planner
// process per GPU would call them.  Built and run by tests/test_gpu_comm.py.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "traj_optimizer.h"

int main() {
    const int r = 4, n = 37;
    std::mt19937_64 rng(20260925);
    std::uniform_real_distribution<double> u(-1.0, 1.0), t(0.4, 2.0);
    std::uniform_int_distribution<int> mm(1, 14);
    std::vector<int32_t> wp_off(n + 1, 0);
    for (int b = 0; b < n; ++b) wp_off[b + 1] = wp_off[b] + mm(rng) + 1;
    const int rows = wp_off[n], segs = rows - n;
    std::vector<double> xyz(3 * rows), T(segs), bc(static_cast<size_t>(n) * 2 * (r - 1) * 3), lo(3 * rows), hi(3 * rows);
    for (int b = 0; b < n; ++b) {
        double p[3] = {u(rng) * 10, u(rng) * 5, 1.5 + 0.5 * u(rng)};
        for (int i = wp_off[b]; i < wp_off[b + 1]; ++i)
            for (int a = 0; a < 3; ++a) {
                p[a] += 1.5 * u(rng);
                xyz[3 * i + a] = p[a];
                lo[3 * i + a] = p[a] - 0.4;
                hi[3 * i + a] = p[a] + 0.4;
            }
    }
    for (auto& x : T) x = t(rng);
    for (auto& x : bc) x = 0.3 * u(rng);

    int bad = 0;
    for (int corridor = 0; corridor < 2; ++corridor) {
        traj_optimization::TrajOptimizer host(r), shard(r);
        for (auto* o : {&host, &shard}) {
            o->setWaypoints(xyz.data(), wp_off.data(), n);
            o->setTimeAllocation(T.data());
            o->setBoundary(bc.data());
            if (corridor) o->setCorridor(lo.data(), hi.data());
        }
        uavqp_settings st;
        uavqp_default_settings(&st);
        st.max_iter = 1000;   // minimum_control.cpp:162
        if (!shard.setSettings(st)) { std::printf("setSettings failed\n"); return 2; }
        unsigned char id[UAVQP_UNIQUE_ID_BYTES];
        if (!traj_optimization::TrajOptimizer::uniqueId(id) || !shard.initDistributed(0, 1, id)) { std::printf("comm init failed\n"); return 2; }
        if (!host.solve() || !shard.solveSharded()) { std::printf("solve failed (corridor %d)\n", corridor); return 3; }
        const size_t nc = static_cast<size_t>(3) * 2 * r * segs;
        double worst = 0.0, scale = 0.0;
        for (size_t i = 0; i < nc; ++i) {
            worst = std::fmax(worst, std::fabs(host.getPolyCoeff()[i] - shard.getPolyCoeff()[i]));
            scale = std::fmax(scale, std::fabs(host.getPolyCoeff()[i]));
        }
        std::printf("corridor %d: max |host - sharded| = %.3e (scale %.3e)\n", corridor, worst, scale);
        if (!(worst <= 1e-12 * scale)) ++bad;
        for (int b = 0; b < n; ++b) if (host.status()[b] != shard.status()[b]) ++bad;
    }
    // general rows through the facade (setRows -> uavqp_solve_rows_batch_host): a velocity limit at mid-segment that binds somewhere,
    // checked on the emitted polynomials
    {
        traj_optimization::TrajOptimizer rows(r);
        rows.setWaypoints(xyz.data(), wp_off.data(), n);
        rows.setTimeAllocation(T.data());
        rows.setBoundary(bc.data());
        std::vector<double> tau(segs, 0.5), rlo(3 * segs), rhi(3 * segs);
        std::vector<int32_t> drv(segs, 1);
        traj_optimization::TrajOptimizer free_(r);
        free_.setWaypoints(xyz.data(), wp_off.data(), n);
        free_.setTimeAllocation(T.data());
        free_.setBoundary(bc.data());
        if (!free_.solve()) { std::printf("plain solve failed\n"); return 3; }
        auto vel_mid = [&](const traj_optimization::TrajOptimizer& o, int b, int a, int k, double Tk) {
            const double* c = o.getPolyCoeff(b, a) + 2 * r * k;
            double v = 0.0, tp = 1.0;
            for (int q = 1; q < 2 * r; ++q) { v += q * c[q] * tp; tp *= 0.5 * Tk; }
            return v;
        };
        int seg = 0, n_bind = 0;
        for (int b = 0; b < n; ++b)
            for (int k = 0; k < wp_off[b + 1] - wp_off[b] - 1; ++k, ++seg)
                for (int a = 0; a < 3; ++a) {
                    const double v = vel_mid(free_, b, a, k, T[seg]);
                    const double lim = 0.85 * std::fabs(v) + 0.2;     // 15 % below the unconstrained mid-segment speed (plus slack)
                    rlo[3 * seg + a] = -lim;
                    rhi[3 * seg + a] = lim;
                }
        rows.setRows(1, tau.data(), drv.data(), rlo.data(), rhi.data());
        const bool all_solved = rows.solve();
        seg = 0;
        int n_solved = 0;
        double worst_over = -1.0;
        for (int b = 0; b < n; ++b) {
            const bool solved = rows.status()[b] == UAVQP_SOLVED;
            n_solved += solved;
            for (int k = 0; k < wp_off[b + 1] - wp_off[b] - 1; ++k, ++seg)
                for (int a = 0; a < 3 && solved; ++a) {
                    const double v = vel_mid(rows, b, a, k, T[seg]);
                    const double over = std::fabs(v) - rhi[3 * seg + a];
                    worst_over = std::fmax(worst_over, over / (1.0 + rhi[3 * seg + a]));
                    if (std::fabs(over) < 1e-7 * (1.0 + rhi[3 * seg + a])) ++n_bind;
                }
        }
        std::printf("rows: %d of %d solved (all %d), %d binding velocity rows, worst relative excess %.3e\n", n_solved, n, (int)all_solved, n_bind, worst_over);
        if (n_solved < n / 2 || n_bind < 5 || worst_over > 1e-7) ++bad;
        // the SAME rows through the sharded (device-buffer) path: rows must be enforced there too (they used to be ignored
        // with status SOLVED), statuses and coefficients equal the host-pointer solve
        {
            traj_optimization::TrajOptimizer sh(r);
            sh.setWaypoints(xyz.data(), wp_off.data(), n);
            sh.setTimeAllocation(T.data());
            sh.setBoundary(bc.data());
            sh.setRows(1, tau.data(), drv.data(), rlo.data(), rhi.data());
            unsigned char id[UAVQP_UNIQUE_ID_BYTES];
            if (!traj_optimization::TrajOptimizer::uniqueId(id) || !sh.initDistributed(0, 1, id)) { std::printf("comm init failed\n"); return 2; }
            const bool ok_sh = sh.solveSharded();
            double worst = 0.0, scale = 0.0;
            int st_diff = 0;
            for (int b = 0; b < n; ++b) st_diff += sh.status()[b] != rows.status()[b];
            for (size_t i = 0; i < static_cast<size_t>(3) * 2 * r * segs; ++i) {
                worst = std::fmax(worst, std::fabs(sh.getPolyCoeff()[i] - rows.getPolyCoeff()[i]));
                scale = std::fmax(scale, std::fabs(rows.getPolyCoeff()[i]));
            }
            std::printf("rows sharded: ok %d (host %d), status differences %d, max |host - sharded| = %.3e\n", (int)ok_sh, (int)all_solved, st_diff, worst);
            if (ok_sh != all_solved || st_diff || !(worst <= 1e-12 * scale)) ++bad;
        }
        // rows installed for ANOTHER batch size are refused (never read past the arrays)
        {
            traj_optimization::TrajOptimizer o(r);
            o.setWaypoints(xyz.data(), wp_off.data(), 3);                 // 3 trajectories ...
            std::vector<double> T3(T.begin(), T.begin() + (wp_off[3] - 3));
            o.setTimeAllocation(T3.data());
            o.setRows(1, tau.data(), drv.data(), rlo.data(), rhi.data()); // ... rows sized for them
            o.setWaypoints(xyz.data(), wp_off.data(), n);                 // ... then the batch grows
            o.setTimeAllocation(T.data());
            if (o.solve()) { std::printf("mismatched rows were accepted\n"); ++bad; }
            o.setRows(1, nullptr, nullptr, nullptr, nullptr);             // null arrays = no rows
            if (!o.solve()) { std::printf("solve after dropping the rows failed\n"); ++bad; }
        }
    }
    // BASELINE config 5 from C++: TrajOptimizer::solvePipeline (uavqp_corridor_pipeline_host) on a small pillar map
    {
        traj_optimization::TrajOptimizer pl(r);
        pl.setWaypoints(xyz.data(), wp_off.data(), n);
        std::vector<double> Tp(segs, 0.3);                                // short durations: the re-allocation has to stretch them
        pl.setTimeAllocation(Tp.data());
        pl.setBoundary(bc.data());
        std::vector<double> obs;
        for (int k = 0; k < 40; ++k) {                                    // 40 pillars of radius 0.5, points every 0.2 m
            const double cx = 12.0 * u(rng), cy = 7.0 * u(rng);
            for (int a = 0; a < 16; ++a)
                for (int z = 0; z < 15; ++z) { obs.push_back(cx + 0.5 * std::cos(a * 0.3927)); obs.push_back(cy + 0.5 * std::sin(a * 0.3927)); obs.push_back(0.2 * z); }
        }
        uavqp_pipeline_params pp;
        uavqp_default_pipeline_params(&pp);
        const bool okp = pl.solvePipeline(obs.data(), static_cast<int>(obs.size() / 3), &pp);
        const uavqp_pipeline_result& pr = pl.pipelineResult();
        int n_free = 0, shrunk = 0, box_viol = 0;
        double vmax2 = 0.0;
        int seg = 0;
        for (int b = 0; b < n; ++b) {
            n_free += pl.firstHit()[b] >= pp.check_samples;
            const int M = wp_off[b + 1] - wp_off[b] - 1;
            for (int k = 0; k < M; ++k, ++seg) {
                shrunk += pl.timeAllocation()[seg] < 0.3 * (1 - 1e-15);
                double v2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    const double* c = pl.getPolyCoeff(b, a) + 2 * r * k;
                    if (k > 0) {                                          // interior knot k: start of segment k must sit in its box
                        const size_t row = static_cast<size_t>(wp_off[b] + k);
                        box_viol += c[0] < pl.corridorLo()[3 * row + a] - 1e-9 || c[0] > pl.corridorHi()[3 * row + a] + 1e-9;
                    }
                    double v = 0.0, tp = 1.0;
                    for (int q = 1; q < 2 * r; ++q) { v += q * c[q] * tp; tp *= 0.5 * pl.timeAllocation()[seg]; }
                    v2 += v * v;
                }
                vmax2 = std::fmax(vmax2, v2);
            }
        }
        std::printf("pipeline: ok %d, rounds %d, repairs %d, still stretching %d, colliding %d -> %d (blocked %d), collision-free %d of %d, unsolved %d, "
                    "mid-segment speed max %.2f m/s, box violations %d, shrunk durations %d\n", (int)okp, pr.rounds, pr.repairs, pr.still_stretching,
                    pr.colliding_before_repair, pr.colliding_after, pr.colliding_with_blocked_waypoints, n_free, n, pr.unsolved, std::sqrt(vmax2), box_viol, shrunk);
        if (!okp || pr.rounds < 1 || pr.rounds > pp.max_rounds || pr.repairs > pp.repair_rounds || pr.unsolved != 0 || box_viol || shrunk ||
            n_free != n - pr.colliding_after || pr.colliding_after > pr.colliding_before_repair || (pr.still_stretching == 0 && std::sqrt(vmax2) > 1.05 * pp.v_max))
            ++bad;
    }
    std::printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 1 : 0;
}
