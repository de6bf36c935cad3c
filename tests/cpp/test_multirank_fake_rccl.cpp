// world > 1 through the C ABI on ONE GPU: the ranks are threads of this process, each with its own uavqp_ctx on device 0, and
// librccl.so.1 is the TEST-ONLY stand-in of tests/cpp/fake_rccl/ (selected through LD_LIBRARY_PATH by
// tests/test_gpu_multirank_fake_rccl.py; this program has no torch / real RCCL in it).  What runs here for the first time with more than
// one rank: uavqp_comm_create with world 2 and 3, uavqp_allgather_coeffs / _status on UNEQUAL shards (the grouped ncclSend / ncclRecv
// branch of allgather_shards, uavqp.hip) including a ZERO-sized shard and d_local aliasing its own slot of d_full, the equal-shard
// ncclAllGather branch, and TrajOptimizer::solveSharded (plain, corridor, rows) on two ranks against the single-process host solve.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "traj_optimizer.h"

static int g_bad = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); __atomic_add_fetch(&g_bad, 1, __ATOMIC_SEQ_CST); } } while (0)

// ---- 1. raw all-gather(-v): world ranks, counts[g] doubles from rank g (value = 1000 g + i), in place or not
static void gather_case(int world, const std::vector<int64_t>& counts, bool in_place) {
    unsigned char id[UAVQP_UNIQUE_ID_BYTES];
    if (uavqp_comm_unique_id(id) != UAVQP_OK) { CHECK(false, "unique id: %s", uavqp_last_error()); return; }
    int64_t total = 0;
    std::vector<int64_t> off(world + 1, 0);
    for (int g = 0; g < world; ++g) { off[g + 1] = off[g] + counts[g]; total += counts[g]; }
    std::vector<std::thread> th;
    for (int rank = 0; rank < world; ++rank)
        th.emplace_back([&, rank] {
            hipSetDevice(0);
            uavqp_ctx* ctx = nullptr;
            if (uavqp_create(&ctx, 0) != UAVQP_OK) { CHECK(false, "create"); return; }
            CHECK(uavqp_comm_create(ctx, rank, world, id) == UAVQP_OK, "comm_create rank %d: %s", rank, uavqp_last_error());
            {   // what the communicator library itself says (ncclCommUserRank / ncclCommCount): bench.py's allgather.rccl_world
                int32_t rk = -1, wd = -1;
                CHECK(uavqp_comm_info(ctx, &rk, &wd) == UAVQP_OK && rk == rank && wd == world, "comm_info rank %d: got (%d, %d): %s", rank, rk, wd, uavqp_last_error());
            }
            double *d_full = nullptr, *d_loc = nullptr;
            int32_t *s_full = nullptr, *s_loc = nullptr;
            hipMalloc((void**)&d_full, sizeof(double) * (total + 1));
            hipMalloc((void**)&s_full, sizeof(int32_t) * (total + 1));
            hipMemset(d_full, 0xff, sizeof(double) * (total + 1));
            hipMemset(s_full, 0xff, sizeof(int32_t) * (total + 1));
            std::vector<double> mine(counts[rank] + 1);
            std::vector<int32_t> mine_i(counts[rank] + 1);
            for (int64_t i = 0; i < counts[rank]; ++i) { mine[i] = 1000.0 * rank + i; mine_i[i] = 1000 * rank + (int)i; }
            if (in_place) { d_loc = d_full + off[rank]; s_loc = s_full + off[rank]; }
            else { hipMalloc((void**)&d_loc, sizeof(double) * (counts[rank] + 1)); hipMalloc((void**)&s_loc, sizeof(int32_t) * (counts[rank] + 1)); }
            hipMemcpy(d_loc, mine.data(), sizeof(double) * counts[rank], hipMemcpyHostToDevice);
            hipMemcpy(s_loc, mine_i.data(), sizeof(int32_t) * counts[rank], hipMemcpyHostToDevice);
            CHECK(uavqp_allgather_coeffs(ctx, counts[rank] ? d_loc : nullptr, counts.data(), d_full) == UAVQP_OK, "allgather_coeffs rank %d: %s", rank, uavqp_last_error());
            CHECK(uavqp_allgather_status(ctx, counts[rank] ? s_loc : nullptr, counts.data(), s_full) == UAVQP_OK, "allgather_status rank %d: %s", rank, uavqp_last_error());
            CHECK(uavqp_synchronize(ctx) == UAVQP_OK, "sync");
            std::vector<double> full(total);
            std::vector<int32_t> full_i(total);
            hipMemcpy(full.data(), d_full, sizeof(double) * total, hipMemcpyDeviceToHost);
            hipMemcpy(full_i.data(), s_full, sizeof(int32_t) * total, hipMemcpyDeviceToHost);
            int wrong = 0;
            for (int g = 0; g < world; ++g)
                for (int64_t i = 0; i < counts[g]; ++i) wrong += full[off[g] + i] != 1000.0 * g + i || full_i[off[g] + i] != 1000 * g + (int)i;
            CHECK(wrong == 0, "world %d rank %d in_place %d: %d wrong entries", world, rank, (int)in_place, wrong);
            if (!in_place) { hipFree(d_loc); hipFree(s_loc); }
            hipFree(d_full); hipFree(s_full);
            uavqp_destroy(ctx);
        });
    for (auto& t : th) t.join();
}

int main() {
    hipSetDevice(0);
    gather_case(2, {7, 7}, false);          // equal shards: ncclAllGather
    gather_case(2, {7, 7}, true);
    gather_case(2, {12, 5}, false);         // unequal: grouped send / recv
    gather_case(2, {12, 5}, true);
    gather_case(3, {5, 0, 9}, false);       // a zero-sized shard in the middle
    gather_case(3, {5, 0, 9}, true);
    gather_case(3, {0, 0, 4}, true);
    // the driver's node: eight ranks -- equal shards (configs 2 / 3: ncclAllGather) and the ragged shards of configs 4 / 5 (grouped send / recv)
    gather_case(8, {6, 6, 6, 6, 6, 6, 6, 6}, true);
    gather_case(8, {6, 6, 6, 6, 6, 6, 6, 6}, false);
    gather_case(8, {9, 4, 7, 0, 11, 5, 8, 3}, true);
    gather_case(8, {9, 4, 7, 0, 11, 5, 8, 3}, false);
    std::printf("raw gathers done, failures so far %d\n", g_bad);

    // ---- 2. TrajOptimizer::solveSharded on two ranks vs the single-process host solve
    const int r = 4, n = 61;
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
            for (int a = 0; a < 3; ++a) { p[a] += 1.5 * u(rng); xyz[3 * i + a] = p[a]; lo[3 * i + a] = p[a] - 0.4; hi[3 * i + a] = p[a] + 0.4; }
    }
    for (auto& x : T) x = t(rng);
    for (auto& x : bc) x = 0.3 * u(rng);
    std::vector<double> tau(segs, 0.5), rlo(3 * segs, -2.5), rhi(3 * segs, 2.5);   // a mid-segment velocity limit per axis
    std::vector<int32_t> drv(segs, 1);
    for (int mode = 0; mode < 3; ++mode) {   // 0 equalities, 1 corridor, 2 corridor + rows
        auto setup = [&](traj_optimization::TrajOptimizer& o) {
            o.setWaypoints(xyz.data(), wp_off.data(), n);
            o.setTimeAllocation(T.data());
            o.setBoundary(bc.data());
            if (mode >= 1) o.setCorridor(lo.data(), hi.data());
            if (mode == 2) o.setRows(1, tau.data(), drv.data(), rlo.data(), rhi.data());
        };
        traj_optimization::TrajOptimizer host(r);
        setup(host);
        const bool host_ok = host.solve();
        unsigned char id[UAVQP_UNIQUE_ID_BYTES];
        CHECK(traj_optimization::TrajOptimizer::uniqueId(id), "unique id");
        const size_t nc = static_cast<size_t>(3) * 2 * r * segs;
        std::vector<std::thread> th;
        const int world_sh = mode == 0 ? 8 : 2;      // (the equality solve also on the driver's eight ranks)
        for (int rank = 0; rank < world_sh; ++rank)
            th.emplace_back([&, rank] {
                hipSetDevice(0);
                traj_optimization::TrajOptimizer sh(r);
                setup(sh);
                CHECK(sh.initDistributed(rank, world_sh, id), "initDistributed rank %d", rank);
                const bool ok = sh.solveSharded();
                double worst = 0.0, scale = 0.0;
                int st_diff = 0;
                for (int b = 0; b < n; ++b) st_diff += sh.status()[b] != host.status()[b];
                for (size_t i = 0; i < nc; ++i) {
                    worst = std::fmax(worst, std::fabs(sh.getPolyCoeff()[i] - host.getPolyCoeff()[i]));
                    scale = std::fmax(scale, std::fabs(host.getPolyCoeff()[i]));
                }
                std::printf("mode %d rank %d: ok %d (host %d), status differences %d, max |host - sharded| = %.3e (scale %.3e)\n", mode, rank, (int)ok,
                            (int)host_ok, st_diff, worst, scale);
                CHECK(ok == host_ok && st_diff == 0 && worst <= 1e-9 * scale, "mode %d rank %d disagrees with the host solve", mode, rank);
            });
        for (auto& x : th) x.join();
    }
    std::printf(g_bad ? "FAILED (%d)\n" : "OK\n", g_bad);
    return g_bad ? 1 : 0;
}
