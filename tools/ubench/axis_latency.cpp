// Latency of the reference's call shape through the C ABI, no Python: MinimumControl::solve = uavqp_solve_axis_host, three axes one after the other
// (test_minimum_jerk.cpp:75,100,125), BASELINE config 1 (8 waypoints, r = 4).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/uavqp.h"
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000, M = 7, r = 4;
    uavqp_ctx* ctx; if (uavqp_create(&ctx, 0) != 0) { printf("create failed: %s\n", uavqp_last_error()); return 1; }
    std::vector<double> pos[3], T(M, 1.0), coef(2 * r * M);
    for (int ax = 0; ax < 3; ++ax) for (int i = 0; i <= M; ++i) pos[ax].push_back(0.37 * i * (ax + 1) + 0.1 * ((i * 7 + ax) % 5));
    const double bv[2] = {0.3, 0.0}, ba[2] = {0.0, 0.0}, bj[2] = {0.0, 0.0};
    int32_t st = 0;
    for (int i = 0; i < 50; ++i) uavqp_solve_axis_host(ctx, r, M, pos[i % 3].data(), bv, ba, bj, T.data(), coef.data(), &st);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; ++i)
        for (int ax = 0; ax < 3; ++ax) uavqp_solve_axis_host(ctx, r, M, pos[ax].data(), bv, ba, bj, T.data(), coef.data(), &st);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    printf("three axis calls: %.1f us (%.1f us per call), status %d, coef[0] %.6f\n", us, us / 3, st, coef[0]);
    uavqp_destroy(ctx);
    return 0;
}
