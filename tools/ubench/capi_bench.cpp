// Times uavqp_solve_batch_device through the C ABI with plain hipMalloc buffers (no torch in the process).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../include/uavqp.h"

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 200, r = argc > 3 ? atoi(argv[3]) : 4, M = argc > 4 ? atoi(argv[4]) : 8;
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-2, 2), ut(0.5, 2.0);
    std::vector<double> wp((size_t)B * (M + 1) * 3), T((size_t)B * M), bc((size_t)B * 6 * (r - 1), 0.0);
    for (auto& x : wp) x = u(g);
    for (auto& x : T) x = ut(g);
    double *dwp, *dT, *dbc, *dout; int* dst;
    hipMalloc(&dwp, wp.size() * 8); hipMalloc(&dT, T.size() * 8); hipMalloc(&dbc, bc.size() * 8);
    hipMalloc(&dout, (size_t)B * 3 * 2 * r * M * 8); hipMalloc(&dst, (size_t)B * 4);
    hipMemcpy(dwp, wp.data(), wp.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dbc, bc.data(), bc.size() * 8, hipMemcpyHostToDevice);
    { hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("multiProcessorCount=%d clock=%d kHz l2=%d\n", p.multiProcessorCount, p.clockRate, p.l2CacheSize); }
    uavqp_ctx* ctx; if (uavqp_create(&ctx, 0) != 0) { printf("create failed: %s\n", uavqp_last_error()); return 1; }
    hipStream_t s; if (getenv("BLK")) hipStreamCreate(&s); else hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uavqp_set_stream(ctx, s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) uavqp_solve_batch_device(ctx, r, B, M, M, nullptr, dwp, dT, dbc, dout, dst);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < K; ++i) uavqp_solve_batch_device(ctx, r, B, M, M, nullptr, dwp, dT, dbc, dout, dst);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (getenv("STREAMS")) {
        const int NS = atoi(getenv("STREAMS"));
        std::vector<uavqp_ctx*> cs(NS); std::vector<hipStream_t> ss(NS); std::vector<double*> outs(NS);
        for (int i = 0; i < NS; ++i) { uavqp_create(&cs[i], 0); hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking); uavqp_set_stream(cs[i], ss[i]); hipMalloc(&outs[i], (size_t)B * 3 * 2 * r * M * 8); }
        for (int i = 0; i < 4 * NS; ++i) uavqp_solve_batch_device(cs[i % NS], r, B, M, M, nullptr, dwp, dT, dbc, outs[i % NS], dst);
        hipDeviceSynchronize();
        hipEvent_t ev[16]; for (int i = 0; i < NS; ++i) hipEventCreate(&ev[i]);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < K; ++i) uavqp_solve_batch_device(cs[i % NS], r, B, M, M, nullptr, dwp, dT, dbc, outs[i % NS], dst);
        hipDeviceSynchronize();
        auto t1 = std::chrono::high_resolution_clock::now();
        ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
        printf("[%d streams, wall clock] ", NS);
    }
    if (getenv("GRAPH")) {
        void* g = nullptr;
        uavqp_capture_begin(ctx);
        for (int i = 0; i < K; ++i) uavqp_solve_batch_device(ctx, r, B, M, M, nullptr, dwp, dT, dbc, dout, dst);
        if (uavqp_capture_end(ctx, &g) != 0) { printf("capture failed: %s\n", uavqp_last_error()); return 1; }
        uavqp_graph_launch(ctx, g); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        uavqp_graph_launch(ctx, g);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("[graph of %d steps] ", K);
    }
    printf("C ABI B=%d: %.2f us/step  %.3e traj/s  %.2f TB/s algorithmic\n", B, ms * 1e3 / K, B / (ms / K * 1e-3), B * (8.0 * (3 * (M + 1) + M + 6 * (r - 1)) + 8.0 * 3 * 2 * r * M) / (ms / K * 1e-3) / 1e12);
    uavqp_destroy(ctx);
    return 0;
}
