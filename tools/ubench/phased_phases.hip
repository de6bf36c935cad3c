// Phase-level cycle stamps of the twisted kernel (wave 0) + back-to-back launch timing.  Debug tool.
#ifndef NO_STAMPS
#define UAVQP_PHASE_TIMING 1
#endif
#ifndef TILE_
#define TILE_ 32
#endif
#ifndef LPT_
#define LPT_ 2
#endif
#include "../../uav_motion_planning_amd/csrc/uavqp.hip"
#include <vector>
#include <random>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1234567) *p = 1; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, M = 8, r = 4;
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-2, 2), ut(0.5, 2.0);
    std::vector<double> wp((size_t)B * (M + 1) * 3), T((size_t)B * M), bc((size_t)B * 18, 0.0);
    for (auto& x : wp) x = u(g);
    for (auto& x : T) x = ut(g);
    double *dwp, *dT, *dbc, *dout; int* dst; long long* dstamps;
    hipMalloc(&dwp, wp.size() * 8); hipMalloc(&dT, T.size() * 8); hipMalloc(&dbc, bc.size() * 8);
    hipMalloc(&dout, (size_t)B * 192 * 8); hipMalloc(&dst, B * 4); hipMalloc(&dstamps, 64 * 8);
    hipMemcpy(dwp, wp.data(), wp.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dbc, bc.data(), bc.size() * 8, hipMemcpyHostToDevice);
    uavqp::BatchArgs a{};
    a.n_traj = B; a.uniform = M; a.max_segments = M; a.waypoints = dwp; a.times = dT; a.bc = dbc; a.coeff = dout; a.status = dst;
    hipMalloc(&a.dummy, 4096);
#ifdef UAVQP_PHASE_TIMING
    a.stamps = dstamps;
#endif
    const int n_tiles = (B + 15) / 16, grid = n_tiles < 1024 ? n_tiles : 1024;
    hipStream_t s; if (getenv("NB")) hipStreamCreateWithFlags(&s, hipStreamNonBlocking); else hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((uavqp::solve_phased_kernel<4, 8>), dim3(grid), dim3(256), 0, s, a);
        hipStreamSynchronize(s);
        long long st[8]; hipMemcpy(st, dstamps, 64, hipMemcpyDeviceToHost);
        printf("wg0 cycles: P0 load %lld | P1 blocks %lld | P2 chain %lld | P3 rhs %lld | P4 emit %lld | total %lld\n", st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[5] - st[0]);
    }
    const int K = 200;
    hipEventRecord(e0, s);
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL((uavqp::solve_phased_kernel<4, 8>), dim3(grid), dim3(256), 0, s, a);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("twisted<4,8> B=%d: %.2f us/launch back-to-back\n", B, ms * 1e3 / K);
    hipEventRecord(e0, s);
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, (int*)nullptr);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel grid=%d: %.2f us/launch back-to-back\n", grid, ms * 1e3 / K);
    return 0;
}
