// Floor of a 4096-trajectory launch: same grid / bytes as the latency shape (512 single-wave workgroups, 3.4 KB in,
// 12 KB out each), no arithmetic.  Back-to-back launches on one stream, like bench.py.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void io_only(const double* in, double* out, int mode) {
    __shared__ double s[512];
    const int lane = threadIdx.x, wg = blockIdx.x;
    const double* g = in + (size_t)wg * 424;   // 8 trajectories x 53 doubles
    double acc = 0;
    if (mode >= 1) {
        for (int i = lane; i < 424; i += 64) s[i] = g[i];
        __syncthreads();
        acc = s[lane] + s[(lane * 7) % 424];
    }
    double* o = out + (size_t)wg * 1536;       // 8 trajectories x 192 doubles
    if (mode >= 2)
        for (int k = 0; k < 12; ++k) *reinterpret_cast<double2*>(o + k * 128 + lane * 2) = make_double2(acc, (double)k);
    else if (lane == 0) o[0] = acc;
}
int main() {
    double *in, *out; hipMalloc(&in, 4096 * 53 * 8); hipMalloc(&out, 4096 * 192 * 8);
    hipMemset(in, 0, 4096 * 53 * 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(io_only, dim3(512), dim3(64), 0, s, in, out, mode);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        const int K = 1000;
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(io_only, dim3(512), dim3(64), 0, s, in, out, mode);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.2f us/launch\n", mode, mode == 0 ? "empty" : mode == 1 ? "load only" : "load + 6.3 MB store", ms * 1e3 / K);
    }
    return 0;
}
