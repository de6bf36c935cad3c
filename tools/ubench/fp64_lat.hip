// Micro-benchmark: FP64 FMA dependent-chain latency / issue rate, LDS round trip, s_memtime clock, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void fma_chain(double* out, int iters, long long* cyc) {
    double a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 1e-9 + i;
    double m = 1.0000001, c = 1e-9;
    long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) a[i] = fma(a[i], m, c);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void lds_chain(double* out, int iters, long long* cyc) {
    __shared__ double s[64 * 8];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    double v = 0;
    for (int k = 0; k < iters; ++k) {
        v = s[idx];
        idx = ((int)v + 1) & 63;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

__global__ void rcp_chain(double* out, int iters, long long* cyc) {
    double a = 1.5 + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; ++k) a = __builtin_amdgcn_rcp(a) + 1.0;
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <typename F>
float time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    long long h;
    const int iters = 100000;
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc); });
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("dep FMA f64 (1 wave, ILP1): %.2f ns/fma  counter %.2f ticks/fma  (%.3f ms)\n", ms * 1e6 / iters, (double)h / iters, ms);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain<2>, dim3(1), dim3(64), 0, 0, out, iters, cyc); });
        printf("ILP2: %.2f ns/fma\n", ms * 1e6 / iters / 2);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain<4>, dim3(1), dim3(64), 0, 0, out, iters, cyc); });
        printf("ILP4: %.2f ns/fma\n", ms * 1e6 / iters / 4);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain<8>, dim3(1), dim3(64), 0, 0, out, iters, cyc); });
        printf("ILP8: %.2f ns/fma\n", ms * 1e6 / iters / 8);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain<8>, dim3(256 * 8), dim3(64), 0, 0, out, iters, cyc); });
        double fl = 2.0 * 8 * iters * 64.0 * 256 * 8;
        printf("chip ILP8 x 2 waves/SIMD: %.3f ms  -> %.1f TFLOP/s f64\n", ms, fl / ms / 1e9);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(lds_chain, dim3(1), dim3(64), 0, 0, out, iters, cyc); });
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("dep LDS read round trip: %.2f ns  counter %.2f ticks\n", ms * 1e6 / iters, (double)h / iters);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(rcp_chain, dim3(1), dim3(64), 0, 0, out, iters, cyc); });
        printf("dep rcp_f64+add: %.2f ns\n", ms * 1e6 / iters);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain<1>, dim3(1), dim3(64), 0, 0, out, 1, cyc); });
        printf("empty-ish kernel launch+run: %.2f us\n", ms * 1e3);
    }
    return 0;
}
