// Back-to-back launch cost of an (almost) empty kernel as a function of grid shape: does the number of workgroups matter
// at the headline batch's scale (512 single-wave workgroups)?   hipcc -O3 --offload-arch=gfx950 launch_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(double* p) { if (p && threadIdx.x == 63 && blockIdx.x == 0) p[0] = 1.0; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double* d; hipMalloc(&d, 8);
    const int shapes[][2] = {{1, 64}, {64, 64}, {128, 64}, {256, 64}, {512, 64}, {1024, 64}, {2048, 64}, {128, 256}, {256, 256}, {512, 256}, {64, 512}, {128, 512}, {64, 1024}};
    for (auto& sh : shapes) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(tiny, dim3(sh[0]), dim3(sh[1]), 0, s, d);
        hipStreamSynchronize(s);
        const int K = 500;
        hipEventRecord(e0, s);
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, dim3(sh[0]), dim3(sh[1]), 0, s, d);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d x block %4d (%6d waves): %.2f us/launch\n", sh[0], sh[1], sh[0] * sh[1] / 64, ms * 1e3 / K);
    }
    return 0;
}
