// v_rcp_f64 + n Newton steps against IEEE division: worst error in ulps over 2^24 arguments per decade.  hipcc --offload-arch=gfx950 rcp_accuracy.hip -o rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
__global__ void k(const double* x, int n, double* err) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i], exact = 1.0 / v;
    double r = __builtin_amdgcn_rcp(v);
    const double e0 = fabs(r - exact) / (fabs(exact) * 1.1102230246251565e-16);
    r = fma(fma(-v, r, 1.0), r, r);
    const double e1 = fabs(r - exact) / (fabs(exact) * 1.1102230246251565e-16);
    r = fma(fma(-v, r, 1.0), r, r);
    const double e2 = fabs(r - exact) / (fabs(exact) * 1.1102230246251565e-16);
    err[3 * i] = e0; err[3 * i + 1] = e1; err[3 * i + 2] = e2;
}
int main() {
    const int n = 1 << 22;
    double *hx = new double[n], *he = new double[3 * n], *dx, *de;
    uint64_t s = 88172645463325252ull;
    double w[3] = {0, 0, 0};
    hipMalloc(&dx, n * 8); hipMalloc(&de, 3 * n * 8);
    for (int dec = -6; dec <= 6; dec += 3) {
        for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = std::pow(10.0, dec) * (1.0 + (double)(s >> 11) / 9007199254740992.0 * 9.0); }
        hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, de);
        hipMemcpy(he, de, 3 * n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) w[j] = he[3 * i + j] > w[j] ? he[3 * i + j] : w[j];
    }
    printf("worst error in units of 2^-53 relative (0.5 ulp): raw v_rcp_f64 %.3g, one Newton step %.3g, two steps %.3g\n", w[0], w[1], w[2]);
    return 0;
}
