// Micro-benchmark: achieved HBM write bandwidth for different store patterns of the coefficient layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// pattern 0: fully contiguous 1 KiB per store instruction
// pattern 1: the twisted kernel's pattern: 64-B chunks (4 lanes x 16 B), L/R lanes write segments j and 7-j
// pattern 2: 128-B lines (8 lanes x 16 B): two adjacent segments per line, 8 lines per instruction
// pattern 3: 512-B rows (32 lanes x 16 B): a whole (traj, axis) row per half wave
template <int P>
__global__ __launch_bounds__(64) void wr(double* out, int n_tiles) {
    const int lane = threadIdx.x;
    const double2 v = make_double2(lane, 1.0);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        double* base = out + (size_t)tile * 32 * 192;  // 32 trajectories x 192 doubles
        if (P == 0) {
#pragma unroll
            for (int k = 0; k < 48; ++k) *reinterpret_cast<double2*>(base + k * 128 + lane * 2) = v;
        } else if (P == 1) {
#pragma unroll
            for (int jj = 3; jj >= 0; --jj)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int pl = ii * 16 + (lane >> 2), q = lane & 3, t = pl >> 1, r = pl & 1;
                        const int seg = r ? 7 - jj : jj;
                        *reinterpret_cast<double2*>(base + ((t * 3 + ax) * 8 + seg) * 8 + 2 * q) = v;
                    }
        } else if (P == 2) {
            // instruction (pair p = 0..3, ax, ii 0..3): 8 lanes per 128-B line (segments 2p, 2p+1), 8 trajectories
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int t = ii * 8 + (lane >> 3), o = lane & 7;
                        *reinterpret_cast<double2*>(base + ((t * 3 + ax) * 8 + 2 * p) * 8 + 2 * o) = v;
                    }
        } else if (P == 4) {
            // pattern 1 chunks, but the two 64-B halves of a 128-B line are stored back-to-back
#pragma unroll
            for (int p = 1; p >= 0; --p)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int pl = ii * 16 + (lane >> 2), q = lane & 3, t = pl >> 1, r = pl & 1;
                            const int jj = 2 * p + hh;
                            const int seg = r ? 7 - jj : jj;
                            *reinterpret_cast<double2*>(base + ((t * 3 + ax) * 8 + seg) * 8 + 2 * q) = v;
                        }
        } else if (P == 5) {
            // min-jerk-like: 48-B chunks (3 lanes of 4 active), rows of 16 segments x 48 B = 768 B, tile = 16 traj
#pragma unroll
            for (int jj = 7; jj >= 0; --jj)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int pl = ii * 16 + (lane >> 2), q = lane & 3, t = pl >> 1, r = pl & 1;
                        const int seg = r ? 15 - jj : jj;
                        if (q < 3) *reinterpret_cast<double2*>(base + ((t * 3 + ax) * 16 + seg) * 6 + 2 * q) = v;
                    }
        } else if (P == 6) {
            // min-jerk M=16 pair mode, axis-major: 48-B chunks, two adjacent chunks (96 B) back-to-back; tile = 32 traj x 288 doubles
            double* b6 = out + (size_t)tile * 32 * 288;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                for (int pp = 3; pp >= 0; --pp)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int pl = ii * 16 + (lane >> 2), q = lane & 3, t = pl >> 1, r = pl & 1;
                            const int j1 = 2 * pp + 1, j0 = 2 * pp;
                            const int seg = (r ? 15 - j1 : j0) + hh;
                            if (q < 3) *reinterpret_cast<double2*>(b6 + ((t * 3 + ax) * 16 + seg) * 6 + 2 * q) = v;
                        }
        } else if (P == 7) {
            // min-jerk M=16 row mode: per axis the 32 rows of 768 B are written linearly (48 pieces of 16 B per row)
            double* b7 = out + (size_t)tile * 32 * 288;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                for (int it = 0; it < 24; ++it) {
                    const int g = it * 64 + lane, row = g / 48, col = g % 48;
                    *reinterpret_cast<double2*>(b7 + ((row * 3 + ax) * 16) * 6 + 2 * col) = v;
                }
        } else {
#pragma unroll
            for (int k = 0; k < 48; ++k) {
                const int row = k * 2 + (lane >> 5), o = lane & 31;  // 96 rows of 512 B
                *reinterpret_cast<double2*>(base + row * 64 + 2 * o) = v;
            }
        }
    }
}

template <int P>
void run(double* out, int n_tiles, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wr<P>, dim3(grid), dim3(64), 0, 0, out, n_tiles);
    hipEventRecord(a);
    const int K = 20;
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(wr<P>, dim3(grid), dim3(64), 0, 0, out, n_tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n_tiles * 32 * (P >= 6 ? 288 : 192) * 8;
    printf("pattern %d grid %5d: %.1f us  %.2f TB/s\n", P, grid, ms * 1e3 / K, bytes / (ms / K * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int n_traj = 1 << 20, n_tiles = n_traj / 32;
    double* out; hipMalloc(&out, (size_t)n_traj * 192 * 8);
    for (int grid : {1024, 4096}) {
        run<0>(out, n_tiles, grid); run<1>(out, n_tiles, grid); run<2>(out, n_tiles, grid); run<4>(out, n_tiles, grid); run<5>(out, n_tiles, grid); run<6>(out, n_tiles * 2 / 3, grid); run<7>(out, n_tiles * 2 / 3, grid);
    }
    return 0;
}
