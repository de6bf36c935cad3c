// obstacle_grid.h -- uniform grid over an obstacle point cloud and the SE(3) ellipsoid check that queries it.
//
// Replaces the kd-tree radius search of KinoAstar::isCollisionFree (src/planner/path_searching/src/kino_astar.cpp:747-750:
// kdtree radiusSearch(pt, robot_r + 0.1)) for batched queries: points are bucketed by cell (counting sort, cell >= the
// search radius by default so that a query visits 27 cells), the query walks the cells overlapping the radius box and
// applies exactly the tests of ellipsoid_kernel (uavqp.hip) -- same candidate set, same arithmetic per pair, so the
// flags are identical to the exhaustive scan.
#pragma once
#include "qp_core_kernels.h"
#include "qp_wave_utils.h"

namespace uavqp {

struct GridView {
    double org[3];     // lower corner of cell (0,0,0)
    double inv_cell;   // 1 / cell size
    int dim[3];        // cells per axis
    const int32_t* cell_start;  // [ncell + 1] exclusive prefix of the per-cell counts
    const double* pts;          // [n_obs][3] sorted by cell
};

__device__ __forceinline__ int grid_coord(const GridView& g, double x, int ax) {
    int c = (int)floor((x - g.org[ax]) * g.inv_cell);
    return c < 0 ? 0 : (c >= g.dim[ax] ? g.dim[ax] - 1 : c);
}

// per-block partial bounds: out[block][6] = min xyz, max xyz
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ void grid_bounds_kernel(const double* __restrict__ obs, int n_obs, double* __restrict__ out) {
    __shared__ double s[256][6];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_obs; i += gridDim.x * 256)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double v = obs[(size_t)i * 3 + ax];
            if (fabs(v) < INFINITY) {
                mn[ax] = fmin(mn[ax], v);
                mx[ax] = fmax(mx[ax], v);
            } else {  // NaN / inf: poison the bounds so that the host rejects the cloud
                mn[ax] = -INFINITY;
                mx[ax] = INFINITY;
            }
        }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { s[threadIdx.x][ax] = mn[ax]; s[threadIdx.x][3 + ax] = mx[ax]; }
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                s[threadIdx.x][ax] = fmin(s[threadIdx.x][ax], s[threadIdx.x + d][ax]);
                s[threadIdx.x][3 + ax] = fmax(s[threadIdx.x][3 + ax], s[threadIdx.x + d][3 + ax]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out[blockIdx.x * 6 + threadIdx.x] = s[0][threadIdx.x];
}
#endif

#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ void grid_count_kernel(GridView g, const double* __restrict__ obs, int n_obs, int32_t* __restrict__ cell_of,
                                  int32_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_obs) return;
    const int cx = grid_coord(g, obs[(size_t)i * 3], 0), cy = grid_coord(g, obs[(size_t)i * 3 + 1], 1),
              cz = grid_coord(g, obs[(size_t)i * 3 + 2], 2);
    const int c = (cz * g.dim[1] + cy) * g.dim[0] + cx;
    cell_of[i] = c;
    atomicAdd(&counts[c], 1);
}
#endif

#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ void grid_scatter_kernel(const double* __restrict__ obs, int n_obs, const int32_t* __restrict__ cell_of,
                                    int32_t* __restrict__ cursor, double* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_obs) return;
    const int pos = atomicAdd(&cursor[cell_of[i]], 1);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) sorted[(size_t)pos * 3 + ax] = obs[(size_t)i * 3 + ax];
}
#endif

// ---------------------------------------------------------------------------------------------------
// Corridor boxes from an obstacle cloud (include/uavqp.h: uavqp_corridor_from_cloud_device).
// One lane per waypoint row.  The ellipsoid metric is evaluated as d' Q d with Q = sum_j b_j b_j' / s_j^2 (6 unique
// entries in registers): 12 FP64 operations per (row, obstacle) pair, no early exit (it is a min); the cloud streams
// through LDS in 1024-point tiles (one 24-byte broadcast read per pair).
// A grid-pruned variant (scan a box around the row, widen once to g * max(robot_r, robot_h): exact, bit-identical) was
// built and measured 9x SLOWER on config 5 (25.6 vs 2.8 ms): the flat axis of the robot ellipsoid (robot_h = 0.1) makes
// the clearance of a free waypoint reach 10-25, i.e. a Euclidean radius of 4-10 m in a 40 x 20 m map -- nearly every
// cell, visited with divergent per-lane loops instead of LDS broadcasts.  The grid pays for the radius-limited
// collision check below (71x), not here.
// ---------------------------------------------------------------------------------------------------
struct CloudCorridorArgs {
    int n_traj, uniform, n_rows, n_obs;
    const int32_t* seg_offsets;
    const double* waypoints;
    const double* times;
    const double* coeff;
    const double* obs;
    double robot_r, robot_h, h_max;
    double* lo;
    double* hi;
    double* clearance;
    // window variant (cloud_window_kernel): rows and points sorted along one axis, see CloudSort
    const struct CloudSort* sort;
    const int32_t* row_perm;    // [n_rows] row ids in bin order
    const int32_t* row_start;   // [CLOUD_ROW_BINS + 1]
    const int32_t* pt_start;    // [CLOUD_PT_BINS + 1]
    const double* pts_sorted;   // [n_obs][3]
    double reach;
};

// v_min_f64 as is: fmin() makes the compiler quiet a possible signalling NaN of the loop-carried operand first (one v_max_f64 x, x per
// call); the scan's operands are sums of squares of finite numbers or +inf -- and a NaN coordinate would be dropped by the IEEE
// minimum either way.
__device__ __forceinline__ double min_nn(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int R>
struct CorridorRow {
    double p[3];
    double qxx, qyy, qzz;                          // diagonal of Q = E^-T E^-1 = sum_j b_j b_j' / s_j^2 (box widths)
    double u00, u01, u02, u11, u12, u22, c0, c1, c2;   // Q = U'U, c = U p (the metric of the scan)
    double b3v[3];                                 // body z axis: (Q^-1)_aa = r^2 - (r^2 - h^2) b3_a^2
    bool interior;

    // position, attitude (kino_astar.cpp:724-737) and quadratic form of waypoint row g
    __device__ __forceinline__ void setup(const CloudCorridorArgs& a, long long g) {
        constexpr int NC = 2 * R;
        int b, k, M, s0;
        if (a.uniform > 0) {
            M = a.uniform;
            b = (int)(g / (M + 1));
            k = (int)(g - (long long)b * (M + 1));
            s0 = b * M;
        } else {
            // row g belongs to the trajectory b with seg_offsets[b] + b <= g < seg_offsets[b+1] + b + 1
            int lo_b = 0, hi_b = a.n_traj - 1;
            while (lo_b < hi_b) {
                const int mid = (lo_b + hi_b + 1) >> 1;
                if ((long long)a.seg_offsets[mid] + mid <= g) lo_b = mid; else hi_b = mid - 1;
            }
            b = lo_b;
            s0 = a.seg_offsets[b];
            M = a.seg_offsets[b + 1] - s0;
            k = (int)(g - ((long long)s0 + b));
        }
        interior = (k > 0) && (k < M);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) p[ax] = a.waypoints[(size_t)g * 3 + ax];
        double acc[3] = {0.0, 0.0, 0.0};
        if (a.coeff && M >= 1) {  // a zero-segment trajectory (flagged invalid by the solver) has no polynomial: hover attitude
            // acceleration at the knot: start of segment k (2 c_2), or the end of the last segment for k = M
            const int seg = k < M ? k : M - 1;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double* ca = a.coeff + (size_t)3 * NC * s0 + ((size_t)ax * M + seg) * NC;
                if (k < M) {
                    acc[ax] = 2.0 * ca[2];
                } else {
                    const double t = a.times[s0 + seg];
                    double av = 0.0;
#pragma unroll
                    for (int j = NC - 1; j >= 2; --j) av = fma(av, t, (double)(j * (j - 1)) * ca[j]);
                    acc[ax] = av;
                }
            }
        }
        // kino_astar.cpp:724-727
        const double n3 = sqrt(acc[0] * acc[0] + acc[1] * acc[1] + (acc[2] + 9.81) * (acc[2] + 9.81));
        const double b3[3] = {acc[0] / n3, acc[1] / n3, (acc[2] + 9.81) / n3};
        const double c2y = b3[2], c2z = -b3[1];  // b3 x (1,0,0) = (0, b3z, -b3y)
        const double n2 = sqrt(c2y * c2y + c2z * c2z);
        const double b2[3] = {0.0, c2y / n2, c2z / n2};
        double b1[3] = {b2[1] * b3[2] - b2[2] * b3[1], b2[2] * b3[0] - b2[0] * b3[2], b2[0] * b3[1] - b2[1] * b3[0]};
        const double n1 = sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
        b1[0] /= n1; b1[1] /= n1; b1[2] /= n1;
        b3v[0] = b3[0]; b3v[1] = b3[1]; b3v[2] = b3[2];
        const double wr = 1.0 / (a.robot_r * a.robot_r), wh = 1.0 / (a.robot_h * a.robot_h);
        qxx = (b1[0] * b1[0] + b2[0] * b2[0]) * wr + b3[0] * b3[0] * wh;
        qyy = (b1[1] * b1[1] + b2[1] * b2[1]) * wr + b3[1] * b3[1] * wh;
        qzz = (b1[2] * b1[2] + b2[2] * b2[2]) * wr + b3[2] * b3[2] * wh;
        const double qxy = (b1[0] * b1[1] + b2[0] * b2[1]) * wr + b3[0] * b3[1] * wh;
        const double qxz = (b1[0] * b1[2] + b2[0] * b2[2]) * wr + b3[0] * b3[2] * wh;
        const double qyz = (b1[1] * b1[2] + b2[1] * b2[2]) * wr + b3[1] * b3[2] * wh;
        // Q is SPD with condition (robot_r / robot_h)^2: plain Cholesky
        u00 = sqrt(qxx);
        u01 = qxy / u00;
        u02 = qxz / u00;
        u11 = sqrt(qyy - u01 * u01);
        u12 = (qyz - u01 * u02) / u11;
        u22 = sqrt(qzz - u02 * u02 - u12 * u12);
        c0 = u00 * p[0] + u01 * p[1] + u02 * p[2];
        c1 = u11 * p[1] + u12 * p[2];
        c2 = u22 * p[2];
    }
    // |E^-1 (o - p)|^2 = |U o - U p|^2 with Q = U'U (Cholesky, U upper triangular): six FMAs for y = U o - c, three for y'y --
    // 9 FP64 instructions + the min per (row, point) pair instead of 12 + 1 for d = o - p, d'Qd.  The subtraction happens after the
    // products, so the absolute rounding error of y is that of |U| |o| ~ 10 x 40 m: ~1e-13, i.e. 1e-13 relative on a clearance >= 1
    // (the only range the box uses: h = (g - 1) / ...).
    __device__ __forceinline__ double metric2(double ox, double oy, double oz) const {
        const double y0 = fma(u00, ox, fma(u01, oy, fma(u02, oz, -c0)));
        const double y1 = fma(u11, oy, fma(u12, oz, -c1));
        const double y2 = fma(u22, oz, -c2);
        return fma(y0, y0, fma(y1, y1, y2 * y2));
    }
    // No point of a cloud inside the axis-aligned box [bb_lo, bb_hi] can change this row's box: for any offset d with d_a = t,
    // d'Qd >= t^2 / (Q^-1)_aa (Cauchy-Schwarz), so the clearance is at least  g_lb = max_a dist_a(p, box) / sqrt((Q^-1)_aa),  and from
    // g_cap = 1 + 3 h_max max_i sqrt(Q_ii) on every half-width is h_max whatever the clearance is.  Exact, with a relative slack of 1e-9.
    __device__ __forceinline__ bool culled_by_box(const CloudCorridorArgs& a, const double (&bb_lo)[3], const double (&bb_hi)[3]) const {
        const double r2 = a.robot_r * a.robot_r, h2 = a.robot_h * a.robot_h;
        double glb2 = 0.0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double d = fmax(fmax(bb_lo[ax] - p[ax], p[ax] - bb_hi[ax]), 0.0);
            const double qi = r2 - (r2 - h2) * b3v[ax] * b3v[ax];
            glb2 = fmax(glb2, d * d / qi);
        }
        const double gcap = 1.0 + 3.0 * a.h_max * sqrt(fmax(qxx, fmax(qyy, qzz)));
        return glb2 >= gcap * gcap * (1.0 + 1e-9);      // (NaN anywhere: not culled)
    }
    // clearance g = sqrt(min metric^2) -> box and outputs
    __device__ __forceinline__ void emit(const CloudCorridorArgs& a, long long g, double min2) const {
        const double gmin = sqrt(fmax(min2, 0.0));
        if (a.clearance) a.clearance[g] = gmin;
        const double margin = gmin > 1.0 ? gmin - 1.0 : 0.0;
        const double q[3] = {qxx, qyy, qzz};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double h = 0.0;
            if (interior) {
                h = margin / (3.0 * sqrt(q[ax]));
                h = h < a.h_max ? h : a.h_max;  // also maps margin = inf (empty cloud) to h_max
            }
            a.lo[(size_t)g * 3 + ax] = p[ax] - h;
            a.hi[(size_t)g * 3 + ax] = p[ax] + h;
        }
    }
};

template <int R>
__global__ __launch_bounds__(256) void cloud_corridor_kernel(CloudCorridorArgs a) {
    constexpr int TILE = 1024;
    __shared__ double s_obs[TILE * 3];
    const long long n_round = ((long long)a.n_rows + 255) / 256 * 256;  // every thread of a block joins the LDS tile loads
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n_round; g += (long long)gridDim.x * 256) {
        const bool live = g < a.n_rows;
        CorridorRow<R> row;
        if (live) row.setup(a, g);
        double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;  // four independent min chains
        for (int o0 = 0; o0 < a.n_obs; o0 += TILE) {
            const int nt = min(TILE, a.n_obs - o0);
            __syncthreads();
            for (int i = threadIdx.x; i < nt * 3; i += 256) s_obs[i] = a.obs[(size_t)o0 * 3 + i];
            __syncthreads();
            if (live) {
                int i = 0;
                for (; i + 3 < nt; i += 4) {
                    const double* o = s_obs + 3 * i;   // 96 contiguous bytes: six 16-byte LDS broadcasts
                    m0 = min_nn(m0, row.metric2(o[0], o[1], o[2]));
                    m1 = min_nn(m1, row.metric2(o[3], o[4], o[5]));
                    m2 = min_nn(m2, row.metric2(o[6], o[7], o[8]));
                    m3 = min_nn(m3, row.metric2(o[9], o[10], o[11]));
                }
                for (; i < nt; ++i) m0 = min_nn(m0, row.metric2(s_obs[3 * i], s_obs[3 * i + 1], s_obs[3 * i + 2]));
            }
        }
        if (live) row.emit(a, g, fmin(fmin(m0, m1), fmin(m2, m3)));
    }
}

// ---------------------------------------------------------------------------------------------------
// Window variant (round 3).  A point farther from a waypoint than  reach = max(r, h) (1 + 3 h_max / min(r, h))  cannot change its
// box: the metric |E^-1 d| is at least |d| / max(r, h), and once the clearance g exceeds 1 + 3 h_max max_i |E^-1 e_i|  (<= the
// g of that distance) every half-width is capped at h_max whatever g is.  So rows and points are counting-sorted along ONE axis (the
// longest of the cloud's bounding box), a block takes 256 rows that are neighbours along it and scans only the points within
// `reach` of the block's interval -- still through LDS broadcasts, nothing diverges; the boxes are bit-identical to the exhaustive
// scan (same per-pair arithmetic, min over a superset of the points that matter).  Config 5 (40 m x 20 m map, reach 10 m):
// 44 % of the pairs.  Not used when the caller wants the clearance itself (an exact min over the whole cloud).
// ---------------------------------------------------------------------------------------------------
constexpr int CLOUD_PT_BINS = 1024, CLOUD_ROW_BINS = 4096;
struct CloudSort {
    int axis;
    double p_lo, p_inv;   // point bins over the cloud's extent along `axis`
    double r_lo, r_inv, r_w;   // row bins over [p_lo - reach, p_hi + reach], bin width r_w
    double bb_lo[3], bb_hi[3]; // bounding box of the finite points: rows it proves capped (CorridorRow::culled_by_box) go to row bin CLOUD_ROW_BINS and are never scanned
};
__device__ __forceinline__ int cloud_bin(double v, double lo, double inv, int nb) {
    const double t = (v - lo) * inv;
    int b = t >= 0.0 ? (t < (double)nb ? (int)t : nb - 1) : 0;   // NaN -> 0
    return b;
}
// one block: bounding box of the cloud -> sort axis and bin geometry; zeroes the histograms
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void cloud_sort_setup_kernel(const double* __restrict__ obs, int n_obs, double reach, CloudSort* __restrict__ cs,
                                                                int32_t* __restrict__ pt_hist, int32_t* __restrict__ row_hist) {
    __shared__ double s[1024][6];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n_obs; i += 1024)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double v = obs[(size_t)i * 3 + ax];
            if (fabs(v) < INFINITY) { mn[ax] = fmin(mn[ax], v); mx[ax] = fmax(mx[ax], v); }
        }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { s[threadIdx.x][ax] = mn[ax]; s[threadIdx.x][3 + ax] = mx[ax]; }
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                s[threadIdx.x][ax] = fmin(s[threadIdx.x][ax], s[threadIdx.x + d][ax]);
                s[threadIdx.x][3 + ax] = fmax(s[threadIdx.x][3 + ax], s[threadIdx.x + d][3 + ax]);
            }
        __syncthreads();
    }
    for (int i = threadIdx.x; i <= CLOUD_PT_BINS; i += 1024) pt_hist[i] = 0;
    for (int i = threadIdx.x; i <= CLOUD_ROW_BINS + 1; i += 1024) row_hist[i] = 0;
    if (threadIdx.x == 0) {
        int axis = 0;
        double ext = -1.0;
        for (int ax = 0; ax < 3; ++ax) {
            const double e = s[0][3 + ax] - s[0][ax];
            if (e > ext) { ext = e; axis = ax; }   // (no finite point at all: ext stays -1, axis 0, everything lands in bin 0)
        }
        double lo = s[0][axis], hi = s[0][3 + axis];
        if (!(hi > lo)) { lo = (fabs(lo) < INFINITY) ? lo : 0.0; hi = lo + 1.0; }
        cs->axis = axis;
        cs->p_lo = lo;
        cs->p_inv = (double)CLOUD_PT_BINS / (hi - lo);
        cs->r_lo = lo - reach;
        cs->r_w = (hi - lo + 2.0 * reach) / (double)CLOUD_ROW_BINS;
        cs->r_inv = 1.0 / cs->r_w;
        for (int ax = 0; ax < 3; ++ax) { cs->bb_lo[ax] = s[0][ax]; cs->bb_hi[ax] = s[0][3 + ax]; }     // (+inf / -inf without a finite point: every row is culled -- no point, every box h_max)
    }
}
#endif
// histograms of the points and of the rows (one launch), bins at [1..]: hist[b + 1] counts bin b, so that the scan leaves starts.
// Counted in LDS first (neighbouring rows share bins: global atomics on the same address serialise -- 77 us for 266 k keys), one
// global add per block and non-empty bin.  Every block works on ONE contiguous slice of the keys, the same slice in the scatter.
constexpr int CLOUD_BINS_ALL = CLOUD_PT_BINS + CLOUD_ROW_BINS + 1;     // (+ 1: the bin of the rows the cloud's bounding box proves capped)
__device__ __forceinline__ void cloud_slice(long long total, long long& i0, long long& i1) {
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    i0 = (long long)blockIdx.x * per;
    i1 = i0 + per < total ? i0 + per : total;
}
template <int R>
__global__ __launch_bounds__(256) void cloud_sort_hist_kernel(CloudCorridorArgs a, const CloudSort* __restrict__ csp, int32_t* __restrict__ pt_hist,
                                                              int32_t* __restrict__ row_hist, int32_t* __restrict__ row_bin) {
    __shared__ int s_h[CLOUD_BINS_ALL];
    for (int i = threadIdx.x; i < CLOUD_BINS_ALL; i += 256) s_h[i] = 0;
    __syncthreads();
    const CloudSort cs = *csp;
    const int axis = cs.axis, n_obs = a.n_obs;
    long long i0, i1;
    cloud_slice((long long)n_obs + a.n_rows, i0, i1);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (i < n_obs) atomicAdd(&s_h[cloud_bin(a.obs[(size_t)i * 3 + axis], cs.p_lo, cs.p_inv, CLOUD_PT_BINS)], 1);
        else {
            CorridorRow<R> row;
            row.setup(a, i - n_obs);
            const int b = row.culled_by_box(a, cs.bb_lo, cs.bb_hi) ? CLOUD_ROW_BINS : cloud_bin(row.p[axis], cs.r_lo, cs.r_inv, CLOUD_ROW_BINS);
            row_bin[i - n_obs] = b;
            atomicAdd(&s_h[CLOUD_PT_BINS + b], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CLOUD_BINS_ALL; i += 256) {
        const int c = s_h[i];
        if (c) atomicAdd(i < CLOUD_PT_BINS ? &pt_hist[1 + i] : &row_hist[1 + i - CLOUD_PT_BINS], c);
    }
}
// inclusive scans in place (start[b] = first element of bin b, start[NB] = total) and cursor copies; one block
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void cloud_sort_scan_kernel(int32_t* __restrict__ pt_start, int32_t* __restrict__ pt_cursor, int32_t* __restrict__ row_start,
                                                               int32_t* __restrict__ row_cursor) {
    __shared__ int s_tot[1024];
    auto scan = [&](int32_t* a, int32_t* cur, int nb) {   // nb + 1 entries, a[0] = 0; per thread a contiguous run
        const int per = (nb + 1 + 1023) / 1024, b0 = threadIdx.x * per;
        int run = 0;
        for (int k = 0; k < per; ++k) if (b0 + k <= nb) run += a[b0 + k];
        s_tot[threadIdx.x] = run;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int v = (int)threadIdx.x >= d ? s_tot[threadIdx.x - d] : 0;
            __syncthreads();
            s_tot[threadIdx.x] += v;
            __syncthreads();
        }
        int acc = threadIdx.x ? s_tot[threadIdx.x - 1] : 0;
        for (int k = 0; k < per; ++k)
            if (b0 + k <= nb) { acc += a[b0 + k]; a[b0 + k] = acc; cur[b0 + k] = acc; }
        __syncthreads();
    };
    scan(pt_start, pt_cursor, CLOUD_PT_BINS);
    scan(row_start, row_cursor, CLOUD_ROW_BINS + 1);
}
#endif
// scatter: the block counts its slice per bin in LDS again, reserves a range per non-empty bin with ONE global add, and hands out the
// positions inside the ranges with LDS atomics (the order inside a bin is arbitrary: it decides which lane scans a row / where in a
// tile a point sits, never a result -- the minimum over a set does not depend on the order)
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void cloud_sort_scatter_kernel(const double* __restrict__ obs, int n_obs, const int32_t* __restrict__ row_bin, int n_rows,
                                                                 const CloudSort* __restrict__ cs, int32_t* __restrict__ pt_cursor,
                                                                 int32_t* __restrict__ row_cursor, double* __restrict__ pts_sorted,
                                                                 int32_t* __restrict__ row_perm) {
    __shared__ int s_h[CLOUD_BINS_ALL];
    for (int i = threadIdx.x; i < CLOUD_BINS_ALL; i += 256) s_h[i] = 0;
    __syncthreads();
    const int axis = cs->axis;
    long long i0, i1;
    cloud_slice((long long)n_obs + n_rows, i0, i1);
    auto bin_of = [&](long long i) -> int {
        return i < n_obs ? cloud_bin(obs[(size_t)i * 3 + axis], cs->p_lo, cs->p_inv, CLOUD_PT_BINS) : CLOUD_PT_BINS + row_bin[i - n_obs];
    };
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&s_h[bin_of(i)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < CLOUD_BINS_ALL; i += 256) {
        const int c = s_h[i];
        s_h[i] = c ? atomicAdd(i < CLOUD_PT_BINS ? &pt_cursor[i] : &row_cursor[i - CLOUD_PT_BINS], c) : 0;   // first position of the block's range
    }
    __syncthreads();
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int pos = atomicAdd(&s_h[bin_of(i)], 1);
        if (i < n_obs) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) pts_sorted[(size_t)pos * 3 + ax] = obs[(size_t)i * 3 + ax];
        } else {
            row_perm[pos] = (int)(i - n_obs);
        }
    }
}
#endif

// PS waves share the rows of a block and split the points of every tile between them (PS = 2: 128 rows per 256-thread block): the number of
// waves of the launch is rows / 64 x PS -- config 5 has only ~2200 row-waves for 1024 SIMDs, two per SIMD, too few to hide the LDS round trip
// in front of every four points (48 % of the FP64 peak); the partial minima meet in LDS at the end.  Same pairs, same arithmetic per pair.
template <int R, int PS = 2>
__global__ __launch_bounds__(256) void cloud_window_kernel(CloudCorridorArgs a) {
    constexpr int TILE = 1024, RPB = 256 / PS;      // rows per block
    __shared__ double s_obs[TILE * 3];
    __shared__ double s_min[256];
    __shared__ int s_win[2];
    const CloudSort cs = *a.sort;
    const int rl = threadIdx.x % RPB, part = threadIdx.x / RPB;      // row of the block, share of the points
    const long long n_round = ((long long)a.n_rows + RPB - 1) / RPB * RPB;
    for (long long g0 = (long long)blockIdx.x * RPB; g0 < n_round; g0 += (long long)gridDim.x * RPB) {
        const long long g = g0 + rl;
        const bool present = g < a.n_rows;
        const int n_scan = a.row_start[CLOUD_ROW_BINS];          // rows [n_scan, n_rows): proven capped by the cloud's bounding box -- emitted without a scan
        const bool live = present && g < n_scan;
        const int rid = present ? a.row_perm[g] : 0;
        CorridorRow<R> row;
        if (present) row.setup(a, rid);
        if (g0 >= n_scan) {                                       // (block-uniform: nothing of this block is scanned)
            if (present && part == 0) row.emit(a, rid, INFINITY);
            continue;
        }
        __syncthreads();   // (s_win / s_min of the previous round have been read by everybody)
        if (threadIdx.x == 0) {
            // the block's rows are consecutive in bin order: bins [kb_lo, kb_hi], found by binary search over the bin starts;
            // one bin of slack either side covers the rounding of the bin function, the outermost bins are open-ended
            const int gl = (int)g0, gh = (int)((g0 + RPB - 1 < n_scan ? g0 + RPB - 1 : n_scan - 1));
            auto bin_of = [&](int idx) -> int {   // largest b with row_start[b] <= idx
                int lo_b = 0, hi_b = CLOUD_ROW_BINS - 1;
                while (lo_b < hi_b) {
                    const int mid = (lo_b + hi_b + 1) >> 1;
                    if (a.row_start[mid] <= idx) lo_b = mid; else hi_b = mid - 1;
                }
                return lo_b;
            };
            const int kb_lo = bin_of(gl), kb_hi = bin_of(gh);
            const double x_lo = kb_lo <= 0 ? -INFINITY : cs.r_lo + (double)(kb_lo - 1) * cs.r_w;
            const double x_hi = kb_hi >= CLOUD_ROW_BINS - 1 ? INFINITY : cs.r_lo + (double)(kb_hi + 2) * cs.r_w;
            const int pb_lo = cloud_bin(x_lo - a.reach, cs.p_lo, cs.p_inv, CLOUD_PT_BINS);
            const int pb_hi = cloud_bin(x_hi + a.reach, cs.p_lo, cs.p_inv, CLOUD_PT_BINS);
            // (a non-finite coordinate was binned to 0 or the last bin: those two are always inside open-ended windows only, which is
            //  fine -- the IEEE minimum ignores what such a point produces, in the exhaustive scan as here)
            s_win[0] = a.pt_start[pb_lo];
            s_win[1] = a.pt_start[pb_hi + 1];
        }
        __syncthreads();
        const int p0 = s_win[0], p1 = s_win[1];
        double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        for (int o0 = p0; o0 < p1; o0 += TILE) {
            const int nt = min(TILE, p1 - o0);
            __syncthreads();
            for (int i = threadIdx.x; i < nt * 3; i += 256) s_obs[i] = a.pts_sorted[(size_t)o0 * 3 + i];
            __syncthreads();
            if (live) {
                int i = 4 * part;                                 // groups of four points, dealt round-robin to the PS shares
                for (; i + 3 < nt; i += 4 * PS) {
                    const double* o = s_obs + 3 * i;
                    m0 = min_nn(m0, row.metric2(o[0], o[1], o[2]));
                    m1 = min_nn(m1, row.metric2(o[3], o[4], o[5]));
                    m2 = min_nn(m2, row.metric2(o[6], o[7], o[8]));
                    m3 = min_nn(m3, row.metric2(o[9], o[10], o[11]));
                }
                if (part == 0)
                    for (int j = nt & ~3; j < nt; ++j) m0 = min_nn(m0, row.metric2(s_obs[3 * j], s_obs[3 * j + 1], s_obs[3 * j + 2]));
            }
        }
        double mm = fmin(fmin(m0, m1), fmin(m2, m3));
        if (PS > 1) {
            s_min[threadIdx.x] = mm;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PS; ++q) mm = fmin(mm, s_min[q * RPB + rl]);
        }
        if (present && part == 0) row.emit(a, rid, live ? mm : INFINITY);
    }
}

struct EllipsoidGridArgs {
    int n_traj, uniform, n_samples;
    const int32_t* seg_offsets;
    const double* times;
    const double* coeff;
    GridView grid;
    double t0, dt, robot_r, robot_h;
    int32_t* first_hit;
    uint8_t* flags;
    const unsigned long long* tmax_bits;   // optional: dt = (double with these bits) / (n_samples - 1) instead of the field above
};

// One lane per (trajectory, sample), as ellipsoid_kernel; the candidates come from the cells overlapping the axis-aligned box of
// half-width robot_r + 0.1 around the sample.  One wave per workgroup, and the wave POOLS its candidates: a lane next to a pillar has
// 40-odd points to test while the average lane has 2 (measured on BASELINE config 5, tools/ellipsoid_probe.py), and a loop per lane
// costs the wave its longest list.  So every lane leaves its frame (sample position, body axes) and its <= 9 point ranges in LDS, the
// list lengths are scanned across the wave, and the (lane, point) pairs are tested 64 at a time by whichever lane comes by -- the
// verdict of a sample is an OR over its candidates, so neither the order nor the tester matters.  (No early exit inside a list; the
// work saved by it was less than the lanes it idled.)
template <int R>
__global__ __launch_bounds__(64) void ellipsoid_grid_kernel(EllipsoidGridArgs a) {
    constexpr int NC = 2 * R;
    __shared__ double s_frame[12][64];            // [p, b1, b2, b3][lane]
    __shared__ int s_rb[9][64], s_cum[9][64];     // first point of the x-row ranges of a lane, running sum of their lengths
    __shared__ int s_off[64], s_hit[64];          // candidates of the lanes before this one; verdict
    const int lane = threadIdx.x;
    const long long total = (long long)a.n_traj * a.n_samples;
    const long long total_round = (total + 63) / 64 * 64;                           // whole waves take part in the shuffles
    const double dt = a.tmax_bits ? __longlong_as_double((long long)*a.tmax_bits) / (double)(a.n_samples - 1) : a.dt;
    const double rad = a.robot_r + 1e-1, rad2 = rad * rad;
    const double ir = 1.0 / a.robot_r, ih = 1.0 / a.robot_h;
    auto inside = [&](const double* q, const double* f) -> bool {                   // f = p, b1, b2, b3
        const double dx = q[0] - f[0], dy = q[1] - f[1], dz = q[2] - f[2];
        if (dx * dx + dy * dy + dz * dz <= rad2) {  // the reference's radius search (r + 0.1)
            const double e1 = (f[3] * dx + f[4] * dy + f[5] * dz) * ir;
            const double e2 = (f[6] * dx + f[7] * dy + f[8] * dz) * ir;
            const double e3 = (f[9] * dx + f[10] * dy + f[11] * dz) * ih;
            return e1 * e1 + e2 * e2 + e3 * e3 <= 1.0;  // |E^-1 d| <= 1
        }
        return false;
    };
    for (long long g0 = (long long)blockIdx.x * 64; g0 < total_round; g0 += (long long)gridDim.x * 64) {
        const long long g = g0 + lane;
        const bool live = g < total;
        const int b = live ? (int)(g / a.n_samples) : 0;
        const int s = live ? (int)(g - (long long)b * a.n_samples) : 0;
        int s0 = 0, M = 0;
        if (live) {
            if (a.uniform > 0) { M = a.uniform; s0 = b * M; } else { s0 = a.seg_offsets[b]; M = a.seg_offsets[b + 1] - s0; }
        }
        const double* __restrict__ T = a.times + s0;
        double t = a.t0 + s * dt;
        int idx = 0;
        while (idx < M && t > T[idx] + 1e-4) { t -= T[idx]; ++idx; }
        // Past the end the sample is the end point.  All trajectories are sampled on one grid (the pipeline's dt comes from the LONGEST of
        // the batch), so a trajectory of average length has most of its samples there -- and every one after the first repeats that
        // one's verdict at a larger index: it cannot lower first_hit.  Left out when no per-sample flags are asked for.  (Whether the
        // previous sample is past the end is its lane's own finding, one lane down; lane 0 has nobody to ask and is simply tested.)
        const int past = (M >= 1 && idx == M) ? 1 : 0;
        const int past_prev = __shfl_up(past, 1, 64);
        const bool repeats = !a.flags && past && s > 0 && lane > 0 && past_prev;
        const bool act = live && M >= 1 && !repeats;   // (M < 1: zero-segment trajectory, flagged invalid by the solver: reported collision-free)
        bool hit = false;
        int cnt = 0;
        if (act) {
            if (idx == M) { --idx; t = T[idx]; }
            double f[12], acc[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double* ca = a.coeff + (size_t)3 * NC * s0 + ((size_t)ax * M + idx) * NC;
                double pv = 0.0, av = 0.0;
#pragma unroll
                for (int j = NC - 1; j >= 0; --j) pv = fma(pv, t, ca[j]);
#pragma unroll
                for (int j = NC - 1; j >= 2; --j) av = fma(av, t, (double)(j * (j - 1)) * ca[j]);
                f[ax] = pv;
                acc[ax] = av;
            }
            // kino_astar.cpp:724-727
            const double n3 = sqrt(acc[0] * acc[0] + acc[1] * acc[1] + (acc[2] + 9.81) * (acc[2] + 9.81));
            const double b3[3] = {acc[0] / n3, acc[1] / n3, (acc[2] + 9.81) / n3};
            const double c2[3] = {0.0, b3[2], -b3[1]};  // b3 x (1,0,0)
            const double n2 = sqrt(c2[1] * c2[1] + c2[2] * c2[2]);
            const double b2[3] = {0.0, c2[1] / n2, c2[2] / n2};
            const double c1[3] = {b2[1] * b3[2] - b2[2] * b3[1], b2[2] * b3[0] - b2[0] * b3[2], b2[0] * b3[1] - b2[1] * b3[0]};
            const double n1 = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                f[3 + ax] = c1[ax] / n1;
                f[6 + ax] = b2[ax];
                f[9 + ax] = b3[ax];
            }
            int lo_c[3], hi_c[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                lo_c[ax] = grid_coord(a.grid, f[ax] - rad, ax);
                hi_c[ax] = grid_coord(a.grid, f[ax] + rad, ax);
            }
            if (hi_c[2] - lo_c[2] <= 2 && hi_c[1] - lo_c[1] <= 2) {
                // (the usual case: cells no smaller than the search radius, so the box meets at most 3 x 3 x-rows; the cells of one
                // x-row are consecutive: one contiguous range of sorted points)
                int rb[9], re[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int cz = lo_c[2] + k / 3, cy = lo_c[1] + k % 3;
                    const bool valid = cz <= hi_c[2] && cy <= hi_c[1];
                    const int row = (cz * a.grid.dim[1] + cy) * a.grid.dim[0];
                    rb[k] = valid ? a.grid.cell_start[row + lo_c[0]] : 0;
                    re[k] = valid ? a.grid.cell_start[row + hi_c[0] + 1] : 0;
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    cnt += re[k] - rb[k];
                    s_rb[k][lane] = rb[k];
                    s_cum[k][lane] = cnt;
                }
#pragma unroll
                for (int c = 0; c < 12; ++c) s_frame[c][lane] = f[c];
            } else {
                // finer cells than the radius: this lane walks its rows itself
                for (int cz = lo_c[2]; cz <= hi_c[2] && !hit; ++cz)
                    for (int cy = lo_c[1]; cy <= hi_c[1] && !hit; ++cy) {
                        const int row = (cz * a.grid.dim[1] + cy) * a.grid.dim[0];
                        const int beg = a.grid.cell_start[row + lo_c[0]], end = a.grid.cell_start[row + hi_c[0] + 1];
                        for (int i = beg; i < end; ++i)
                            if (inside(a.grid.pts + (size_t)i * 3, f)) { hit = true; break; }
                    }
            }
        }
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc, d, 64);
            if (lane >= d) inc += v;
        }
        const int W = __shfl(inc, 63, 64);
        s_off[lane] = inc - cnt;
        s_hit[lane] = 0;
        wave_lds_sync();
        for (int w0 = 0; w0 < W; w0 += 64) {
            const int item = w0 + lane;
            if (item < W) {
                // owner: the last lane whose offset is <= item (lanes without candidates share the offset of the owner behind them)
                int lo = 0, hi = 63;
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_off[mid] <= item) lo = mid; else hi = mid - 1;
                }
                const int L = lo, j = item - s_off[L];
                int k = 0, prev = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = s_cum[q][L];
                    if (c <= j) { k = q + 1; prev = c; }
                }
                const int i = s_rb[k][L] + (j - prev);
                double f[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) f[c] = s_frame[c][L];
                if (inside(a.grid.pts + (size_t)i * 3, f)) s_hit[L] = 1;
            }
        }
        wave_lds_sync();
        hit = act && (hit || s_hit[lane] != 0);
        if (live && a.flags) a.flags[g] = hit ? 1 : 0;
        // one atomic per trajectory and wave: the lanes are in (trajectory, sample) order, so the lowest hit lane of a trajectory holds
        // its lowest hit sample of this wave
        const unsigned long long hits = __ballot(hit);
        const int first_lane = lane - s > 0 ? lane - s : 0;                       // where this trajectory's samples start in the wave
        const unsigned long long below = ((1ull << lane) - 1ull) & ~((1ull << first_lane) - 1ull);
        if (hit && (hits & below) == 0ull) atomicMin(&a.first_hit[b], s);
        wave_lds_sync();                                                           // the next trip overwrites the LDS records
    }
}

}  // namespace uavqp
