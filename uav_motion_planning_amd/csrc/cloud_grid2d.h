// cloud_grid2d.h -- corridor boxes from an obstacle cloud with the candidate points of a row taken from a 2-D cell grid, nearest cells
// first (round 5; the third variant of uavqp_corridor_from_cloud_device after cloud_corridor_kernel and cloud_window_kernel).
//
// What a box needs of the cloud is g = min_o |E^-1 (o - w)| -- and only while g < g_cap = 1 + 3 h_max max_i |E^-1 e_i|: beyond that every
// half-width is h_max whatever g is (obstacle_grid.h).  A point at Euclidean distance d has metric >= d / max(r, h), so once the points
// within D of a row have been scanned and D >= min(g_cap, g so far) max(r, h), no other point can change its box.  cloud_window_kernel
// uses the worst case of that (D = reach = 10 m on config 5) along ONE axis: 44 % of all (row, point) pairs.  Here:
//   * points and rows are counting-sorted by the cell of a 2-D grid over the two longest axes of the cloud's bounding box (row-major), so
//     that the 256 rows of a block are neighbours in the plane (about one cell) and the points of a run of cells along the first axis
//     are contiguous;
//   * a block scans the cells within a small radius of its rows' bounding box first, every lane against every point through LDS
//     broadcasts as before (nothing diverges); then each row asks whether min(g_cap, g) max(r, h) is covered by that radius -- in a
//     pillar forest nearly always: the nearest obstacle is a few metres away -- and only if some row of the block is not, the next ring
//     (2 -> 4 -> reach, in cells) is scanned, cells already seen excluded.
// The minimum is taken over a superset of the points that can matter with the same arithmetic per pair: boxes bit-identical to the
// exhaustive scan (tests/test_gpu_cloud_corridor.py).  Not used when the caller wants the clearance itself (an exact min over the cloud).
#pragma once
#include "obstacle_grid.h"

namespace uavqp {

constexpr int CLOUD2D_MAX_CELLS = 4096;
struct Cloud2D {
    int axA, axB, nx, ny;          // cell (ca, cb) has index cb * nx + ca; axA / axB: the two longest axes of the cloud's bounding box
    double loA, loB, inv_cell, cell;
    double bb_lo[3], bb_hi[3];     // bounding box of the finite points (the exact cull of CorridorRow::culled_by_box)
};
__device__ __forceinline__ int cloud2d_coord(double v, double lo, double inv, int n) {
    const double t = (v - lo) * inv;
    return t >= 0.0 ? (t < (double)n ? (int)t : n - 1) : 0;   // NaN -> 0; everything outside is clamped to the border cells
}
__device__ __forceinline__ int cloud2d_cell(const Cloud2D& g, double a, double b) {
    return cloud2d_coord(b, g.loB, g.inv_cell, g.ny) * g.nx + cloud2d_coord(a, g.loA, g.inv_cell, g.nx);
}

// one block: bounding box -> axes, cell size (at least min_cell, at most CLOUD2D_MAX_CELLS cells); zeroes both histograms
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void cloud2d_setup_kernel(const double* __restrict__ obs, int n_obs, double min_cell, Cloud2D* __restrict__ cg,
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
    for (int i = threadIdx.x; i <= CLOUD2D_MAX_CELLS + 1; i += 1024) { if (i <= CLOUD2D_MAX_CELLS) pt_hist[i] = 0; row_hist[i] = 0; }
    if (threadIdx.x == 0) {
        double ext[3];
        for (int ax = 0; ax < 3; ++ax) {
            const double e = s[0][3 + ax] - s[0][ax];
            ext[ax] = (e >= 0.0 && e < INFINITY) ? e : 0.0;      // (no finite coordinate on this axis: extent 0)
        }
        int axA = 0;
        for (int ax = 1; ax < 3; ++ax) if (ext[ax] > ext[axA]) axA = ax;
        int axB = axA == 0 ? 1 : 0;
        for (int ax = 0; ax < 3; ++ax) if (ax != axA && ext[ax] > ext[axB]) axB = ax;
        const double eA = ext[axA] > 0.0 ? ext[axA] : 1.0, eB = ext[axB] > 0.0 ? ext[axB] : 1.0;
        double cell = sqrt(eA * eB / (0.5 * CLOUD2D_MAX_CELLS));
        cell = cell > min_cell ? cell : min_cell;
        int nx = (int)ceil(eA / cell), ny = (int)ceil(eB / cell);
        nx = nx < 1 ? 1 : nx; ny = ny < 1 ? 1 : ny;
        while ((long long)nx * ny > CLOUD2D_MAX_CELLS) { cell *= 1.25; nx = (int)ceil(eA / cell); ny = (int)ceil(eB / cell); nx = nx < 1 ? 1 : nx; ny = ny < 1 ? 1 : ny; }
        cg->axA = axA; cg->axB = axB; cg->nx = nx; cg->ny = ny;
        cg->loA = fabs(s[0][axA]) < INFINITY ? s[0][axA] : 0.0;
        cg->loB = fabs(s[0][axB]) < INFINITY ? s[0][axB] : 0.0;
        cg->cell = cell;
        cg->inv_cell = 1.0 / cell;
        for (int ax = 0; ax < 3; ++ax) { cg->bb_lo[ax] = s[0][ax]; cg->bb_hi[ax] = s[0][3 + ax]; }
    }
}
#endif

__device__ __forceinline__ void cloud2d_slice(long long total, long long& i0, long long& i1) {
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    i0 = (long long)blockIdx.x * per;
    i1 = i0 + per < total ? i0 + per : total;
}
// histograms of the points and of the rows over the cells, hist[c + 1] counts cell c (LDS counts first, one global add per block and cell)
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void cloud2d_hist_kernel(const double* __restrict__ obs, int n_obs, const double* __restrict__ wp, int n_rows,
                                                           const Cloud2D* __restrict__ cgp, int32_t* __restrict__ pt_hist, int32_t* __restrict__ row_hist) {
    __shared__ int s_h[2 * CLOUD2D_MAX_CELLS];
    const Cloud2D cg = *cgp;
    const int nc = cg.nx * cg.ny;
    for (int i = threadIdx.x; i < 2 * nc; i += 256) s_h[i] = 0;
    __syncthreads();
    long long i0, i1;
    cloud2d_slice((long long)n_obs + n_rows, i0, i1);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (i < n_obs) atomicAdd(&s_h[cloud2d_cell(cg, obs[(size_t)i * 3 + cg.axA], obs[(size_t)i * 3 + cg.axB])], 1);
        else atomicAdd(&s_h[nc + cloud2d_cell(cg, wp[(size_t)(i - n_obs) * 3 + cg.axA], wp[(size_t)(i - n_obs) * 3 + cg.axB])], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nc; i += 256) {
        const int c = s_h[i];
        if (c) atomicAdd(i < nc ? &pt_hist[1 + i] : &row_hist[1 + i - nc], c);
    }
}
#endif
// inclusive scans in place (start[c] = first element of cell c, start[n cells] = total) and cursor copies; one block
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void cloud2d_scan_kernel(const Cloud2D* __restrict__ cgp, int32_t* __restrict__ pt_start, int32_t* __restrict__ pt_cursor,
                                                            int32_t* __restrict__ row_start, int32_t* __restrict__ row_cursor) {
    __shared__ int s_tot[1024];
    const int nb = cgp->nx * cgp->ny;
    auto scan = [&](int32_t* a, int32_t* cur) {
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
    scan(pt_start, pt_cursor);
    scan(row_start, row_cursor);
}
#endif
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void cloud2d_scatter_kernel(const double* __restrict__ obs, int n_obs, const double* __restrict__ wp, int n_rows,
                                                              const Cloud2D* __restrict__ cgp, int32_t* __restrict__ pt_cursor, int32_t* __restrict__ row_cursor,
                                                              double* __restrict__ pts_sorted, int32_t* __restrict__ row_perm) {
    __shared__ int s_h[2 * CLOUD2D_MAX_CELLS];
    const Cloud2D cg = *cgp;
    const int nc = cg.nx * cg.ny;
    for (int i = threadIdx.x; i < 2 * nc; i += 256) s_h[i] = 0;
    __syncthreads();
    long long i0, i1;
    cloud2d_slice((long long)n_obs + n_rows, i0, i1);
    auto bin_of = [&](long long i) -> int {
        return i < n_obs ? cloud2d_cell(cg, obs[(size_t)i * 3 + cg.axA], obs[(size_t)i * 3 + cg.axB])
                         : nc + cloud2d_cell(cg, wp[(size_t)(i - n_obs) * 3 + cg.axA], wp[(size_t)(i - n_obs) * 3 + cg.axB]);
    };
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&s_h[bin_of(i)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nc; i += 256) {
        const int c = s_h[i];
        s_h[i] = c ? atomicAdd(i < nc ? &pt_cursor[i] : &row_cursor[i - nc], c) : 0;
    }
    __syncthreads();
    // (the order inside a cell is arbitrary: it decides which lane scans a row / where in a tile a point sits, never a result)
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

struct Cloud2DArgs {
    CloudCorridorArgs c;          // row_perm, pt_start, pts_sorted, reach as in the window variant (pt_start per CELL here)
    const Cloud2D* grid;
    unsigned int* phase_count;    // [4] blocks that needed phase k (statistics; may be null)
};

template <int R>
__global__ __launch_bounds__(256) void cloud_grid2d_kernel(Cloud2DArgs aa) {
    const CloudCorridorArgs& a = aa.c;
    constexpr int TILE = 1024, MAXR = 192, NPH = 3;
    __shared__ double s_obs[TILE * 3];
    __shared__ double s_box[4][4];      // per wave: min A, max A, min B, max B (cell units)
    __shared__ int s_rs[MAXR], s_pf[MAXR + 1];
    __shared__ int s_more;
    const Cloud2D cg = *aa.grid;
    const int wave = threadIdx.x >> 6;
    const double rmax = a.robot_r > a.robot_h ? a.robot_r : a.robot_h;
    const double reach_c = a.reach * cg.inv_cell;                  // worst-case radius in cells
    const long long n_round = ((long long)a.n_rows + 255) / 256 * 256;
    for (long long g0 = (long long)blockIdx.x * 256; g0 < n_round; g0 += (long long)gridDim.x * 256) {
        const long long g = g0 + threadIdx.x;
        const bool live = g < a.n_rows;
        const int rid = live ? a.row_perm[g] : 0;
        CorridorRow<R> row;
        double uA = 0.0, uB = 0.0, cap2 = 0.0;
        if (live) {
            row.setup(a, rid);
            uA = (row.p[cg.axA] - cg.loA) * cg.inv_cell;           // position in cell units (may lie outside the grid: the windows are clamped)
            uB = (row.p[cg.axB] - cg.loB) * cg.inv_cell;
            const double qm = fmax(row.qxx, fmax(row.qyy, row.qzz));
            const double gcap = 1.0 + 3.0 * a.h_max * sqrt(qm);   // beyond this clearance every half-width of THIS row is h_max
            cap2 = gcap * gcap * (1.0 + 1e-9);
        }
        // bounding box of the block's rows in cell units
        {
            const bool fin = live && fabs(uA) < 1e300 && fabs(uB) < 1e300;      // (a non-finite waypoint: its row takes part in no window)
            double a0 = fin ? uA : INFINITY, a1 = fin ? uA : -INFINITY, b0 = fin ? uB : INFINITY, b1 = fin ? uB : -INFINITY;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                a0 = fmin(a0, __shfl_xor(a0, off, 64)); a1 = fmax(a1, __shfl_xor(a1, off, 64));
                b0 = fmin(b0, __shfl_xor(b0, off, 64)); b1 = fmax(b1, __shfl_xor(b1, off, 64));
            }
            __syncthreads();      // (the previous round's ranges and boxes have been read by everybody)
            if ((threadIdx.x & 63) == 0) { s_box[wave][0] = a0; s_box[wave][1] = a1; s_box[wave][2] = b0; s_box[wave][3] = b1; }
            __syncthreads();
        }
        const double bA0 = fmin(fmin(s_box[0][0], s_box[1][0]), fmin(s_box[2][0], s_box[3][0]));
        const double bA1 = fmax(fmax(s_box[0][1], s_box[1][1]), fmax(s_box[2][1], s_box[3][1]));
        const double bB0 = fmin(fmin(s_box[0][2], s_box[1][2]), fmin(s_box[2][2], s_box[3][2]));
        const double bB1 = fmax(fmax(s_box[0][3], s_box[1][3]), fmax(s_box[2][3], s_box[3][3]));
        double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        int pA0 = 1, pA1 = 0, pB0 = 1, pB1 = 0;       // cells scanned so far (empty)
        if (bA0 <= bA1) {
            for (int ph = 0; ph < NPH; ++ph) {
                double rad = ph == 0 ? 2.0 : (ph == 1 ? 4.0 : reach_c);
                rad = rad < reach_c ? rad : reach_c;
                if (ph > 0 && rad <= (ph == 1 ? 2.0 : 4.0)) break;                   // (the previous ring already reached that far)
                // cells that overlap [box - rad, box + rad], clamped (everything outside the grid was binned into the border cells)
                auto cl = [](double v, int n) -> int { return v >= 0.0 ? (v < (double)n ? (int)v : n - 1) : 0; };
                const int wA0 = cl(floor(bA0 - rad), cg.nx), wA1 = cl(floor(bA1 + rad), cg.nx);
                const int wB0 = cl(floor(bB0 - rad), cg.ny), wB1 = cl(floor(bB1 + rad), cg.ny);
                // runs of cells along A: two slots per cell row (the cells of the previous ring are cut out of the middle), MAXR / 2 cell rows
                // per pass; every slot is looked up by its own thread, one thread makes the prefix sums (empty runs stay in the list)
                for (int cb0 = wB0; cb0 <= wB1; cb0 += MAXR / 2) {
                    const int nrow = min(MAXR / 2, wB1 - cb0 + 1), nr = 2 * nrow;
                    __syncthreads();      // (the previous pass / phase is done with the run list)
                    if ((int)threadIdx.x < nr) {
                        const int cb = cb0 + ((int)threadIdx.x >> 1), right = threadIdx.x & 1;
                        const bool cut = cb >= pB0 && cb <= pB1;
                        const int ca0 = right ? (cut ? pA1 + 1 : 1) : wA0, ca1 = right ? (cut ? wA1 : 0) : (cut ? pA0 - 1 : wA1);
                        int st = 0, en = 0;
                        if (ca0 <= ca1) { st = a.pt_start[cb * cg.nx + ca0]; en = a.pt_start[cb * cg.nx + ca1 + 1]; }
                        s_rs[threadIdx.x] = st;
                        s_pf[1 + threadIdx.x] = en - st;
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        int tot = 0;
                        s_pf[0] = 0;
                        for (int i = 1; i <= nr; ++i) { tot += s_pf[i]; s_pf[i] = tot; }
                        if (aa.phase_count && cb0 == wB0) atomicAdd(&aa.phase_count[ph], 1u);
                    }
                    __syncthreads();
                    const int total = s_pf[nr];
                    for (int v0 = 0; v0 < total; v0 += TILE) {
                        const int nt = min(TILE, total - v0);
                        __syncthreads();
                        for (int i = threadIdx.x; i < nt; i += 256) {
                            const int v = v0 + i;
                            int lo_r = 0, hi_r = nr - 1;       // the run that holds element v of the concatenation (the last one that starts at or before it)
                            while (lo_r < hi_r) {
                                const int mid = (lo_r + hi_r + 1) >> 1;
                                if (s_pf[mid] <= v) lo_r = mid; else hi_r = mid - 1;
                            }
                            const double* src = a.pts_sorted + (size_t)(s_rs[lo_r] + (v - s_pf[lo_r])) * 3;
                            s_obs[3 * i] = src[0]; s_obs[3 * i + 1] = src[1]; s_obs[3 * i + 2] = src[2];
                        }
                        __syncthreads();
                        if (live) {
                            int i = 0;
                            for (; i + 3 < nt; i += 4) {
                                const double* o = s_obs + 3 * i;
                                m0 = min_nn(m0, row.metric2(o[0], o[1], o[2]));
                                m1 = min_nn(m1, row.metric2(o[3], o[4], o[5]));
                                m2 = min_nn(m2, row.metric2(o[6], o[7], o[8]));
                                m3 = min_nn(m3, row.metric2(o[9], o[10], o[11]));
                            }
                            for (; i < nt; ++i) m0 = min_nn(m0, row.metric2(s_obs[3 * i], s_obs[3 * i + 1], s_obs[3 * i + 2]));
                        }
                    }
                }
                pA0 = wA0; pA1 = wA1; pB0 = wB0; pB1 = wB1;
                if (ph == NPH - 1 || rad >= reach_c) break;
                // every point within rad cells of a row of the block has been seen.  Does some row need more?  A point at distance d has
                // metric >= d / max(r, h): the box of a row is final once  min(g_cap, g) max(r, h) <= rad cell.
                const double g2 = fmin(fmin(m0, m1), fmin(m2, m3));
                const double need = sqrt(fmin(g2, cap2)) * rmax * (1.0 + 1e-9);
                const bool more = live && !(need <= rad * cg.cell * (1.0 - 1e-9));      // (NaN: more)
                __syncthreads();
                if (threadIdx.x == 0) s_more = 0;
                __syncthreads();
                if (more) s_more = 1;
                __syncthreads();
                if (!s_more) break;
            }
        }
        if (live) row.emit(a, rid, fmin(fmin(m0, m1), fmin(m2, m3)));
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Two passes (cloud_window = 3, the default since round 5).  The ring kernel above is exact but a block goes on until its LAST row is
// satisfied, and the radius a row needs depends on its height and attitude, not only on where it is in the plane: on config 5 every
// block ended in the outermost ring.  What predicts the radius a row needs is the clearance it has found so far -- so:
//   pass 1: every row the bounding box does not cull (obstacle_grid.h: culled_by_box) scans the ring of 2 cells around its block; a row
//           whose clearance g so far satisfies  min(g_cap, g) max(r, h) <= 2 cells  is final (42 % of them on config 5).  The others record
//           their minimum and the radius it implies -- an UPPER bound of what they need: the minimum can only fall -- as a sort key
//           (radius bucket, cell);
//   pass 2: those rows, counting-sorted by that key: a block holds rows that need about the same radius around about the same place and
//           scans exactly that window, once.
// Same arithmetic per (row, point) pair, minimum over a superset of the points that can matter: boxes bit-identical to the exhaustive scan.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int CLOUD2D_BUCKETS = 7;     // radius buckets of pass 2: (2 + b, 3 + b] cells, the last one open-ended

// histograms with the bounding-box cull: a culled row goes to the bin behind the last cell (rows only; the points as cloud2d_hist_kernel)
template <int R>
__global__ __launch_bounds__(256) void cloud2d_hist_cull_kernel(CloudCorridorArgs a, const Cloud2D* __restrict__ cgp,
                                                                int32_t* __restrict__ pt_hist, int32_t* __restrict__ row_hist, int32_t* __restrict__ row_bin) {
    __shared__ int s_h[2 * CLOUD2D_MAX_CELLS + 1];
    const Cloud2D cg = *cgp;
    const int nc = cg.nx * cg.ny;
    for (int i = threadIdx.x; i < 2 * nc + 1; i += 256) s_h[i] = 0;
    __syncthreads();
    double bb_lo[3], bb_hi[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { bb_lo[ax] = cg.bb_lo[ax]; bb_hi[ax] = cg.bb_hi[ax]; }
    long long i0, i1;
    cloud2d_slice((long long)a.n_obs + a.n_rows, i0, i1);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (i < a.n_obs) atomicAdd(&s_h[cloud2d_cell(cg, a.obs[(size_t)i * 3 + cg.axA], a.obs[(size_t)i * 3 + cg.axB])], 1);
        else {
            CorridorRow<R> row;
            row.setup(a, i - a.n_obs);
            const int b = row.culled_by_box(a, bb_lo, bb_hi) ? nc : cloud2d_cell(cg, row.p[cg.axA], row.p[cg.axB]);
            row_bin[i - a.n_obs] = b;
            atomicAdd(&s_h[nc + b], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nc + 1; i += 256) {
        const int c = s_h[i];
        if (c) atomicAdd(i < nc ? &pt_hist[1 + i] : &row_hist[1 + i - nc], c);
    }
}
// scans over nb_pt + 1 / nb_row + 1 entries (bin counts at [1..]); one block
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(1024) void cloud2d_scan_n_kernel(const Cloud2D* __restrict__ cgp, int extra_row_bins, int row_mult, int32_t* __restrict__ pt_start,
                                                              int32_t* __restrict__ pt_cursor, int32_t* __restrict__ row_start, int32_t* __restrict__ row_cursor) {
    __shared__ int s_tot[1024];
    const int nc = cgp->nx * cgp->ny;
    auto scan = [&](int32_t* a, int32_t* cur, int nb) {
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
    if (pt_start) scan(pt_start, pt_cursor, nc);
    scan(row_start, row_cursor, nc * row_mult + extra_row_bins);
}
#endif
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void cloud2d_scatter_bin_kernel(const double* __restrict__ obs, int n_obs, const int32_t* __restrict__ row_bin, int n_rows,
                                                                  const Cloud2D* __restrict__ cgp, int32_t* __restrict__ pt_cursor, int32_t* __restrict__ row_cursor,
                                                                  double* __restrict__ pts_sorted, int32_t* __restrict__ row_perm) {
    __shared__ int s_h[2 * CLOUD2D_MAX_CELLS + 1];
    const Cloud2D cg = *cgp;
    const int nc = cg.nx * cg.ny;
    for (int i = threadIdx.x; i < 2 * nc + 1; i += 256) s_h[i] = 0;
    __syncthreads();
    long long i0, i1;
    cloud2d_slice((long long)n_obs + n_rows, i0, i1);
    auto bin_of = [&](long long i) -> int {
        return i < n_obs ? cloud2d_cell(cg, obs[(size_t)i * 3 + cg.axA], obs[(size_t)i * 3 + cg.axB]) : nc + row_bin[i - n_obs];
    };
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&s_h[bin_of(i)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nc + 1; i += 256) {
        const int c = s_h[i];
        s_h[i] = c ? atomicAdd(i < nc ? &pt_cursor[i] : &row_cursor[i - nc], c) : 0;
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
// rows of pass 2 by their key (-1: not in pass 2): one global cursor add per row (about a third of the rows, a few thousand bins)
#ifndef UAVQP_KERNEL_TU   // a plain (non-template) kernel: emitted once, by the host translation unit (uavqp.hip)
__global__ __launch_bounds__(256) void cloud2d_scatter2_kernel(const int32_t* __restrict__ row_key, int n_rows, int32_t* __restrict__ cursor2, int32_t* __restrict__ perm2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (long long)gridDim.x * 256) {
        const int k = row_key[i];
        if (k >= 0) perm2[atomicAdd(&cursor2[k], 1)] = (int)i;
    }
}
#endif

struct Cloud2PArgs {
    CloudCorridorArgs c;          // row_perm (pass 1: by cell, culled rows last), pt_start (per cell), pts_sorted, reach
    const Cloud2D* grid;
    const int32_t* row_start;     // [cells + 2]: rows [row_start[cells], n_rows) are culled
    int32_t* row_key;             // [n_rows] out of pass 1: sort key of pass 2, -1 = final
    double* row_m;                // [n_rows] out of pass 1: minimum of the squared metric so far
    int32_t* hist2;               // [buckets * cells + 1] counts at [1 + key] (pass 1 adds), starts after the scan (pass 2 reads)
    const int32_t* perm2;         // rows of pass 2 in key order
};

// the points of the cells [wA0, wA1] x [wB0, wB1] against the block's rows (every lane, every point, through LDS tiles)
template <int R>
__device__ __forceinline__ void cloud2d_scan_window(const CloudCorridorArgs& a, const Cloud2D& cg, int wA0, int wA1, int wB0, int wB1, const CorridorRow<R>& row, bool live,
                                                    double* s_obs, int* s_rs, int* s_pf, double& m0, double& m1, double& m2, double& m3) {
    constexpr int TILE = 1024, MAXR = 192;
    for (int cb0 = wB0; cb0 <= wB1; cb0 += MAXR) {
        const int nr = min(MAXR, wB1 - cb0 + 1);
        __syncthreads();
        if ((int)threadIdx.x < nr) {
            const int cb = cb0 + (int)threadIdx.x;
            const int st = a.pt_start[cb * cg.nx + wA0], en = a.pt_start[cb * cg.nx + wA1 + 1];
            s_rs[threadIdx.x] = st;
            s_pf[1 + threadIdx.x] = en - st;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            s_pf[0] = 0;
            for (int i = 1; i <= nr; ++i) { tot += s_pf[i]; s_pf[i] = tot; }
        }
        __syncthreads();
        const int total = s_pf[nr];
        for (int v0 = 0; v0 < total; v0 += TILE) {
            const int nt = min(TILE, total - v0);
            __syncthreads();
            for (int i = threadIdx.x; i < nt; i += 256) {
                const int v = v0 + i;
                int lo_r = 0, hi_r = nr - 1;
                while (lo_r < hi_r) {
                    const int mid = (lo_r + hi_r + 1) >> 1;
                    if (s_pf[mid] <= v) lo_r = mid; else hi_r = mid - 1;
                }
                const double* src = a.pts_sorted + (size_t)(s_rs[lo_r] + (v - s_pf[lo_r])) * 3;
                s_obs[3 * i] = src[0]; s_obs[3 * i + 1] = src[1]; s_obs[3 * i + 2] = src[2];
            }
            __syncthreads();
            if (live) {
                int i = 0;
                for (; i + 3 < nt; i += 4) {
                    const double* o = s_obs + 3 * i;
                    m0 = min_nn(m0, row.metric2(o[0], o[1], o[2]));
                    m1 = min_nn(m1, row.metric2(o[3], o[4], o[5]));
                    m2 = min_nn(m2, row.metric2(o[6], o[7], o[8]));
                    m3 = min_nn(m3, row.metric2(o[9], o[10], o[11]));
                }
                for (; i < nt; ++i) m0 = min_nn(m0, row.metric2(s_obs[3 * i], s_obs[3 * i + 1], s_obs[3 * i + 2]));
            }
        }
    }
}

// block-wide bounding box (cell units) of the live rows' positions: (a0, a1, b0, b1), empty if a0 > a1
__device__ __forceinline__ void cloud2d_block_box(bool fin, double uA, double uB, double (*s_box)[4], double& bA0, double& bA1, double& bB0, double& bB1) {
    double a0 = fin ? uA : INFINITY, a1 = fin ? uA : -INFINITY, b0 = fin ? uB : INFINITY, b1 = fin ? uB : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a0 = fmin(a0, __shfl_xor(a0, off, 64)); a1 = fmax(a1, __shfl_xor(a1, off, 64));
        b0 = fmin(b0, __shfl_xor(b0, off, 64)); b1 = fmax(b1, __shfl_xor(b1, off, 64));
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_box[wave][0] = a0; s_box[wave][1] = a1; s_box[wave][2] = b0; s_box[wave][3] = b1; }
    __syncthreads();
    bA0 = fmin(fmin(s_box[0][0], s_box[1][0]), fmin(s_box[2][0], s_box[3][0]));
    bA1 = fmax(fmax(s_box[0][1], s_box[1][1]), fmax(s_box[2][1], s_box[3][1]));
    bB0 = fmin(fmin(s_box[0][2], s_box[1][2]), fmin(s_box[2][2], s_box[3][2]));
    bB1 = fmax(fmax(s_box[0][3], s_box[1][3]), fmax(s_box[2][3], s_box[3][3]));
}

template <int R>
__global__ __launch_bounds__(256) void cloud2d_pass1_kernel(Cloud2PArgs aa) {
    const CloudCorridorArgs& a = aa.c;
    constexpr int TILE = 1024, MAXR = 192;
    __shared__ double s_obs[TILE * 3];
    __shared__ double s_box[4][4];
    __shared__ int s_rs[MAXR], s_pf[MAXR + 1];
    const Cloud2D cg = *aa.grid;
    const int nc = cg.nx * cg.ny;
    const int n_scan = aa.row_start[nc];                      // rows [n_scan, n_rows): culled by the bounding box -- emitted without a scan
    const double rmax = a.robot_r > a.robot_h ? a.robot_r : a.robot_h;
    const double reach_c = a.reach * cg.inv_cell;
    const double rad0 = reach_c < 2.0 ? reach_c : 2.0;        // the ring of pass 1, in cells
    auto cl = [](double v, int n) -> int { return v >= 0.0 ? (v < (double)n ? (int)v : n - 1) : 0; };
    const long long n_round = ((long long)a.n_rows + 255) / 256 * 256;
    for (long long g0 = (long long)blockIdx.x * 256; g0 < n_round; g0 += (long long)gridDim.x * 256) {
        const long long g = g0 + threadIdx.x;
        const bool present = g < a.n_rows;
        const bool live = present && g < n_scan;
        const int rid = present ? a.row_perm[g] : 0;
        CorridorRow<R> row;
        if (present) row.setup(a, rid);
        if (g0 >= n_scan) {                                    // (block-uniform)
            if (present) { row.emit(a, rid, INFINITY); aa.row_key[rid] = -1; }
            continue;
        }
        double uA = 0.0, uB = 0.0, cap2 = 0.0;
        if (live) {
            uA = (row.p[cg.axA] - cg.loA) * cg.inv_cell;
            uB = (row.p[cg.axB] - cg.loB) * cg.inv_cell;
            const double gcap = 1.0 + 3.0 * a.h_max * sqrt(fmax(row.qxx, fmax(row.qyy, row.qzz)));
            cap2 = gcap * gcap * (1.0 + 1e-9);
        }
        const bool fin = live && fabs(uA) < 1e300 && fabs(uB) < 1e300;
        double bA0, bA1, bB0, bB1;
        cloud2d_block_box(fin, uA, uB, s_box, bA0, bA1, bB0, bB1);
        double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        if (bA0 <= bA1)
            cloud2d_scan_window<R>(a, cg, cl(floor(bA0 - rad0), cg.nx), cl(floor(bA1 + rad0), cg.nx), cl(floor(bB0 - rad0), cg.ny), cl(floor(bB1 + rad0), cg.ny),
                                   row, live, s_obs, s_rs, s_pf, m0, m1, m2, m3);
        const double g2 = fmin(fmin(m0, m1), fmin(m2, m3));
        if (present) {
            // every point within rad0 cells of this row has been seen; a point at distance d has metric >= d / max(r, h)
            const double need = sqrt(fmin(g2, cap2)) * rmax * (1.0 + 1e-9);                    // metres (an upper bound of what the row needs)
            // (a row with a non-finite coordinate sees NaN / inf metrics only -- the minimum stays +inf whatever is scanned, as in the exhaustive scan)
            const bool done = !live || !fin || rad0 >= reach_c || need <= rad0 * cg.cell * (1.0 - 1e-9);
            if (done) {
                row.emit(a, rid, live ? g2 : INFINITY);
                aa.row_key[rid] = -1;
            } else {
                int bk = (int)floor(need * cg.inv_cell - rad0);                                // radius bucket: need in (rad0 + bk, rad0 + bk + 1] cells
                bk = (bk >= 0 && bk < CLOUD2D_BUCKETS) ? bk : (bk < 0 ? 0 : CLOUD2D_BUCKETS - 1);      // (NaN need: the last bucket -- the whole reach)
                const int key = bk * nc + cloud2d_cell(cg, row.p[cg.axA], row.p[cg.axB]);
                aa.row_key[rid] = key;
                aa.row_m[rid] = g2;
                atomicAdd(&aa.hist2[1 + key], 1);
            }
        }
    }
}

template <int R>
__global__ __launch_bounds__(256) void cloud2d_pass2_kernel(Cloud2PArgs aa) {
    const CloudCorridorArgs& a = aa.c;
    constexpr int TILE = 1024, MAXR = 192;
    __shared__ double s_obs[TILE * 3];
    __shared__ double s_box[4][4];
    __shared__ double s_rad[4];
    __shared__ int s_rs[MAXR], s_pf[MAXR + 1];
    const Cloud2D cg = *aa.grid;
    const int nc = cg.nx * cg.ny;
    const int n2 = aa.hist2[CLOUD2D_BUCKETS * nc];             // (after the scan: the number of rows of pass 2)
    const double rmax = a.robot_r > a.robot_h ? a.robot_r : a.robot_h;
    const double reach_c = a.reach * cg.inv_cell;
    auto cl = [](double v, int n) -> int { return v >= 0.0 ? (v < (double)n ? (int)v : n - 1) : 0; };
    const long long n_round = ((long long)n2 + 255) / 256 * 256;
    for (long long g0 = (long long)blockIdx.x * 256; g0 < n_round; g0 += (long long)gridDim.x * 256) {
        const long long g = g0 + threadIdx.x;
        const bool live = g < n2;
        const int rid = live ? aa.perm2[g] : 0;
        CorridorRow<R> row;
        double uA = 0.0, uB = 0.0, need_c = 0.0, mprev = INFINITY;
        if (live) {
            row.setup(a, rid);
            uA = (row.p[cg.axA] - cg.loA) * cg.inv_cell;
            uB = (row.p[cg.axB] - cg.loB) * cg.inv_cell;
            const double gcap = 1.0 + 3.0 * a.h_max * sqrt(fmax(row.qxx, fmax(row.qyy, row.qzz)));
            mprev = aa.row_m[rid];
            need_c = sqrt(fmin(mprev, gcap * gcap * (1.0 + 1e-9))) * rmax * (1.0 + 1e-9) * cg.inv_cell * (1.0 + 1e-9);      // cells
        }
        const bool fin = live;                                   // (pass 1 kept the rows with non-finite coordinates)
        if (!(need_c <= reach_c)) need_c = reach_c;              // (NaN: the whole reach)
        double bA0, bA1, bB0, bB1;
        cloud2d_block_box(fin, uA, uB, s_box, bA0, bA1, bB0, bB1);
        // the block's radius: the largest one a row of it needs (rows are sorted by it: about a cell apart)
        double rad = live ? need_c : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) rad = fmax(rad, __shfl_xor(rad, off, 64));
        if ((threadIdx.x & 63) == 0) s_rad[threadIdx.x >> 6] = rad;
        __syncthreads();
        rad = fmax(fmax(s_rad[0], s_rad[1]), fmax(s_rad[2], s_rad[3]));
        double m0 = mprev, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        if (bA0 <= bA1)
            cloud2d_scan_window<R>(a, cg, cl(floor(bA0 - rad), cg.nx), cl(floor(bA1 + rad), cg.nx), cl(floor(bB0 - rad), cg.ny), cl(floor(bB1 + rad), cg.ny),
                                   row, live, s_obs, s_rs, s_pf, m0, m1, m2, m3);
        if (live) row.emit(a, rid, fmin(fmin(m0, m1), fmin(m2, m3)));
    }
}

}  // namespace uavqp
