/*
 * poly_eval.c -- CPU restatement of the reference's trajectory evaluator.  TEST INFRASTRUCTURE ONLY.
 * Follows src/planner/traj_utils/include/traj_utils/poly_traj.hpp:
 *   segment search   :77-88 (same loop in evaluateVel :109-120, evaluateAcc :141-152)
 *   evaluatePos      :90-103   p = sum_i c_i t^i
 *   evaluateVel      :122-135  v = sum_i (i+1) c_{i+1} t^i
 *   evaluateAcc      :154-167  a = sum_i (i+2)(i+1) c_{i+2} t^i
 * with the power vector built by repeated multiplication (tv[i] = tv[i-1]*t) and a plain dot product,
 * as the reference does.  The reference has no recorded outputs for it; the restatement is pinned on the
 * reference's own header instead: poly_traj.hpp compiled from where it lies against the stand-in Eigen of
 * oracle/ref_shim/ (oracle/_ref, tests/test_oracle_vs_reference_source.py: 1e-13 at segment boundaries, the 1e-4
 * slack and past the end).
 */
#include <stddef.h>

/* coef: one trajectory in the C-ABI layout [axis][segment][nc]; out[3*K]: (pos, vel, acc selected by what) x xyz */
void oracle_poly_eval(int nc, int M, const double* times, const double* coef, double t, int what, double* out) {
    int idx = 0;
    while (idx < M && t > times[idx] + 1e-4) { t -= times[idx]; idx++; }   /* :77-81 (bounds check first: no OOB read) */
    if (idx == M) { idx--; t = times[idx]; }                                /* :83-87 */
    int k = 0;
    for (int d = 0; d < 3; ++d) {
        if (!((what >> d) & 1)) continue;
        for (int ax = 0; ax < 3; ++ax) {
            const double* c = coef + ((size_t)ax * M + idx) * nc;
            double tv = 1.0, acc = 0.0;
            for (int i = 0; i < nc - d; ++i) {
                double f = 1.0;
                for (int q = 0; q < d; ++q) f *= (double)(i + d - q);
                if (i > 0) tv *= t;
                acc += tv * (f * c[i + d]);
            }
            out[3 * k + ax] = acc;
        }
        ++k;
    }
}

/* PolyTraj::getTraj (:175-187) followed by getLength (:189-202) and getMeanVel (:204-207): positions sampled at t = 0, then
 * t += dt while t < total_time (the reference accumulates t in floating point with dt = 0.01 -- so does this loop: for a
 * total time that is a multiple of dt the accumulated t decides whether the last sample exists), length = sum of the chord
 * lengths, mean velocity = length / total_time.  out2[0] = length, out2[1] = mean velocity; returns the number of samples. */
#include <math.h>
int oracle_traj_length(int nc, int M, const double* times, const double* coef, double dt, double* out2) {
    double total = 0.0;
    for (int i = 0; i < M; ++i) total += times[i];                /* PolyTraj::init :59-72 */
    double t = 0.0, len = 0.0, pl[3] = {0, 0, 0}, pn[3];
    int n = 0;
    while (t < total) {
        oracle_poly_eval(nc, M, times, coef, t, 1, pn);
        if (n > 0) len += sqrt((pn[0] - pl[0]) * (pn[0] - pl[0]) + (pn[1] - pl[1]) * (pn[1] - pl[1]) + (pn[2] - pl[2]) * (pn[2] - pl[2]));
        pl[0] = pn[0]; pl[1] = pn[1]; pl[2] = pn[2];
        t += dt;
        ++n;
    }
    out2[0] = len;
    out2[1] = len / total;
    return n;
}
