/*
 * ellipsoid.c -- CPU restatement of the reference's SE(3) ellipsoid collision test.  TEST INFRASTRUCTURE ONLY.
 * Follows KinoAstar::isCollisionFree, src/planner/path_searching/src/kino_astar.cpp:721-758:
 *   :724  b3 = (acc + 9.81 z).normalized()
 *   :725-727  c1 = (cos 0, sin 0, 0);  b2 = b3.cross(c1).normalized();  b1 = b2.cross(b3).normalized()
 *   :729-737  Rot = [b1 b2 b3];  P = diag(robot_r, robot_r, robot_h);  E = Rot P Rot'
 *   :747-755  radius search robot_r + 0.1 around pt; collision if |E^-1 (o - pt)| <= 1 for a found point
 * The PCL kd-tree radius search is replaced by an exhaustive scan over the obstacle points (identical candidate
 * set up to float rounding at the radius boundary, which cannot matter: the ellipsoid lies inside radius robot_r).
 * Pinned on the reference's OWN code: oracle/_ref/libref_kino.so is KinoAstar::isCollisionFree + toPCL cut out of
 * kino_astar.cpp at build time and compiled inside the reference's own class declaration (oracle/Makefile,
 * ref_shim/ref_kino_capi.cpp; Eigen and PCL are stand-ins with Eigen's formulas for normalized() / 3 x 3 inverse() and a
 * float exhaustive radius search); tests/test_ellipsoid_oracle_vs_reference_source.py checks this restatement against it on
 * random clouds, on points straddling the float search sphere and on the ellipsoid surface.  The reference holds no recorded
 * outputs for this function.
 */
#include <math.h>

static void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static void normalize3(double* v) { double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= n; v[1] /= n; v[2] /= n; }

/* returns 1 if collision-free (the reference's return value), 0 if some obstacle point lies in the ellipsoid */
int oracle_is_collision_free(const double* pt, const double* acc, const double* obs, int n_obs, double robot_r, double robot_h) {
    double b3[3] = {acc[0], acc[1], acc[2] + 9.81}, c1[3] = {1.0, 0.0, 0.0}, b2[3], b1[3];
    normalize3(b3);
    cross3(b3, c1, b2); normalize3(b2);
    cross3(b2, b3, b1); normalize3(b1);
    /* E^-1 = Rot P^-1 Rot'  (Rot orthonormal) */
    const double radius = robot_r + 1e-1;
    for (int i = 0; i < n_obs; ++i) {
        const double d[3] = {obs[3 * i] - pt[0], obs[3 * i + 1] - pt[1], obs[3 * i + 2] - pt[2]};
        if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] > radius * radius) continue;
        const double u[3] = {(b1[0] * d[0] + b1[1] * d[1] + b1[2] * d[2]) / robot_r,
                             (b2[0] * d[0] + b2[1] * d[1] + b2[2] * d[2]) / robot_r,
                             (b3[0] * d[0] + b3[1] * d[1] + b3[2] * d[2]) / robot_h};
        /* tmp = E^-1 d = Rot u;  |tmp| = |u| */
        if (sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) <= 1.0) return 0;
    }
    return 1;
}

/* Corridor box of one waypoint from the obstacle cloud (config 5, "ellipsoid-derived corridor widths"): the
 * specification of include/uavqp.h uavqp_corridor_from_cloud_device restated with the reference's own
 * construction of E (kino_astar.cpp:724-737) and its collision metric |E^-1 (o - pt)| (:751-753), written with
 * the explicit body-frame projections rather than the device's quadratic form.
 *   g = min_o |E^-1 (o - pt)|;  h_i = min(h_max, max(0, g - 1) / (3 |E^-1 e_i|));  lo = pt - h, hi = pt + h.
 * Returns g (INFINITY for an empty cloud). */
double oracle_corridor_box(const double* pt, const double* acc, const double* obs, int n_obs, double robot_r, double robot_h,
                           double h_max, double* lo, double* hi) {
    double b3[3] = {acc[0], acc[1], acc[2] + 9.81}, c1[3] = {1.0, 0.0, 0.0}, b2[3], b1[3];
    normalize3(b3);
    cross3(b3, c1, b2); normalize3(b2);
    cross3(b2, b3, b1); normalize3(b1);
    double g = INFINITY;
    for (int i = 0; i < n_obs; ++i) {
        const double d[3] = {obs[3 * i] - pt[0], obs[3 * i + 1] - pt[1], obs[3 * i + 2] - pt[2]};
        const double u[3] = {(b1[0] * d[0] + b1[1] * d[1] + b1[2] * d[2]) / robot_r,
                             (b2[0] * d[0] + b2[1] * d[1] + b2[2] * d[2]) / robot_r,
                             (b3[0] * d[0] + b3[1] * d[1] + b3[2] * d[2]) / robot_h};
        const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (n < g) g = n;
    }
    const double margin = g > 1.0 ? g - 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
        /* |E^-1 e_i| = |P^-1 Rot' e_i| */
        const double u[3] = {b1[i] / robot_r, b2[i] / robot_r, b3[i] / robot_h};
        const double w = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        double h = margin / (3.0 * w);
        if (!(h < h_max)) h = h_max;
        lo[i] = pt[i] - h;
        hi[i] = pt[i] + h;
    }
    return g;
}
