// C wrapper around the reference's own PolyTraj (header-only: /root/reference/src/planner/traj_utils/include/traj_utils/
// poly_traj.hpp), compiled against the stand-in <Eigen/Eigen> of this directory.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
// The segment-search rule and the derivative factors are the reference's own code; only the dot product's summation
// order is the stand-in's (left to right).
#include <traj_utils/poly_traj.hpp>

extern "C" {

// coef: one trajectory in the C-ABI layout [axis][segment][nc]; out[9] = pos xyz, vel xyz, acc xyz at trajectory time t
// (PolyTraj::evaluatePos / evaluateVel / evaluateAcc, poly_traj.hpp:74-168).
void ref_polytraj_eval(int nc, int M, const double* times, const double* coef, double t, double* out) {
    PolyTraj traj;
    traj.reset();
    for (int i = 0; i < M; ++i) {
        std::vector<double> cx(coef + ((size_t)0 * M + i) * nc, coef + ((size_t)0 * M + i + 1) * nc);
        std::vector<double> cy(coef + ((size_t)1 * M + i) * nc, coef + ((size_t)1 * M + i + 1) * nc);
        std::vector<double> cz(coef + ((size_t)2 * M + i) * nc, coef + ((size_t)2 * M + i + 1) * nc);
        traj.addSegment(cx, cy, cz, times[i]);
    }
    traj.init();
    const Eigen::Vector3d p = traj.evaluatePos(t), v = traj.evaluateVel(t), a = traj.evaluateAcc(t);
    for (int k = 0; k < 3; ++k) { out[k] = p[k]; out[3 + k] = v[k]; out[6 + k] = a[k]; }
}

// PolyTraj::getTraj + getLength + getMeanVel (poly_traj.hpp:175-207: samples every 0.01 s, accumulated t); returns the sample count
int ref_polytraj_length(int nc, int M, const double* times, const double* coef, double* out2) {
    PolyTraj traj;
    traj.reset();
    for (int i = 0; i < M; ++i) {
        std::vector<double> cx(coef + ((size_t)0 * M + i) * nc, coef + ((size_t)0 * M + i + 1) * nc);
        std::vector<double> cy(coef + ((size_t)1 * M + i) * nc, coef + ((size_t)1 * M + i + 1) * nc);
        std::vector<double> cz(coef + ((size_t)2 * M + i) * nc, coef + ((size_t)2 * M + i + 1) * nc);
        traj.addSegment(cx, cy, cz, times[i]);
    }
    traj.init();
    const int n = (int)traj.getTraj().size();
    out2[0] = traj.getLength();
    out2[1] = traj.getMeanVel();
    return n;
}

double ref_polytraj_total_time(int M, const double* times) {
    PolyTraj traj;
    traj.reset();
    std::vector<double> z(2, 0.0);
    for (int i = 0; i < M; ++i) traj.addSegment(z, z, z, times[i]);
    traj.init();
    return traj.getTotalTIme();
}

}  // extern "C"
