// C wrapper around the reference's own traj_optimization::MinimumControl, compiled from
// /root/reference/src/planner/traj_optimization/src/minimum_control.cpp against the stand-in headers of this directory
// (oracle/Makefile target _ref).  TEST INFRASTRUCTURE ONLY.  What is the reference's here: the assembly of P, q, A, l, u
// (minimum_control.cpp:5-125) and the call sequence of solve() (:127-192).  What is NOT: Eigen, osqp-eigen, OSQP (see
// ref_shim/OsqpEigen/OsqpEigen.h: the QP is solved exactly through its KKT system instead of by ADMM).
#include <traj_optimization/minimum_control.h>

#include <cstring>
#include <iostream>

extern "C" {

// Runs MinimumControl::solve on one axis.  Outputs (all may be null): coef[6 M]; dense row-major P[n*n], A[m*n]; l[m], u[m];
// info[6] = {n, m, max_iter, warm_start, entries inserted into P, entries inserted into A}; eps_prim_inf.
// Returns 1 if solve() returned true, 0 if false, -1 if it threw.
int ref_minimum_control_solve(int n_seg, const double* pos_1d, const double* bound_vel, const double* bound_acc,
                              const double* time_vec, double* coef, double* P, double* A, double* l, double* u, int* info,
                              double* eps_prim_inf) {
    Eigen::VectorXd pos(n_seg + 1), T(n_seg);
    for (int i = 0; i <= n_seg; ++i) pos(i) = pos_1d[i];
    for (int i = 0; i < n_seg; ++i) T(i) = time_vec[i];
    Eigen::Vector2d v(bound_vel[0], bound_vel[1]), a(bound_acc[0], bound_acc[1]);
    traj_optimization::MinimumControl mc;
    bool ok = false;
    std::cout.setstate(std::ios_base::failbit);   // the reference dumps P, q, A, lb, ub on every solve (:154-158)
    try {
        ok = mc.solve(pos, v, a, T);
    } catch (...) {
        std::cout.clear();
        return -1;
    }
    std::cout.clear();
    const OsqpEigen::Captured& c = OsqpEigen::lastCaptured();
    if (P) std::memcpy(P, c.P.data(), sizeof(double) * c.P.size());
    if (A) std::memcpy(A, c.A.data(), sizeof(double) * c.A.size());
    if (l) std::memcpy(l, c.l.data(), sizeof(double) * c.l.size());
    if (u) std::memcpy(u, c.u.data(), sizeof(double) * c.u.size());
    if (info) { info[0] = c.n; info[1] = c.m; info[2] = c.max_iter; info[3] = c.warm_start; info[4] = c.p_inserted; info[5] = c.a_inserted; }
    if (eps_prim_inf) *eps_prim_inf = c.eps_prim_inf;
    if (ok && coef) {
        Eigen::VectorXd x = mc.getCoef1d();
        for (int i = 0; i < x.size(); ++i) coef[i] = x(i);
    }
    return ok ? 1 : 0;
}

}  // extern "C"
