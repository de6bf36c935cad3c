// Stand-in for <OsqpEigen/OsqpEigen.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  NOT osqp-eigen and NOT OSQP: it records
// what minimum_control.cpp:160-170 hands to the solver (settings, sizes, P, q, A, l, u) and "solves" the QP -- every row
// of which is an equality (l == u, minimum_control.cpp:98-125,146-147) -- exactly, through its dense KKT system in
// long double with partial pivoting.  The ADMM iteration of the real OSQP is restated separately in oracle/osqp_port.c.
#pragma once
#include <Eigen/Eigen>
#include <cmath>
#include <memory>
#include <vector>

namespace OsqpEigen {

class Settings {
  public:
    void setWarmStart(bool v) { warm_start = v; }
    void setPrimalInfeasibilityTollerance(double v) { eps_prim_inf = v; }
    void setPrimalInfeasibilityTolerance(double v) { eps_prim_inf = v; }
    void setMaxIteration(int v) { max_iter = v; }
    bool warm_start = false;
    double eps_prim_inf = 1e-4;
    int max_iter = 4000;
};

class Data {
  public:
    void setNumberOfVariables(int n) { n_ = n; }
    void setNumberOfConstraints(int m) { m_ = m; }
    bool setHessianMatrix(const Eigen::SparseMatrix<double>& P) { P_ = &P; return true; }
    bool setGradient(Eigen::VectorXd& q) { q_ = &q; return true; }
    bool setLinearConstraintsMatrix(const Eigen::SparseMatrix<double>& A) { A_ = &A; return true; }
    bool setLowerBound(Eigen::VectorXd& l) { l_ = &l; return true; }
    bool setUpperBound(Eigen::VectorXd& u) { u_ = &u; return true; }
    void clearHessianMatrix() { P_ = nullptr; }
    void clearLinearConstraintsMatrix() { A_ = nullptr; }
    int n_ = 0, m_ = 0;
    const Eigen::SparseMatrix<double>* P_ = nullptr;
    const Eigen::SparseMatrix<double>* A_ = nullptr;
    const Eigen::VectorXd *q_ = nullptr, *l_ = nullptr, *u_ = nullptr;
};

// what the last initSolver() saw, for the C wrapper (oracle/ref_shim/ref_capi.cpp)
struct Captured {
    int n = 0, m = 0, max_iter = 0, warm_start = 0, p_inserted = 0, a_inserted = 0;
    double eps_prim_inf = 0.0;
    std::vector<double> P, A, q, l, u;   // dense, row-major
};
inline Captured& lastCaptured() { static Captured c; return c; }

class Solver {
  public:
    Solver() : settings_(new Settings()), data_(new Data()) {}
    const std::unique_ptr<Settings>& settings() const { return settings_; }
    const std::unique_ptr<Data>& data() const { return data_; }
    bool initSolver() {
        const Data& d = *data_;
        if (!d.P_ || !d.A_ || !d.q_ || !d.l_ || !d.u_) return false;
        if (d.P_->rows() != d.n_ || d.P_->cols() != d.n_ || d.A_->rows() != d.m_ || d.A_->cols() != d.n_) return false;
        if (d.q_->size() != d.n_ || d.l_->size() != d.m_ || d.u_->size() != d.m_) return false;
        Captured& c = lastCaptured();
        c.n = d.n_; c.m = d.m_; c.max_iter = settings_->max_iter; c.warm_start = settings_->warm_start; c.eps_prim_inf = settings_->eps_prim_inf;
        c.p_inserted = d.P_->nInserted(); c.a_inserted = d.A_->nInserted();
        c.P.assign(static_cast<size_t>(c.n) * c.n, 0.0);
        c.A.assign(static_cast<size_t>(c.m) * c.n, 0.0);
        for (int i = 0; i < c.n; ++i) for (int j = 0; j < c.n; ++j) c.P[static_cast<size_t>(i) * c.n + j] = d.P_->at(i, j);
        for (int i = 0; i < c.m; ++i) for (int j = 0; j < c.n; ++j) c.A[static_cast<size_t>(i) * c.n + j] = d.A_->at(i, j);
        c.q = d.q_->raw(); c.l = d.l_->raw(); c.u = d.u_->raw();
        ready_ = true;
        return true;
    }
    bool solve() {
        if (!ready_) return false;
        const Captured& c = lastCaptured();
        for (int i = 0; i < c.m; ++i) if (c.l[i] != c.u[i]) return false;   // the stand-in only knows the all-equality case
        const int N = c.n + c.m;
        std::vector<long double> K(static_cast<size_t>(N) * (N + 1), 0.0L);
        auto at = [&](int i, int j) -> long double& { return K[static_cast<size_t>(i) * (N + 1) + j]; };
        for (int i = 0; i < c.n; ++i) {
            for (int j = 0; j < c.n; ++j) at(i, j) = c.P[static_cast<size_t>(i) * c.n + j];
            at(i, N) = -static_cast<long double>(c.q[i]);
        }
        for (int i = 0; i < c.m; ++i) {
            for (int j = 0; j < c.n; ++j) { at(c.n + i, j) = c.A[static_cast<size_t>(i) * c.n + j]; at(j, c.n + i) = c.A[static_cast<size_t>(i) * c.n + j]; }
            at(c.n + i, N) = c.l[i];
        }
        for (int col = 0; col < N; ++col) {   // Gauss-Jordan, partial pivoting
            int piv = col;
            for (int i = col + 1; i < N; ++i) if (fabsl(at(i, col)) > fabsl(at(piv, col))) piv = i;
            if (at(piv, col) == 0.0L) return false;
            if (piv != col) for (int j = 0; j <= N; ++j) std::swap(at(piv, j), at(col, j));
            const long double inv = 1.0L / at(col, col);
            for (int j = col; j <= N; ++j) at(col, j) *= inv;
            for (int i = 0; i < N; ++i) {
                if (i == col) continue;
                const long double f = at(i, col);
                if (f != 0.0L) for (int j = col; j <= N; ++j) at(i, j) -= f * at(col, j);
            }
        }
        sol_.resize(c.n);
        for (int i = 0; i < c.n; ++i) sol_[i] = static_cast<double>(at(i, N));
        return true;
    }
    Eigen::VectorXd getSolution() const { return sol_; }
    void clearSolver() { ready_ = false; }
  private:
    std::unique_ptr<Settings> settings_;
    std::unique_ptr<Data> data_;
    Eigen::VectorXd sol_;
    bool ready_ = false;
};

}  // namespace OsqpEigen
