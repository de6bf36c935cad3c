/*
 * osqp_port.c -- OSQP-faithful CPU restatement of the reference's solver path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; never the product.
 *
 * PARITY UNPINNED (see qp_oracle.c): the arithmetic of the reference path lives in a third-party
 * dependency that is absent from /root/reference -- OSQP, pinned only by the devcontainer to tag
 * v0.6.2 (.devcontainer/Dockerfile:51-53; submodule 3rd/osqp is empty, .gitmodules:1-3), driven
 * through osqp-eigen (unpinned, .gitmodules:4-6).  This file restates OSQP's *published* algorithm
 * (Stellato, Banjac, Goulart, Bemporad, Boyd: "OSQP: an operator splitting solver for quadratic
 * programs", Math. Prog. Comp. 12, 2020) with v0.6.2's default settings, anchored on the reference's
 * own call sites:
 *   problem data   minimum_control.cpp:5-125   (same entries, same row/column indexing, including the
 *                                               explicitly inserted structural zeros :55,61,64,65,70,71,83,90,91)
 *   settings       minimum_control.cpp:160-162 (warm_start=true [no effect: solver cleared every call,
 *                                               :188-190], eps_prim_inf=1e-3, max_iter=1000; all else default)
 *   call pattern   minimum_control.cpp:164-190 (data copy, initSolver = full setup, solve, clear) --
 *                                               once per axis, 3x per trajectory (test_minimum_jerk.cpp:75,100,125)
 * Restated pieces: Ruiz equilibration (10 passes) + cost scaling, rho_vec with 1e3*rho on equality rows,
 * quasi-definite KKT [[P+sigma I, A'],[A, -diag(1/rho)]], sparse LDL' (up-looking, elimination tree --
 * the scheme QDLDL implements), ADMM iteration with relaxation alpha=1.6, termination check every 25
 * iterations on unscaled residuals, adaptive rho with numeric refactorisation.
 * Known deviations (documented):
 *   - AMD ordering is replaced by a fixed banded ordering (segment variables interleaved with the knot's
 *     constraint rows), which is at least as good for this block-banded structure;
 *   - OSQP v0.6.2 picks adaptive_rho_interval from *measured setup time* (non-deterministic); the port
 *     pins it to 25 iterations (what the timing rule yields for problems this small; 100 is OSQP's
 *     rule when built without profiling) -- settable;
 *   - no polishing (default off), no printing (the reference dumps P, q, A, l, u to stdout on every
 *     solve, minimum_control.cpp:154-158, and OSQP runs verbose: both excluded from timing).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* pthread_setaffinity_np, sched_getaffinity: the all-cores baseline pins one thread per core */
#endif
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

#define OSQP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_TOL 1e-4
#define RHO_EQ_OVER_RHO_INEQ 1e3

enum { PORT_SOLVED = 1, PORT_SOLVED_INACCURATE = 2, PORT_MAX_ITER_REACHED = -2, PORT_PRIMAL_INFEASIBLE = -3,
       PORT_DUAL_INFEASIBLE = -4, PORT_UNSOLVED = -10 };

typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, adaptive_rho_tolerance;
    int max_iter, scaling, adaptive_rho, adaptive_rho_interval, check_termination;
} port_settings;

/* OSQP v0.6.2 defaults (include/constants.h) overridden by minimum_control.cpp:160-162 */
void osqp_port_default_settings(port_settings* s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
    s->eps_abs = 1e-3; s->eps_rel = 1e-3;
    s->eps_prim_inf = 1e-3;   /* setPrimalInfeasibilityTollerance(1e-3), minimum_control.cpp:161 */
    s->eps_dual_inf = 1e-4;
    s->adaptive_rho_tolerance = 5.0;
    s->max_iter = 1000;       /* setMaxIteration(1000), minimum_control.cpp:162 */
    s->scaling = 10; s->adaptive_rho = 1; s->adaptive_rho_interval = 25; s->check_termination = 25;
}

typedef struct { int nr, nc, nnz; int* p; int* i; double* x; } csc;

static csc csc_alloc(int nr, int nc, int nnz) {
    csc m; m.nr = nr; m.nc = nc; m.nnz = nnz;
    m.p = (int*)calloc((size_t)nc + 1, sizeof(int));
    m.i = (int*)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    m.x = (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    return m;
}
static void csc_free(csc* m) { free(m->p); free(m->i); free(m->x); }

/* triplets -> CSC (entries within a column sorted by row; duplicates not expected) */
typedef struct { int r, c; double v; } trip;
static int trip_cmp(const void* a, const void* b) {
    const trip* x = (const trip*)a; const trip* y = (const trip*)b;
    if (x->c != y->c) return x->c - y->c;
    return x->r - y->r;
}
static csc csc_from_triplets(int nr, int nc, trip* t, int nt) {
    qsort(t, (size_t)nt, sizeof(trip), trip_cmp);
    csc m = csc_alloc(nr, nc, nt);
    for (int k = 0; k < nt; ++k) m.p[t[k].c + 1]++;
    for (int c = 0; c < nc; ++c) m.p[c + 1] += m.p[c];
    for (int k = 0; k < nt; ++k) { m.i[k] = t[k].r; m.x[k] = t[k].v; }
    return m;
}

static double falling(int k, int d) { double f = 1; for (int j = 0; j < d; ++j) f *= (double)(k - j); return f; }

/* Problem data exactly as minimum_control.cpp:5-125 builds it (r=3), same pattern for r=4.
 * P: upper triangle only (osqp-eigen hands OSQP the upper-triangular view).  A keeps the structural zeros. */
/* Extra inequality rows (north-star extension, no reference counterpart: the reference only builds the equality rows above;
 * OSQP itself accepts any l <= A x <= u, minimum_control.cpp:146-147,164-180): row e bounds the derivative d of segment seg at
 * local time t, i.e. the monomial row  k!/(k-d)! t^(k-d)  on that segment's coefficients. */
typedef struct { int n; const int* seg; const double* t; const int* d; const double* lo; const double* hi; } extra_rows;

static void build_problem(int r, int M, const double* T, const double* pos, const double* bcs, const double* bce,
                          const double* corr_lo, const double* corr_hi, const extra_rows* ex, csc* P, csc* A, double* l, double* u) {
    const int R = 2 * r, n = R * M, m0 = 2 * r + (r + 1) * (M - 1), nex = ex ? ex->n : 0, m = m0 + nex;
    trip* tp = (trip*)malloc(sizeof(trip) * (size_t)(r * (r + 1) / 2 * M));
    int np = 0;
    for (int i = 0; i < M; ++i)
        for (int a = r; a < R; ++a)
            for (int c = a; c < R; ++c) {
                const int e = a + c - 2 * r + 1;
                tp[np].r = R * i + a; tp[np].c = R * i + c;
                tp[np].v = falling(a, r) * falling(c, r) * pow(T[i], e) / (double)e;
                ++np;
            }
    *P = csc_from_triplets(n, n, tp, np);
    free(tp);
    trip* ta = (trip*)malloc(sizeof(trip) * (size_t)(r + (M + 1) * (R + r * (R + r + 1)) + nex * R));
    int na = 0;
#define ADD(rr, cc, vv) do { ta[na].r = (rr); ta[na].c = (cc); ta[na].v = (vv); ++na; } while (0)
    for (int d = 0; d < r; ++d) ADD(d, d, falling(d, d));                                 /* :29-31 */
    for (int i = 0; i < M - 1; ++i) {
        for (int k = 0; k < R; ++k) ADD(r + (r + 1) * i, R * i + k, pow(T[i], k));          /* :34-42 */
        for (int d = 0; d < r; ++d) {                                                       /* :45-74 */
            const int row = (r + 1) * (i + 1) + d;
            for (int k = 0; k < R; ++k) ADD(row, R * i + k, k >= d ? falling(k, d) * pow(T[i], k - d) : 0.0);
            for (int k = 0; k <= d; ++k) ADD(row, R * (i + 1) + k, k == d ? -falling(d, d) : 0.0);
        }
    }
    {                                                                                       /* :77-95 */
        const int i = M - 1;
        for (int k = 0; k < R; ++k) ADD(r + (r + 1) * i, R * i + k, pow(T[i], k));
        for (int d = 1; d < r; ++d)
            for (int k = 0; k < R; ++k) ADD((r + 1) * M + (d - 1), R * i + k, k >= d ? falling(k, d) * pow(T[i], k - d) : 0.0);
    }
    for (int e = 0; e < nex; ++e)
        for (int k = ex->d[e]; k < R; ++k) ADD(m0 + e, R * ex->seg[e] + k, falling(k, ex->d[e]) * pow(ex->t[e], k - ex->d[e]));
#undef ADD
    *A = csc_from_triplets(m, n, ta, na);
    free(ta);
    for (int j = 0; j < m; ++j) l[j] = 0.0;                                                 /* :98-125 */
    l[0] = pos[0];
    for (int d = 1; d < r; ++d) l[d] = bcs[d - 1];
    l[r + (r + 1) * (M - 1)] = pos[M];
    for (int d = 1; d < r; ++d) l[r + (r + 1) * (M - 1) + d] = bce[d - 1];
    for (int i = 0; i < M - 1; ++i) l[r + (r + 1) * i] = pos[i + 1];
    for (int j = 0; j < m; ++j) u[j] = l[j];
    /* North-star extension (SURVEY.md section 8-a'): corridor = the interior waypoint equality p_i(T_i) = w_{i+1}
     * becomes l <= p_i(T_i) <= u -- the only rows with l < u, and the reason OSQP's ADMM is needed at all. */
    if (corr_lo && corr_hi)
        for (int i = 0; i < M - 1; ++i) { l[r + (r + 1) * i] = corr_lo[i]; u[r + (r + 1) * i] = corr_hi[i]; }
    for (int e = 0; e < nex; ++e) { l[m0 + e] = ex->lo[e]; u[m0 + e] = ex->hi[e]; }
}

/* ------------------------------------------------------------------ scaling (Ruiz) */
static double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return v > MAX_SCALING ? MAX_SCALING : v; }

static void inf_norm_cols_sym_triu(const csc* P, double* E) { /* symmetric matrix stored as upper triangle */
    for (int j = 0; j < P->nc; ++j) E[j] = 0;
    for (int j = 0; j < P->nc; ++j)
        for (int p = P->p[j]; p < P->p[j + 1]; ++p) {
            const int i = P->i[p]; const double a = fabs(P->x[p]);
            if (a > E[j]) E[j] = a;
            if (i != j && a > E[i]) E[i] = a;
        }
}
static void inf_norm_cols(const csc* A, double* E) {
    for (int j = 0; j < A->nc; ++j) { E[j] = 0; for (int p = A->p[j]; p < A->p[j + 1]; ++p) if (fabs(A->x[p]) > E[j]) E[j] = fabs(A->x[p]); }
}
static void inf_norm_rows(const csc* A, double* E) {
    for (int i = 0; i < A->nr; ++i) E[i] = 0;
    for (int j = 0; j < A->nc; ++j) for (int p = A->p[j]; p < A->p[j + 1]; ++p) if (fabs(A->x[p]) > E[A->i[p]]) E[A->i[p]] = fabs(A->x[p]);
}

typedef struct { double* D; double* E; double* Dinv; double* Einv; double c, cinv; } scaling_t;

static void scale_data(int n, int m, csc* P, csc* A, double* q, double* l, double* u, int iters, scaling_t* sc) {
    double* Dt = (double*)malloc(sizeof(double) * n); double* DtA = (double*)malloc(sizeof(double) * n);
    double* Et = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    for (int j = 0; j < n; ++j) sc->D[j] = 1.0;
    for (int i = 0; i < m; ++i) sc->E[i] = 1.0;
    sc->c = 1.0;
    for (int it = 0; it < iters; ++it) {
        inf_norm_cols_sym_triu(P, Dt);
        inf_norm_cols(A, DtA);
        for (int j = 0; j < n; ++j) Dt[j] = Dt[j] > DtA[j] ? Dt[j] : DtA[j];
        inf_norm_rows(A, Et);
        for (int j = 0; j < n; ++j) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
        for (int i = 0; i < m; ++i) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
        for (int j = 0; j < n; ++j) for (int p = P->p[j]; p < P->p[j + 1]; ++p) P->x[p] *= Dt[j] * Dt[P->i[p]];
        for (int j = 0; j < n; ++j) for (int p = A->p[j]; p < A->p[j + 1]; ++p) A->x[p] *= Dt[j] * Et[A->i[p]];
        for (int j = 0; j < n; ++j) { q[j] *= Dt[j]; sc->D[j] *= Dt[j]; }
        for (int i = 0; i < m; ++i) sc->E[i] *= Et[i];
        /* cost scaling */
        inf_norm_cols_sym_triu(P, Dt);
        double mean = 0; for (int j = 0; j < n; ++j) mean += Dt[j]; mean /= (double)n;
        double nq = 0; for (int j = 0; j < n; ++j) if (fabs(q[j]) > nq) nq = fabs(q[j]);
        nq = limit_scaling(nq);
        double ct = mean > nq ? mean : nq;
        ct = 1.0 / limit_scaling(ct);
        for (int p = 0; p < P->nnz; ++p) P->x[p] *= ct;
        for (int j = 0; j < n; ++j) q[j] *= ct;
        sc->c *= ct;
    }
    sc->cinv = 1.0 / sc->c;
    for (int j = 0; j < n; ++j) sc->Dinv[j] = 1.0 / sc->D[j];
    for (int i = 0; i < m; ++i) { sc->Einv[i] = 1.0 / sc->E[i]; l[i] *= sc->E[i]; u[i] *= sc->E[i]; }
    free(Dt); free(DtA); free(Et);
}

/* ------------------------------------------------------------------ sparse LDL' (up-looking) */
typedef struct {
    int N; int* Kp; int* Ki; double* Kx;  /* permuted KKT, upper triangle, CSC */
    int* rho_pos;                           /* position in Kx of the -1/rho diagonal of constraint i */
    int* parent; int* Lp; int* Li; double* Lx; double* D; double* Dinv; int* lnz; int* flag; int* pattern; double* y;
    int* perm; double* bp;
} ldl_t;

static void ldl_symbolic(ldl_t* f) {
    const int N = f->N;
    for (int k = 0; k < N; ++k) {
        f->parent[k] = -1; f->flag[k] = k; f->lnz[k] = 0;
        for (int p = f->Kp[k]; p < f->Kp[k + 1]; ++p) {
            int i = f->Ki[p];
            if (i < k)
                for (; f->flag[i] != k; i = f->parent[i]) {
                    if (f->parent[i] == -1) f->parent[i] = k;
                    f->lnz[i]++; f->flag[i] = k;
                }
        }
    }
    f->Lp[0] = 0;
    for (int k = 0; k < N; ++k) f->Lp[k + 1] = f->Lp[k] + f->lnz[k];
}

static int ldl_numeric(ldl_t* f) {
    const int N = f->N;
    for (int k = 0; k < N; ++k) {
        int top = N;
        f->y[k] = 0.0; f->flag[k] = k; f->lnz[k] = 0;
        for (int p = f->Kp[k]; p < f->Kp[k + 1]; ++p) {
            int i = f->Ki[p];
            if (i <= k) {
                f->y[i] += f->Kx[p];
                int len = 0;
                for (; f->flag[i] != k; i = f->parent[i]) { f->pattern[len++] = i; f->flag[i] = k; }
                while (len > 0) f->pattern[--top] = f->pattern[--len];
            }
        }
        f->D[k] = f->y[k]; f->y[k] = 0.0;
        for (; top < N; ++top) {
            const int i = f->pattern[top];
            const double yi = f->y[i];
            f->y[i] = 0.0;
            const int p2 = f->Lp[i] + f->lnz[i];
            for (int p = f->Lp[i]; p < p2; ++p) f->y[f->Li[p]] -= f->Lx[p] * yi;
            const double lki = yi * f->Dinv[i];
            f->D[k] -= lki * yi;
            f->Li[p2] = k; f->Lx[p2] = lki; f->lnz[i]++;
        }
        if (f->D[k] == 0.0) return -1;
        f->Dinv[k] = 1.0 / f->D[k];
    }
    return 0;
}

static void ldl_solve(const ldl_t* f, double* b) { /* b in original ordering, solved in place */
    const int N = f->N;
    double* x = f->bp;
    for (int k = 0; k < N; ++k) x[k] = b[f->perm[k]];
    for (int j = 0; j < N; ++j) { const double xj = x[j]; for (int p = f->Lp[j]; p < f->Lp[j + 1]; ++p) x[f->Li[p]] -= f->Lx[p] * xj; }
    for (int j = 0; j < N; ++j) x[j] *= f->Dinv[j];
    for (int j = N - 1; j >= 0; --j) { double xj = x[j]; for (int p = f->Lp[j]; p < f->Lp[j + 1]; ++p) xj -= f->Lx[p] * x[f->Li[p]]; x[j] = xj; }
    for (int k = 0; k < N; ++k) b[f->perm[k]] = x[k];
}

/* Banded ordering: [start rows][x seg 0][knot 0 rows][x seg 1]...[x seg M-1][end rows] */
static void kkt_ordering(int r, int M, const extra_rows* ex, int* perm /* new -> old */) {
    const int R = 2 * r, n = R * M, m0 = 2 * r + (r + 1) * (M - 1);
    int k = 0;
    for (int d = 0; d < r; ++d) perm[k++] = n + d;
    for (int i = 0; i < M; ++i) {
        for (int c = 0; c < R; ++c) perm[k++] = R * i + c;
        for (int e = 0; ex && e < ex->n; ++e)
            if (ex->seg[e] == i) perm[k++] = n + m0 + e;     /* extra rows right behind the coefficients of their segment */
        if (i < M - 1) {
            perm[k++] = n + r + (r + 1) * i;
            for (int d = 0; d < r; ++d) perm[k++] = n + (r + 1) * (i + 1) + d;
        } else {
            perm[k++] = n + r + (r + 1) * i;
            for (int d = 1; d < r; ++d) perm[k++] = n + (r + 1) * M + (d - 1);
        }
    }
}

static void kkt_build(ldl_t* f, int n, int m, const csc* P, const csc* A, double sigma, const double* rho_inv, const int* perm) {
    const int N = n + m;
    int* inv = (int*)malloc(sizeof(int) * N);
    for (int k = 0; k < N; ++k) inv[perm[k]] = k;
    const int cap = P->nnz + n + A->nnz + m;
    trip* t = (trip*)malloc(sizeof(trip) * (size_t)cap);
    int nt = 0;
    char* has_diag = (char*)calloc((size_t)n, 1);
    for (int j = 0; j < n; ++j)
        for (int p = P->p[j]; p < P->p[j + 1]; ++p) {
            int a = inv[P->i[p]], b = inv[j];
            double v = P->x[p];
            if (P->i[p] == j) { v += sigma; has_diag[j] = 1; }
            t[nt].r = a < b ? a : b; t[nt].c = a < b ? b : a; t[nt].v = v; ++nt;
        }
    for (int j = 0; j < n; ++j) if (!has_diag[j]) { t[nt].r = t[nt].c = inv[j]; t[nt].v = sigma; ++nt; }
    for (int j = 0; j < n; ++j)
        for (int p = A->p[j]; p < A->p[j + 1]; ++p) {
            int a = inv[n + A->i[p]], b = inv[j];
            t[nt].r = a < b ? a : b; t[nt].c = a < b ? b : a; t[nt].v = A->x[p]; ++nt;
        }
    for (int i = 0; i < m; ++i) { t[nt].r = t[nt].c = inv[n + i]; t[nt].v = -rho_inv[i]; ++nt; }
    qsort(t, (size_t)nt, sizeof(trip), trip_cmp);
    f->N = N;
    f->Kp = (int*)calloc((size_t)N + 1, sizeof(int)); f->Ki = (int*)malloc(sizeof(int) * nt); f->Kx = (double*)malloc(sizeof(double) * nt);
    f->rho_pos = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
    for (int k = 0; k < nt; ++k) f->Kp[t[k].c + 1]++;
    for (int c = 0; c < N; ++c) f->Kp[c + 1] += f->Kp[c];
    for (int k = 0; k < nt; ++k) {
        f->Ki[k] = t[k].r; f->Kx[k] = t[k].v;
        if (t[k].r == t[k].c && perm[t[k].c] >= n) f->rho_pos[perm[t[k].c] - n] = k;
    }
    free(t); free(inv); free(has_diag);
    f->parent = (int*)malloc(sizeof(int) * N); f->Lp = (int*)malloc(sizeof(int) * (N + 1)); f->lnz = (int*)malloc(sizeof(int) * N);
    f->flag = (int*)malloc(sizeof(int) * N); f->pattern = (int*)malloc(sizeof(int) * N); f->y = (double*)calloc((size_t)N, sizeof(double));
    f->D = (double*)malloc(sizeof(double) * N); f->Dinv = (double*)malloc(sizeof(double) * N); f->bp = (double*)malloc(sizeof(double) * N);
    f->perm = (int*)malloc(sizeof(int) * N); memcpy(f->perm, perm, sizeof(int) * N);
    ldl_symbolic(f);
    const int lnnz = f->Lp[N];
    f->Li = (int*)malloc(sizeof(int) * (lnnz > 0 ? lnnz : 1)); f->Lx = (double*)malloc(sizeof(double) * (lnnz > 0 ? lnnz : 1));
}
static void ldl_free(ldl_t* f) {
    free(f->Kp); free(f->Ki); free(f->Kx); free(f->rho_pos); free(f->parent); free(f->Lp); free(f->Li); free(f->Lx);
    free(f->D); free(f->Dinv); free(f->lnz); free(f->flag); free(f->pattern); free(f->y); free(f->perm); free(f->bp);
}

/* ------------------------------------------------------------------ small vector helpers */
static double norm_inf(const double* v, int n) { double s = 0; for (int i = 0; i < n; ++i) if (fabs(v[i]) > s) s = fabs(v[i]); return s; }
static double norm_inf_scaled(const double* s, const double* v, int n) { double r = 0; for (int i = 0; i < n; ++i) { double a = fabs(s[i] * v[i]); if (a > r) r = a; } return r; }
static void mat_vec(const csc* A, const double* x, double* y) { /* y = A x */
    for (int i = 0; i < A->nr; ++i) y[i] = 0;
    for (int j = 0; j < A->nc; ++j) { const double xj = x[j]; for (int p = A->p[j]; p < A->p[j + 1]; ++p) y[A->i[p]] += A->x[p] * xj; }
}
static void mat_tpose_vec(const csc* A, const double* x, double* y) { /* y = A' x */
    for (int j = 0; j < A->nc; ++j) { double s = 0; for (int p = A->p[j]; p < A->p[j + 1]; ++p) s += A->x[p] * x[A->i[p]]; y[j] = s; }
}
static void sym_triu_vec(const csc* P, const double* x, double* y) { /* y = P x, P symmetric stored upper */
    for (int j = 0; j < P->nc; ++j) y[j] = 0;
    for (int j = 0; j < P->nc; ++j)
        for (int p = P->p[j]; p < P->p[j + 1]; ++p) {
            const int i = P->i[p];
            y[i] += P->x[p] * x[j];
            if (i != j) y[j] += P->x[p] * x[i];
        }
}

typedef struct { int iters; int status; int rho_updates; double pri_res, dua_res, rho; } port_info;

/* One axis: setup + solve + cleanup, as MinimumControl::solve does per call (minimum_control.cpp:164-190). */
int osqp_port_solve_axis_corridor(int r, int M, const double* pos, const double* bcs, const double* bce, const double* T,
                                  const double* corr_lo, const double* corr_hi,
                                  const port_settings* user, double* coef, port_info* info);

int osqp_port_solve_axis(int r, int M, const double* pos, const double* bcs, const double* bce, const double* T,
                         const port_settings* user, double* coef, port_info* info) {
    return osqp_port_solve_axis_corridor(r, M, pos, bcs, bce, T, NULL, NULL, user, coef, info);
}

/* corr_lo / corr_hi: M-1 bounds for the interior waypoints 1..M-1 of this axis (NULL: equalities at pos). */
static int solve_axis_rows(int r, int M, const double* pos, const double* bcs, const double* bce, const double* T,
                           const double* corr_lo, const double* corr_hi, const extra_rows* ex,
                           const port_settings* user, double* coef, port_info* info);

int osqp_port_solve_axis_corridor(int r, int M, const double* pos, const double* bcs, const double* bce, const double* T,
                                  const double* corr_lo, const double* corr_hi,
                                  const port_settings* user, double* coef, port_info* info) {
    return solve_axis_rows(r, M, pos, bcs, bce, T, corr_lo, corr_hi, NULL, user, coef, info);
}

static int solve_axis_rows(int r, int M, const double* pos, const double* bcs, const double* bce, const double* T,
                           const double* corr_lo, const double* corr_hi, const extra_rows* ex,
                           const port_settings* user, double* coef, port_info* info) {
    if ((r != 3 && r != 4) || M < 1) return -2;
    port_settings st;
    if (user) st = *user; else osqp_port_default_settings(&st);
    const int n = 2 * r * M, m = 2 * r + (r + 1) * (M - 1) + (ex ? ex->n : 0), N = n + m;
    csc P, A;
    double* q = (double*)calloc((size_t)n, sizeof(double));                 /* getGradient: q = 0, :21-24 */
    double* l = (double*)malloc(sizeof(double) * m); double* u = (double*)malloc(sizeof(double) * m);
    build_problem(r, M, T, pos, bcs, bce, corr_lo, corr_hi, ex, &P, &A, l, u);

    /* ---- osqp_setup ---- */
    scaling_t sc;
    sc.D = (double*)malloc(sizeof(double) * n); sc.Dinv = (double*)malloc(sizeof(double) * n);
    sc.E = (double*)malloc(sizeof(double) * m); sc.Einv = (double*)malloc(sizeof(double) * m);
    if (st.scaling > 0) scale_data(n, m, &P, &A, q, l, u, st.scaling, &sc);
    else { for (int j = 0; j < n; ++j) sc.D[j] = sc.Dinv[j] = 1; for (int i = 0; i < m; ++i) sc.E[i] = sc.Einv[i] = 1; sc.c = sc.cinv = 1; }
    double rho = st.rho;
    double* rho_vec = (double*)malloc(sizeof(double) * m); double* rho_inv = (double*)malloc(sizeof(double) * m);
    int* ctype = (int*)malloc(sizeof(int) * m);
    for (int i = 0; i < m; ++i) {
        if (l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING) { ctype[i] = -1; rho_vec[i] = RHO_MIN; }
        else if (u[i] - l[i] < RHO_TOL) { ctype[i] = 1; rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * rho; }
        else { ctype[i] = 0; rho_vec[i] = rho; }
        rho_inv[i] = 1.0 / rho_vec[i];
    }
    int* perm = (int*)malloc(sizeof(int) * N);
    kkt_ordering(r, M, ex, perm);
    ldl_t F;
    kkt_build(&F, n, m, &P, &A, st.sigma, rho_inv, perm);
    int rc = ldl_numeric(&F);

    /* ---- osqp_solve (cold start: x = z = y = 0) ---- */
    double* x = (double*)calloc((size_t)n, sizeof(double)); double* z = (double*)calloc((size_t)m, sizeof(double));
    double* y = (double*)calloc((size_t)m, sizeof(double));
    double* xp = (double*)calloc((size_t)n, sizeof(double)); double* zp = (double*)calloc((size_t)m, sizeof(double));
    double* xz = (double*)calloc((size_t)N, sizeof(double));
    double* Ax = (double*)calloc((size_t)m, sizeof(double)); double* Px = (double*)calloc((size_t)n, sizeof(double));
    double* Aty = (double*)calloc((size_t)n, sizeof(double)); double* tn = (double*)calloc((size_t)N, sizeof(double));
    double* dy = (double*)calloc((size_t)m, sizeof(double));   /* delta_y of the last iteration (the infeasibility certificate candidate) */
    int status = PORT_UNSOLVED, iter = 0, rho_updates = 0;
    double pri_res = 0, dua_res = 0;
    for (iter = 1; rc == 0 && iter <= st.max_iter; ++iter) {
        double* t;
        t = x; x = xp; xp = t; t = z; z = zp; zp = t;   /* swap: xp, zp hold the previous iterate */
        for (int j = 0; j < n; ++j) xz[j] = st.sigma * xp[j] - q[j];
        for (int i = 0; i < m; ++i) xz[n + i] = zp[i] - rho_inv[i] * y[i];
        ldl_solve(&F, xz);
        for (int i = 0; i < m; ++i) xz[n + i] = zp[i] + rho_inv[i] * (xz[n + i] - y[i]);   /* z_tilde */
        for (int j = 0; j < n; ++j) x[j] = st.alpha * xz[j] + (1.0 - st.alpha) * xp[j];
        for (int i = 0; i < m; ++i) {
            const double zr = st.alpha * xz[n + i] + (1.0 - st.alpha) * zp[i];
            double zn = zr + rho_inv[i] * y[i];
            zn = zn < l[i] ? l[i] : (zn > u[i] ? u[i] : zn);                               /* projection */
            z[i] = zn;
            dy[i] = rho_vec[i] * (zr - zn);
            y[i] += dy[i];
        }
        const int can_check = st.check_termination && (iter % st.check_termination == 0);
        const int can_adapt = st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0);
        if (can_check || can_adapt || iter == st.max_iter) {
            mat_vec(&A, x, Ax); sym_triu_vec(&P, x, Px); mat_tpose_vec(&A, y, Aty);
            for (int i = 0; i < m; ++i) tn[i] = Ax[i] - z[i];
            pri_res = norm_inf_scaled(sc.Einv, tn, m);
            for (int j = 0; j < n; ++j) tn[j] = q[j] + Aty[j] + Px[j];
            dua_res = sc.cinv * norm_inf_scaled(sc.Dinv, tn, n);
            if (can_check || iter == st.max_iter) {
                double a1 = norm_inf_scaled(sc.Einv, z, m), a2 = norm_inf_scaled(sc.Einv, Ax, m);
                const double eps_prim = st.eps_abs + st.eps_rel * (a1 > a2 ? a1 : a2);
                double b1 = norm_inf_scaled(sc.Dinv, q, n), b2 = norm_inf_scaled(sc.Dinv, Aty, n), b3 = norm_inf_scaled(sc.Dinv, Px, n);
                double bm = b1 > b2 ? b1 : b2; bm = bm > b3 ? bm : b3;
                const double eps_dual = st.eps_abs + st.eps_rel * sc.cinv * bm;
                if (pri_res < eps_prim && dua_res < eps_dual) { status = PORT_SOLVED; break; }
                /* Primal infeasibility (paper section 3.4; the one tolerance the reference sets, minimum_control.cpp:161): delta_y, projected
                 * onto the polar of the recession cone of [l, u], is a certificate when  ||A' dy||_inf <= eps ||dy||_inf  and
                 * u' max(dy, 0) + l' min(dy, 0) <= -eps ||dy||_inf  (norms unscaled: scaled_termination = 0).  It cannot fire for the
                 * reference's own full-row-rank equality QP; it does for contradictory corridor / general rows (extension).  The dual
                 * infeasibility test cannot fire at all (P is positive definite on null(A)): omitted. */
                {
                    double nd = 0.0, lhs = 0.0;
                    for (int i = 0; i < m; ++i) {
                        double d_ = dy[i];
                        if (u[i] > OSQP_INFTY * MIN_SCALING) d_ = (l[i] < -OSQP_INFTY * MIN_SCALING) ? 0.0 : (d_ < 0.0 ? d_ : 0.0);
                        else if (l[i] < -OSQP_INFTY * MIN_SCALING) d_ = d_ > 0.0 ? d_ : 0.0;
                        tn[n + i] = d_;
                        const double a_ = fabs(sc.E[i] * d_);
                        if (a_ > nd) nd = a_;
                    }
                    if (nd > 1.0 / OSQP_INFTY) {
                        for (int i = 0; i < m; ++i) {
                            const double d_ = tn[n + i];
                            if (d_ > 0.0) lhs += u[i] * d_; else if (d_ < 0.0) lhs += l[i] * d_;
                        }
                        if (lhs < -st.eps_prim_inf * nd) {
                            mat_tpose_vec(&A, tn + n, tn);
                            if (norm_inf_scaled(sc.Dinv, tn, n) < st.eps_prim_inf * nd) { status = PORT_PRIMAL_INFEASIBLE; break; }
                        }
                    }
                }
            }
            if (can_adapt) {
                for (int i = 0; i < m; ++i) tn[i] = Ax[i] - z[i];
                double pr = norm_inf(tn, m);
                double nz = norm_inf(z, m), nax = norm_inf(Ax, m);
                pr /= ((nz > nax ? nz : nax) + 1e-10);
                for (int j = 0; j < n; ++j) tn[j] = q[j] + Aty[j] + Px[j];
                double dr = norm_inf(tn, n);
                double c1 = norm_inf(q, n), c2 = norm_inf(Aty, n), c3 = norm_inf(Px, n);
                double cm = c1 > c2 ? c1 : c2; cm = cm > c3 ? cm : c3;
                dr /= (cm + 1e-10);
                double rho_new = rho * sqrt(pr / (dr + 1e-10));
                rho_new = rho_new < RHO_MIN ? RHO_MIN : (rho_new > RHO_MAX ? RHO_MAX : rho_new);
                if (rho_new > rho * st.adaptive_rho_tolerance || rho_new < rho / st.adaptive_rho_tolerance) {
                    rho = rho_new; ++rho_updates;
                    for (int i = 0; i < m; ++i) {
                        rho_vec[i] = ctype[i] == 1 ? RHO_EQ_OVER_RHO_INEQ * rho : (ctype[i] == 0 ? rho : RHO_MIN);
                        rho_inv[i] = 1.0 / rho_vec[i];
                        F.Kx[F.rho_pos[i]] = -rho_inv[i];
                    }
                    rc = ldl_numeric(&F);
                }
            }
        }
    }
    if (status == PORT_UNSOLVED) { status = PORT_MAX_ITER_REACHED; iter = st.max_iter; }
    if (rc != 0) status = PORT_UNSOLVED;
    for (int j = 0; j < n; ++j) coef[j] = sc.D[j] * x[j];     /* unscale */
    if (info) { info->iters = iter; info->status = status; info->rho_updates = rho_updates; info->pri_res = pri_res; info->dua_res = dua_res; info->rho = rho; }

    /* ---- cleanup (clearSolver) ---- */
    ldl_free(&F); csc_free(&P); csc_free(&A);
    free(q); free(l); free(u); free(sc.D); free(sc.Dinv); free(sc.E); free(sc.Einv); free(rho_vec); free(rho_inv); free(ctype); free(perm);
    free(x); free(z); free(y); free(xp); free(zp); free(xz); free(Ax); free(Px); free(Aty); free(tn); free(dy);
    return status == PORT_SOLVED ? 0 : 1;
}

/* ------------------------------------------------------------------ batch driver (C-ABI layout, optional threads) */
typedef struct {
    int r, b0, b1; const int* so; const double* wp; const double* times; const double* bc;
    const port_settings* st; double* out; int* status; int* iters;
    const double* clo; const double* chi;  /* optional corridor bounds, waypoint layout [sum(M_b+1)][3] */
    int K; const double* rtau; const int* rdrv; const double* rlo; const double* rhi;  /* optional rows, C-ABI layout of uavqp_solve_rows_batch_device */
} job_t;

static void* job_run(void* arg) {
    job_t* j = (job_t*)arg;
    const int r = j->r;
    for (int b = j->b0; b < j->b1; ++b) {
        const int s0 = j->so[b], M = j->so[b + 1] - s0;
        if (M < 1) { if (j->status) j->status[b] = PORT_UNSOLVED; continue; }
        double* pos = (double*)malloc(sizeof(double) * (M + 1));
        const double* w = j->wp + 3 * (size_t)(s0 + b);
        int worst = PORT_SOLVED, it_max = 0;
        for (int ax = 0; ax < 3; ++ax) {
            double bs[3], be[3];
            for (int i = 0; i <= M; ++i) pos[i] = w[3 * i + ax];
            for (int d = 0; d < r - 1; ++d) {
                bs[d] = j->bc[(((size_t)b * 2 + 0) * (r - 1) + d) * 3 + ax];
                be[d] = j->bc[(((size_t)b * 2 + 1) * (r - 1) + d) * 3 + ax];
            }
            port_info info;
            double *lo = NULL, *hi = NULL;
            if (j->clo && j->chi && M > 1) {
                lo = (double*)malloc(sizeof(double) * (M - 1)); hi = (double*)malloc(sizeof(double) * (M - 1));
                for (int i = 1; i < M; ++i) { lo[i - 1] = j->clo[3 * (size_t)(s0 + b + i) + ax]; hi[i - 1] = j->chi[3 * (size_t)(s0 + b + i) + ax]; }
            }
            extra_rows ex = {0, NULL, NULL, NULL, NULL, NULL};
            int* eseg = NULL; int* ed = NULL; double* et = NULL; double* elo = NULL; double* ehi = NULL;
            if (j->K > 0 && j->rdrv) {
                const int cap = M * j->K;
                eseg = (int*)malloc(sizeof(int) * cap); ed = (int*)malloc(sizeof(int) * cap); et = (double*)malloc(sizeof(double) * cap);
                elo = (double*)malloc(sizeof(double) * cap); ehi = (double*)malloc(sizeof(double) * cap);
                for (int i = 0; i < M; ++i)
                    for (int q = 0; q < j->K; ++q) {
                        const size_t row = (size_t)(s0 + i) * j->K + q;
                        if (j->rdrv[row] < 0) continue;
                        eseg[ex.n] = i; ed[ex.n] = j->rdrv[row]; et[ex.n] = j->rtau[row] * j->times[s0 + i];
                        elo[ex.n] = j->rlo[row * 3 + ax]; ehi[ex.n] = j->rhi[row * 3 + ax];
                        ++ex.n;
                    }
                ex.seg = eseg; ex.d = ed; ex.t = et; ex.lo = elo; ex.hi = ehi;
            }
            solve_axis_rows(r, M, pos, bs, be, j->times + s0, lo, hi, ex.n > 0 ? &ex : NULL, j->st, j->out + (size_t)3 * 2 * r * s0 + (size_t)ax * 2 * r * M, &info);
            free(lo); free(hi); free(eseg); free(ed); free(et); free(elo); free(ehi);
            if (info.status != PORT_SOLVED && worst != PORT_PRIMAL_INFEASIBLE) worst = info.status;   /* an infeasible axis decides the trajectory */
            if (info.iters > it_max) it_max = info.iters;
        }
        if (j->status) j->status[b] = worst;
        if (j->iters) j->iters[b] = it_max;
        free(pos);
    }
    return NULL;
}

int osqp_port_solve_batch_corridor(int r, int n_traj, const int* seg_offsets, const double* waypoints, const double* times,
                                   const double* bc, const double* corr_lo, const double* corr_hi, const port_settings* st,
                                   double* coef_out, int* status_out, int* iters_out, int n_threads);

int osqp_port_solve_batch(int r, int n_traj, const int* seg_offsets, const double* waypoints, const double* times,
                          const double* bc, const port_settings* st, double* coef_out, int* status_out, int* iters_out,
                          int n_threads) {
    return osqp_port_solve_batch_corridor(r, n_traj, seg_offsets, waypoints, times, bc, NULL, NULL, st, coef_out, status_out, iters_out, n_threads);
}

int osqp_port_solve_batch_rows(int r, int n_traj, const int* seg_offsets, const double* waypoints, const double* times,
                               const double* bc, const double* corr_lo, const double* corr_hi, int K, const double* row_tau,
                               const int* row_deriv, const double* row_lo, const double* row_hi, const port_settings* st,
                               double* coef_out, int* status_out, int* iters_out, int n_threads);

int osqp_port_solve_batch_corridor(int r, int n_traj, const int* seg_offsets, const double* waypoints, const double* times,
                                   const double* bc, const double* corr_lo, const double* corr_hi, const port_settings* st,
                                   double* coef_out, int* status_out, int* iters_out, int n_threads) {
    return osqp_port_solve_batch_rows(r, n_traj, seg_offsets, waypoints, times, bc, corr_lo, corr_hi, 0, NULL, NULL, NULL, NULL, st,
                                      coef_out, status_out, iters_out, n_threads);
}

/* rows: layout of uavqp_solve_rows_batch_device (include/uavqp.h): [segments][K] tau (fraction of T), derivative order (< 0 unused),
 * [segments][K][3] bounds per axis */
int osqp_port_solve_batch_rows(int r, int n_traj, const int* seg_offsets, const double* waypoints, const double* times,
                               const double* bc, const double* corr_lo, const double* corr_hi, int K, const double* row_tau,
                               const int* row_deriv, const double* row_lo, const double* row_hi, const port_settings* st,
                               double* coef_out, int* status_out, int* iters_out, int n_threads) {
    if (r != 3 && r != 4) return -2;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_traj) n_threads = n_traj > 0 ? n_traj : 1;
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)n_threads);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        job_t j = { r, (int)((long long)n_traj * t / n_threads), (int)((long long)n_traj * (t + 1) / n_threads),
                    seg_offsets, waypoints, times, bc, st, coef_out, status_out, iters_out, corr_lo, corr_hi,
                    K, row_tau, row_deriv, row_lo, row_hi };
        jobs[t] = j;
    }
    if (n_threads == 1) job_run(&jobs[0]);
    else {
        /* one thread per allowed CPU, pinned (un-pinned passes of the all-cores baseline spread 4x on the 128-core host):
         * thread t runs on the t-th CPU of the process's affinity mask, wrapping around if there are more threads than CPUs */
        cpu_set_t allowed;
        int cpus[1024], n_cpu = 0;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
            for (int c = 0; c < CPU_SETSIZE && n_cpu < 1024; ++c)
                if (CPU_ISSET(c, &allowed)) cpus[n_cpu++] = c;
        for (int t = 0; t < n_threads; ++t) {
            pthread_create(&th[t], NULL, job_run, &jobs[t]);
            if (n_cpu > 0) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpus[t % n_cpu], &one);
                (void)pthread_setaffinity_np(th[t], sizeof(one), &one);
            }
        }
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    free(jobs); free(th);
    return 0;
}
