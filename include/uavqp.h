/*
 * uavqp.h -- C ABI of the MI355X-native batched minimum-jerk / minimum-snap trajectory QP back-end.
 *
 * Drop-in boundary for ONE hot path of peiyu-cui/uav_motion_planning: the polynomial-trajectory QP
 * of traj_optimization::MinimumControl (reference: src/planner/traj_optimization/include/
 * traj_optimization/minimum_control.h:10-48, src/minimum_control.cpp:5-202).  The reference has no
 * FFI; its interface for this path is the C++ class, so the bindings a maintainer adds are the two
 * C++ facades in uav_motion_planning_amd/cpp/ (MinimumControl, TrajOptimizer) -- see INTEGRATION.md.
 *
 * Plain pointers and sizes only: no Eigen, STL, torch or HIP types cross this boundary.
 * All floating-point data is IEEE float64 (the reference's c_float/VectorXd type); indices int32.
 *
 * -------------------------------------------------------------------------------------------------
 * Problem solved per trajectory and per axis (r = 3 min-jerk: the reference; r = 4 min-snap):
 *   minimise   sum_i  integral_0^{T_i} (d^r p_i/dt^r)^2 dt          [getHessian,  minimum_control.cpp:5-19]
 *   subject to p_0^(d)(0)   = start derivative d, d = 0..r-1        [rows 0..r-1,          :29-31, :101-107]
 *              p_i(T_i)     = waypoint i+1                          [waypoint rows,        :34-42, :118-124]
 *              p_i^(d)(T_i) = p_{i+1}^(d)(0), d = 0..r-1            [continuity rows,      :45-74]
 *              p_{M-1}^(d)(T) = end derivative d                    [end rows,             :77-95, :109-115]
 * Every row is an equality (lb == ub in the reference), so the QP has a unique minimiser; the device
 * path computes that minimiser directly (reduced SPD block-tridiagonal system)
 * instead of iterating ADMM to OSQP's eps = 1e-3.
 *
 * Data layout (one batch = n_traj independent trajectories, 3 axes each):
 *   seg_offsets [n_traj+1] int32   CSR offsets into the segment arrays; trajectory b has
 *                                  M_b = seg_offsets[b+1]-seg_offsets[b] segments and M_b+1 waypoints.
 *                                  May be NULL when uniform_segments > 0 (then M_b = uniform_segments).
 *   waypoints   [sum_b (M_b+1)][3] xyz interleaved, trajectory b starts at row seg_offsets[b] + b
 *                                  (= what A* / RRT* hand over: std::vector<Eigen::Vector3d>,
 *                                  test_minimum_jerk.cpp:41-57).
 *   times       [sum_b M_b]        segment durations (time_vec, test_minimum_jerk.cpp:65-71), > 0.
 *   bc          [n_traj][2][r-1][3] boundary derivatives: [start|end][vel,acc(,jerk)][xyz]
 *                                  (bound_vel / bound_acc of MinimumControl::solve, minimum_control.h:32-35).
 *   coeff_out   [sum_b 3*M_b*2r]   trajectory b at 3*2r*seg_offsets[b], layout [axis][segment][2r]:
 *                                  each [axis] slice IS the reference's coef_1d_ vector for that axis --
 *                                  coef[2r*i + k] multiplies t^k (ascending powers) in segment-local time
 *                                  (minimum_control.cpp:186, consumer poly_traj_server.cpp:68-78).
 *   status_out  [n_traj] int32     UAVQP_SOLVED or a negative per-trajectory code (may be NULL).
 */
#ifndef UAVQP_H_
#define UAVQP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uavqp_ctx uavqp_ctx;

/* return codes of every entry point (never throws) */
enum {
    UAVQP_OK = 0,
    UAVQP_ERR_INVALID_ARG = -1,
    UAVQP_ERR_HIP = -2,       /* HIP runtime error; uavqp_last_error() has the text */
    UAVQP_ERR_NO_DEVICE = -3, /* no usable gfx950 device: the product path never falls back to CPU */
    UAVQP_ERR_ALLOC = -4,
    UAVQP_ERR_RCCL = -5       /* RCCL missing or a collective failed; uavqp_last_error() has the text */
};

/* per-trajectory status (positive = OSQP's OSQP_SOLVED value, which the reference's solve() maps to true) */
enum {
    UAVQP_SOLVED = 1,
    UAVQP_MAX_ITER_REACHED = -2, /* inequality-constrained solves only (OSQP's OSQP_MAX_ITER_REACHED value): iteration cap reached, or a
                                  * working set went singular and could neither be repaired nor certified infeasible (undecided) */
    UAVQP_PRIMAL_INFEASIBLE = -3, /* general-rows solve only (OSQP's OSQP_PRIMAL_INFEASIBLE value): the rows have no common point, proven
                                   * by a Farkas certificate that passes OSQP's test at uavqp_settings.eps_prim_inf */
    UAVQP_INVALID_INPUT = -10, /* M < 1, M > max_segments, T <= 0 or non-finite input */
    UAVQP_NON_FINITE = -11     /* solution overflowed / NaN (pathological time allocation) */
};

const char* uavqp_version(void);
const char* uavqp_last_error(void);

/* Replaces construction of traj_optimization::MinimumControl + OsqpEigen::Solver
 * (minimum_control.h:14,44).  One ctx per host thread / device; owns a HIP stream and workspaces.  The owned stream is
 * a blocking stream, i.e. ordered with the legacy default (NULL) stream: buffers filled or read there need no extra
 * synchronisation.  A stream passed through uavqp_set_stream is used as it is. */
int uavqp_create(uavqp_ctx** out_ctx, int device);
int uavqp_destroy(uavqp_ctx* ctx);

/* Run all subsequent device work of this ctx on an existing hipStream_t (e.g. the caller's compute
 * stream).  NULL restores the ctx-owned stream.  The ctx's sweep workspaces are shared by its calls and
 * ordered by the stream: work still in flight on the previous stream must be complete before switching
 * (uavqp_synchronize), and concurrent streams need one ctx each. */
int uavqp_set_stream(uavqp_ctx* ctx, void* hip_stream);
int uavqp_synchronize(uavqp_ctx* ctx);

/* Kernel variant selection: 0 = auto, 1 = generic lane-per-trajectory kernel, 2 = register-resident
 * specialised kernel (uniform batches only, tile shape chosen by batch size); 4 / 8 / 16 / 32 = specialised
 * kernel with that many trajectories per wave (4: two lane pairs per axis, the smallest batches; 8: one lane
 * pair per axis, latency shape; 16 / 32: one lane pair per trajectory).
 * For benchmarking/tests; results agree to rounding. */
int uavqp_set_variant(uavqp_ctx* ctx, int variant);

/* Solver settings: plain C struct, replaces the OsqpEigen settings calls of the reference
 * (minimum_control.cpp:160-162: setWarmStart(true), setPrimalInfeasibilityTollerance(1e-3), setMaxIteration(1000))
 * and every tuning knob of this back-end.  uavqp_default_settings fills the defaults (= the reference's three values
 * plus this back-end's measured choices); uavqp_set_settings validates and stores a copy in the ctx, uavqp_get_settings
 * reads it back.  Nothing on the launch path reads the environment: the UAVQP_* variables listed in INTEGRATION.md are
 * read ONCE, by uavqp_create, as overrides of the defaults.
 *   struct_size            sizeof(uavqp_settings) of the caller's build (versioning; must be set)
 *   warm_start             reference: true.  The equality-constrained solve is direct (nothing to warm-start); for the
 *                          inequality-constrained entry points it is the default of their warm_start argument semantics:
 *                          uavqp_solve_corridor_warm_device's explicit argument wins.  Kept for API parity.
 *   eps_prim_inf           reference: 1e-3 (OSQP's primal-infeasibility tolerance, the one tolerance the reference sets).  OSQP calls a problem
 *                          primal infeasible when its dual increment dy satisfies |A' dy|_inf <= eps |dy|_inf and
 *                          u' max(dy, 0) + l' min(dy, 0) <= -eps |dy|_inf.  The general-rows solve applies exactly that test to the Farkas
 *                          certificate its dual active-set method ends with when a violated row depends on the working set and no
 *                          multiplier blocks (dy = the dependency's coefficients, value = minus the row's violation): accepted ->
 *                          UAVQP_PRIMAL_INFEASIBLE; a certificate below the margin -> UAVQP_MAX_ITER_REACHED (undecided, as OSQP would run
 *                          into its iteration cap).  Smaller eps: every provable infeasibility is reported.  (lo > hi on one row or box is
 *                          rejected up front, UAVQP_INVALID_INPUT, as OSQP's data validation does.)  Knot boxes alone are always feasible.
 *   max_iter               reference: 1000 (ADMM iterations).  Here: cap on active-set iterations of the
 *                          inequality-constrained solves; <= 0 = automatic (8 * max_segments + 20).  Hitting it yields
 *                          UAVQP_MAX_ITER_REACHED with a feasible, smooth trajectory (as OSQP's status of the same name).
 *   kernel_variant         as uavqp_set_variant (0 auto)
 *   ragged_window_sort     1: ragged batches >= 2048 are dealt to lanes by segment count inside windows (default; since round 6 the waves of the
 *                          lane-pair kernel rank their window themselves, up to 63 segments), 2: the same dealing from its own launch in front of
 *                          the solve (rounds 2-5; results identical), 0: lane order
 *   generic_lanes_per_traj 0 auto, 1 = one lane per trajectory, 2 = a lane pair per trajectory (two-sided elimination),
 *                          3 = one lane per (trajectory, axis)
 *   generic_waves_per_cu   0 auto, > 0: resident waves per CU of the generic kernel
 *   corridor_pdas_rounds   block-pivoting rounds before the single-pivot active-set phase of a COLD corridor solve (default 3; not
 *                          used with corridor_initial_guess = 2, whose starting set only has to be verified)
 *   corridor_pdas_rounds_warm  the same for a warm-started one (default 0: measured on config 5's outer loop, block rounds from
 *                          the carried working set only disturb it -- largest iteration counts 57 / 54 / 37 / 27 with 3 rounds, 41 / 43 /
 *                          22 / 12 with none and the previous positions as the starting point)
 *   corridor_initial_guess working set a COLD corridor solve starts from.  2 (default): the set a dual active-set method in position space ends
 *                          with (qp_corridor_dual.h, DESIGN.md section 5.13: inverse Hessian of the knot positions once per trajectory,
 *                          Goldfarb-Idnani on a swept tableau) -- the exact block solve then verifies it, one solve per problem instead of
 *                          ~12; trajectories of up to 33 segments, longer batches fall back to 1.  1: the knots whose boxes the end-state
 *                          polynomial misses (closed form, section 5.4).  0: the empty set.  Same result to the last bit whichever is used, for every
 *                          problem that ends UAVQP_SOLVED (one that runs into max_iter hands back the iterate it stopped at, which depends on the start).
 *   rows_lanes_per_problem uavqp_solve_rows_batch_*: 0 auto / 2 = a lane pair per (trajectory, axis) problem with the sweep state in LDS
 *                          (default), 1 = one lane per problem, state in an HBM workspace (the round-2 kernel, kept for A/B).  Same result.
 *   corridor_tail_shape    1 (default): small batches of long r = 4 corridor problems run two waves per CU with twice the sweep state on
 *                          chip (shorter iterations: such a solve is as slow as its slowest problem); 0: always four waves per CU.  Same result.
 *   corridor_prelude_lanes lanes per trajectory of the dual prelude of a cold corridor solve (corridor_initial_guess = 2): 0 (default) / 8 = groups of
 *                          8 or 16 lanes (qp_corridor_dual.h); 1 = ONE lane per trajectory with its tableau in lane-private LDS, single precision
 *                          (qp_corridor_lane.h; batches of at most 16 segments per trajectory, else the groups) -- built in round 5 to get rid of
 *                          the replicated chain and the lockstep of eight, measured at parity on config 3 (330 vs 338 us: a per-lane pivot costs
 *                          ~2.5 k instructions of selects per trip), kept as the independent cross-check of the prelude.  Same result.
 *   cloud_window           1 (default): uavqp_corridor_from_cloud_device sorts rows and points along the cloud's longest axis and scans,
 *                          per block of neighbouring rows, only the points that can still change a box (large clouds, no clearance
 *                          output); rows whose distance to the cloud's bounding box already exceeds the cap of the clearance are
 *                          not scanned at all (an exact bound: Cauchy-Schwarz on the robot ellipsoid's metric); 0: always the
 *                          exhaustive scan.  2 / 3: the two experiments of round 5 on a 2-D grid over the cloud (rings of cells
 *                          around a row; a near pass + a far pass over rows bucketed by the radius they still need) -- both measured
 *                          slower than 1 on BASELINE config 5 (1.23 / 1.25 ms against 0.61 ms per call), kept as cross-checks.
 *                          Identical boxes with every value.
 *   realloc_dead_band      uavqp_time_reallocate_device stretches only when the limit ratio exceeds this (default 1.01)
 *   realloc_overshoot      ... and then by overshoot * ratio (default 1.02) */
typedef struct uavqp_settings {
    int32_t struct_size;
    int32_t warm_start;
    double eps_prim_inf;
    int32_t max_iter;
    int32_t kernel_variant;
    int32_t ragged_window_sort;
    int32_t generic_lanes_per_traj;
    int32_t generic_waves_per_cu;
    int32_t corridor_pdas_rounds;
    int32_t corridor_initial_guess;
    int32_t rows_lanes_per_problem;
    int32_t corridor_pdas_rounds_warm;
    int32_t cloud_window;
    int32_t corridor_tail_shape;
    int32_t corridor_prelude_lanes;   /* (sits in what was padding: sizeof(uavqp_settings) is unchanged) */
    double realloc_dead_band;
    double realloc_overshoot;
} uavqp_settings;
void uavqp_default_settings(uavqp_settings* out);
int uavqp_set_settings(uavqp_ctx* ctx, const uavqp_settings* settings);
int uavqp_get_settings(const uavqp_ctx* ctx, uavqp_settings* out);

/* Batched solve, DEVICE pointers, asynchronous on the ctx stream.
 * Replaces, for a whole batch and 3 axes at once, MinimumControl::solve + getCoef1d
 * (minimum_control.cpp:127-192, :199-202).
 *   r                 3 (min-jerk, reference) or 4 (min-snap)
 *   uniform_segments  > 0: every trajectory has that many segments (seg_offsets may be NULL)
 *                     0  : ragged batch, seg_offsets required
 *   max_segments      upper bound on M_b (ragged); ignored for uniform batches                    */
int uavqp_solve_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                             const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                             const double* d_bc, double* d_coeff_out, int32_t* d_status_out);

/* Same with HOST pointers: H2D copy, solve, D2H copy, synchronous.  total_segments = sum_b M_b.
 * The coefficients of a trajectory flagged UAVQP_INVALID_INPUT come back as zeros (the device entry leaves them
 * untouched; the host entries clear their staging buffer first -- both host entries behave the same); a
 * UAVQP_NON_FINITE trajectory carries the non-finite values it overflowed to. */
int uavqp_solve_batch_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                           const int32_t* seg_offsets, const double* waypoints, const double* times,
                           const double* bc, double* coeff_out, int32_t* status_out);

/* Single-axis entry with the exact argument meaning of
 *   bool MinimumControl::solve(VectorXd& pos_1d, Vector2d& bound_vel, Vector2d& bound_acc, VectorXd& time_vec)
 * (minimum_control.h:32-35): pos_1d[n_seg+1], bound_*[2] = (start, end), time_vec[n_seg]; writes
 * coef_1d[2r*n_seg].  bound_jerk is read only when r == 4 (NULL = zeros).  Host pointers, synchronous.
 * Returns UAVQP_OK and *status_out = UAVQP_SOLVED on success. */
int uavqp_solve_axis_host(uavqp_ctx* ctx, int r, int n_seg, const double* pos_1d, const double* bound_vel,
                          const double* bound_acc, const double* bound_jerk, const double* time_vec,
                          double* coef_1d, int32_t* status_out);

/* Corridor-constrained batched solve (north-star extension, BASELINE configs 3 and 5; no reference
 * counterpart: every row of the reference QP is an equality, minimum_control.cpp:98-125,146-147).
 * The interior-waypoint rows p_i(T_i) = w_{i+1} (minimum_control.cpp:34-42,118-124) become
 * corr_lo <= p_i(T_i) <= corr_hi per axis; everything else is unchanged.
 *   d_corr_lo / d_corr_hi  [sum_b (M_b+1)][3], same indexing as waypoints; the entries of the first and last
 *                          waypoint of a trajectory are ignored (start/end positions stay equalities);
 *                          lo == hi pins that waypoint (so lo = hi = waypoints reproduces uavqp_solve_batch_device).
 *   d_waypoints            start/end positions, and the initial guess (clipped into the box) elsewhere.
 *   d_iters_out            [n_traj] active-set iterations (max over axes), may be NULL.
 * Asynchronous for uniform batches; for a RAGGED batch the device entry points read the last CSR offset back first (4 bytes, one
 * stream synchronisation: workspaces are sized by the batch's total segment count) -- the host entries and the pipeline, which know
 * the total, do not.
 * Exact primal active-set solve in the Hermite variables; status UAVQP_MAX_ITER_REACHED
 * (uavqp_settings.max_iter, default 8 M + 20 iterations) leaves a feasible, smooth, possibly sub-optimal trajectory. */
int uavqp_solve_corridor_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                      const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                                      const double* d_bc, const double* d_corr_lo, const double* d_corr_hi,
                                      double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out);
int uavqp_solve_corridor_batch_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                    const int32_t* seg_offsets, const double* waypoints, const double* times,
                                    const double* bc, const double* corr_lo, const double* corr_hi,
                                    double* coeff_out, int32_t* status_out, int32_t* iters_out);

/* Warm-started corridor solve for outer loops (BASELINE config 5: the corridor QP is re-solved after every time
 * re-allocation, and its working set barely moves).  Same problem, same result as uavqp_solve_corridor_batch_device;
 * d_active_set [n_traj][3][2] uint64 holds per (trajectory, axis) the working set of the active-set method: word 0
 * bit k = the box of interior waypoint k (1 <= k <= M-1 <= 63) is active, word 1 bit k = at its upper bound.
 *   warm_start == 0: the buffer is only written (working set at the solution);
 *   warm_start != 0: it is read as the initial working set (any bit pattern is a valid guess: wrong guesses cost
 *                    iterations, never correctness) and overwritten with the final one.
 *   warm_start == 2: in addition d_coeff_out is READ first: it holds the polynomials of the previous solve of the same batch
 *                    (same boxes, e.g. other durations), and the feasible starting point of the active-set method takes its free
 *                    positions from their knot positions (clipped into the boxes -- any content is admissible) instead of the
 *                    waypoints: a start close to the new minimiser is blocked by few bounds on its way there.
 * A trajectory that ends UAVQP_MAX_ITER_REACHED writes an empty set.  No reference counterpart. */
int uavqp_solve_corridor_warm_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                     const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                                     const double* d_bc, const double* d_corr_lo, const double* d_corr_hi,
                                     double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out,
                                     uint64_t* d_active_set, int warm_start);

/* GENERAL inequality rows (north-star extension: the reference hands any l <= A x <= u to OSQP, minimum_control.cpp:146-147,
 * 164-180, but only ever builds equality rows, :98-125).  On top of the knot boxes of uavqp_solve_corridor_batch_device every
 * segment carries up to rows_per_segment (1 or 2) rows
 *       row_lo <= p_i^(d)(tau T_i) <= row_hi      per axis,   d = 0 position, 1 velocity, 2 acceleration (, 3 jerk for r = 4)
 * i.e. position samples at in-segment times (BASELINE config 3's "K = 2 mid-segment samples"), per-axis velocity / acceleration
 * limits, or -- through the monomial basis -- any row a' c_i on the coefficients of one segment.  This is the ORIGINAL spline
 * space (adapters.refine_with_mid_knots inserts knots instead: cheaper -- it runs on the corridor solver -- but a larger space).
 *   d_corr_lo / d_corr_hi  knot boxes as in uavqp_solve_corridor_batch_device, or both NULL: the reference's waypoint equalities
 *   d_row_tau    [sum_b M_b][rows_per_segment]     position of the row in its segment as a fraction of T_i, 0 <= tau < 1
 *                                                  (tau = 0 with d >= 1 is a limit on a knot derivative; tau = 0, d = 0 is the knot box: rejected)
 *   d_row_deriv  [sum_b M_b][rows_per_segment]     int32 derivative order d, < 0: slot unused
 *   d_row_lo / d_row_hi [sum_b M_b][rows_per_segment][3]   bounds per axis (lo == hi: an equality row; +-1e300 for a one-sided row)
 *   d_active_out [n_traj][3][2 + 2 rows_per_segment] uint64 (may be NULL): working set at the solution -- word 0 / 1: knot boxes
 *                active / at the upper bound (bit k = interior waypoint k), then per row slot j words 2+2j / 3+2j (bit i = segment i)
 * Exact dual active-set solve (Goldfarb-Idnani on the block-tridiagonal KKT system with the rows' multipliers riding in the knot
 * blocks): no feasible starting point is needed, the result is the QP's minimiser to rounding.  A row that enters and turns
 * out to depend on the working set (the KKT system goes singular: rows that no free unknown can move, duplicated or contradictory
 * rows, a degenerate vertex) takes the method's zero-primal-step route: the multipliers move along the dependency until one of the
 * working set reaches zero -- that constraint leaves, the solve goes on (degenerate but feasible problems are SOLVED) -- or none does:
 * a Farkas certificate, status UAVQP_PRIMAL_INFEASIBLE if it passes OSQP's test at uavqp_settings.eps_prim_inf (the reference's 1e-3,
 * minimum_control.cpp:161), else UAVQP_MAX_ITER_REACHED.  UAVQP_MAX_ITER_REACHED also ends a problem at uavqp_settings.max_iter (default
 * 12 M (1 + rows_per_segment) + 30) and one whose working set is regular but too ill-conditioned for the block solve (undecided: the direction
 * solve that would yield the certificate does not vanish, |diag(H) z| > max(eps_prim_inf, 1e-6) |dy| -- the one class of problems the OSQP port proves
 * infeasible and this back-end leaves undecided: 2 of 1246 infeasible soak draws at the reference's eps_prim_inf = 1e-3, 14 at 1e-9, profiles/r06_soak.txt;
 * a feasible problem is never called infeasible).  Both
 * statuses hand back the minimiser of the last regular working set, which need NOT satisfy the remaining rows.  A single-segment
 * trajectory has no free unknown at all: its rows are only checked (violated -> UAVQP_PRIMAL_INFEASIBLE).  M <= 63.
 * Asynchronous for uniform batches; a ragged batch costs one 4-byte read-back + stream synchronisation (see uavqp_solve_corridor_batch_device). */
int uavqp_solve_rows_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                  const int32_t* d_seg_offsets, const double* d_waypoints, const double* d_times,
                                  const double* d_bc, const double* d_corr_lo, const double* d_corr_hi, int rows_per_segment,
                                  const double* d_row_tau, const int32_t* d_row_deriv, const double* d_row_lo,
                                  const double* d_row_hi, double* d_coeff_out, int32_t* d_status_out, int32_t* d_iters_out,
                                  uint64_t* d_active_out);
/* The same solve from HOST pointers (staged through device memory, synchronous; a trajectory flagged UAVQP_INVALID_INPUT comes back
 * as zeros, a UAVQP_MAX_ITER_REACHED / UAVQP_PRIMAL_INFEASIBLE one carries the minimiser of its last regular working set -- which need NOT
 * satisfy the remaining rows: check the status, not the coefficients): what a caller of the reference's solver interface has -- it hands its rows to OSQP from host memory,
 * minimum_control.cpp:164-170. */
int uavqp_solve_rows_batch_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments,
                                const int32_t* seg_offsets, const double* waypoints, const double* times, const double* bc,
                                const double* corr_lo, const double* corr_hi, int rows_per_segment, const double* row_tau,
                                const int32_t* row_deriv, const double* row_lo, const double* row_hi, double* coeff_out,
                                int32_t* status_out, int32_t* iters_out);

/* Time re-allocation step of the outer loop of BASELINE config 5 (north-star extension; the reference uses a
 * constant 1.0 s per segment, test_minimum_jerk.cpp:65-71, and has no such loop -- nothing to mirror, parity is
 * per inner solve).  The 3-axis speed |v| and acceleration |a| of the solved polynomials are sampled at
 * samples_per_seg + 1 uniform points per segment; with rho = max(|v|_peak / v_max, sqrt(|a|_peak / a_max)) over the
 * whole trajectory, if rho > 1.01 EVERY duration of that trajectory is scaled by min(max_stretch, 1.02 rho)
 * (dead band 1.01 and overshoot 1.02 are uavqp_settings.realloc_dead_band / realloc_overshoot)
 * (never shrunk).  Scaling is per trajectory, not per segment: stretching one segment next to short ones makes
 * it overshoot more (it inherits their knot acceleration) and diverges; uniform scaling T -> sT lowers speeds
 * ~1/s and accelerations ~1/s^2.
 * d_times is updated in place; d_changed_out[b] (may be NULL) receives the number of stretched segments of
 * trajectory b, so the caller can stop when it is all zero.  Typical loop: solve, reallocate, solve, ... (<= 5x). */
int uavqp_time_reallocate_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                 double* d_times, const double* d_coeff, double v_max, double a_max,
                                 int samples_per_seg, double max_stretch, int32_t* d_changed_out);

/* Batched evaluation of solved trajectories on a uniform time grid.
 * Replaces, for a whole batch, PolyTraj::evaluatePos / evaluateVel / evaluateAcc
 * (src/planner/traj_utils/include/traj_utils/poly_traj.hpp:74-168) as driven by poly_traj_server's
 * 100 Hz timer (traj_server/src/poly_traj_server.cpp:33-37) and PolyTraj::getTraj (poly_traj.hpp:175-187),
 * and the caller-side sampling loop of test_minimum_jerk.cpp:79-92.
 *   t_s = t0 + s*dt, s = 0..n_samples-1, is trajectory-global time; the segment is found with the
 *   reference's rule (walk while t > T_idx + 1e-4, subtracting; past the end clamp to the last segment's
 *   end, poly_traj.hpp:77-88).
 *   what: bit 0 position, bit 1 velocity, bit 2 acceleration; K = popcount(what) outputs per sample.
 *   d_out [n_traj][n_samples][K][3] float64 (pos, vel, acc order; xyz interleaved = Eigen::Vector3d).
 * d_coeff / d_times / d_seg_offsets: the arrays of uavqp_solve_batch_device.  Asynchronous on the ctx stream. */
int uavqp_eval_batch_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                            const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                            int what, double* d_out);

/* Batched PolyTraj::getTraj + getLength + getMeanVel (src/planner/traj_utils/include/traj_utils/poly_traj.hpp:175-207), the two
 * evaluator members without a per-sample output: positions sampled at t = 0, dt, 2 dt, ... while t < total time (the reference's
 * dt is the constant 0.01 and its t is ACCUMULATED in floating point: the sample count follows that accumulation exactly, so a
 * total time that is a multiple of dt -- the reference's own 1.0 s per segment -- gives the reference's count), length = sum of the
 * chord lengths between consecutive samples, mean velocity = length / total time.
 *   d_length / d_mean_vel [n_traj] float64, d_n_samples [n_traj] int32: any of them may be NULL.  Asynchronous on the ctx stream. */
int uavqp_traj_length_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                             const double* d_times, const double* d_coeff, double dt, double* d_length, double* d_mean_vel,
                             int32_t* d_n_samples);

/* Batched SE(3) ellipsoid collision check of solved trajectories against an obstacle point cloud
 * (SURVEY.md section 8-f, N4).  Replaces, for every sample of every trajectory, KinoAstar::isCollisionFree(pt, acc)
 * (src/planner/path_searching/src/kino_astar.cpp:721-758): body axis b3 = normalize(acc + 9.81 z),
 * b2 = normalize(b3 x (1,0,0)), b1 = normalize(b2 x b3), E = Rot diag(robot_r, robot_r, robot_h) Rot', and a
 * sample collides when some obstacle point o within robot_r + 0.1 of it has |E^-1 (o - p)| <= 1.  The kd-tree
 * radius search of the reference is an exhaustive scan here.  Candidate set: |o - p|^2 <= (robot_r + 0.1)^2 in float64; the
 * reference's PCL search works on float32 points with a float radius (kino_astar.cpp:747-748), so a point within float rounding
 * of the search sphere may be a candidate in one and not in the other.  The VERDICT is unaffected: the ellipsoid (semi-axes
 * robot_r, robot_r, robot_h <= robot_r) lies strictly inside that sphere, so such a point never passes |E^-1 (o - p)| <= 1.
 *   samples: t_s = t0 + s*dt, position and acceleration from the polynomials (segment rule of uavqp_eval_batch_device)
 *   d_obstacles [n_obs][3] float64
 *   d_first_hit [n_traj] int32: index of the first colliding sample, n_samples if the trajectory is collision-free
 *   d_flags     [n_traj][n_samples] uint8 (1 = collides), may be NULL */
int uavqp_ellipsoid_check_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                 const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                 const double* d_obstacles, int n_obs, double robot_r, double robot_h,
                                 int32_t* d_first_hit, uint8_t* d_flags);

/* Uniform grid over an obstacle point cloud and the ellipsoid check that queries it instead of scanning every point:
 * the batched stand-in for the kd-tree radius search of KinoAstar::isCollisionFree
 * (src/planner/path_searching/src/kino_astar.cpp:747-750, kdtree radiusSearch(pt, robot_r + 0.1)).
 *   uavqp_obstacle_grid_build_device: buckets d_obstacles [n_obs][3] by cell (cell_size > 0; robot_r + 0.1 makes a
 *     query visit 27 cells; clouds too large for 2^22 cells get larger cells).  Synchronous (sizes depend on the
 *     cloud's bounds); the grid owns a sorted copy of the points, d_obstacles may be freed afterwards.
 *   uavqp_ellipsoid_check_grid_device: arguments and results of uavqp_ellipsoid_check_device with the grid in place
 *     of the raw cloud -- the same candidate set as uavqp_ellipsoid_check_device and the same arithmetic per candidate, hence
 *     flags identical to it.
 * No reference counterpart as a batch; the per-query semantics are the reference's. */
typedef struct uavqp_grid uavqp_grid;
int uavqp_obstacle_grid_build_device(uavqp_ctx* ctx, const double* d_obstacles, int n_obs, double cell_size, uavqp_grid** out_grid);
int uavqp_obstacle_grid_destroy(uavqp_ctx* ctx, uavqp_grid* grid);
int uavqp_ellipsoid_check_grid_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                      const double* d_times, const double* d_coeff, int n_samples, double t0, double dt,
                                      const uavqp_grid* grid, double robot_r, double robot_h, int32_t* d_first_hit,
                                      uint8_t* d_flags);

/* Corridor boxes from an obstacle point cloud (SURVEY.md section 8-d config 5 "ellipsoid-derived corridor widths",
 * 8-f N4).  No reference counterpart as a function: it turns the reference's SE(3) collision test
 * KinoAstar::isCollisionFree(pt, acc) (src/planner/path_searching/src/kino_astar.cpp:721-758) into the box
 * lo <= p <= hi that uavqp_solve_corridor_batch_device puts in place of the waypoint rows
 * (minimum_control.cpp:34-42,118-124).  For every waypoint row w (n_rows = sum over trajectories of M_b + 1):
 *   attitude: b3 = normalize(acc + 9.81 z), b2 = normalize(b3 x (1,0,0)), b1 = normalize(b2 x b3)
 *             (kino_astar.cpp:724-727), E = Rot diag(robot_r, robot_r, robot_h) Rot' (:729-737), with acc the
 *             acceleration of the solved polynomials d_coeff at that knot, or 0 (hover) when d_coeff is NULL;
 *   clearance g = min over ALL obstacle points o of |E^-1 (o - w)|   (the reference's collision metric, :751-753;
 *             +inf for an empty cloud; g <= 1 means the waypoint itself collides);
 *   half-widths h_i = min(h_max, max(0, g - 1) / (3 |E^-1 e_i|)), i = x, y, z;  lo = w - h, hi = w + h.
 * Guarantee: for any offset d with |d_i| <= h_i and any obstacle o, |E^-1 (o - w - d)| >= g - sum_i |d_i| |E^-1 e_i|
 * >= 1, i.e. the robot ellipsoid (same attitude) translated anywhere in the box contains no obstacle point.
 * A colliding waypoint gets h = 0: lo == hi, the reference's equality row.  First/last rows of a trajectory get
 * lo == hi == w (the corridor solver fixes the end points).
 *   d_times      may be NULL when d_coeff is NULL
 *   d_obstacles  [n_obs][3] float64 (exhaustive scan, like uavqp_ellipsoid_check_device)
 *   d_corr_lo / d_corr_hi  [n_rows][3] float64 out
 *   d_clearance  [n_rows] float64 out (g), may be NULL
 * Asynchronous on the ctx stream. */
int uavqp_corridor_from_cloud_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, const int32_t* d_seg_offsets,
                                     int n_rows, const double* d_waypoints, const double* d_times, const double* d_coeff,
                                     const double* d_obstacles, int n_obs, double robot_r, double robot_h, double h_max,
                                     double* d_corr_lo, double* d_corr_hi, double* d_clearance);

/* BASELINE config 5 as ONE call (north-star extension: "SE(3) ellipsoid-collision corridor + time-reallocation outer loop"; the
 * reference has no such loop -- constant 1.0 s per segment, test_minimum_jerk.cpp:65-71, every row an equality,
 * minimum_control.cpp:98-125 -- so parity is per inner solve, SURVEY.md section 8-a').  Host-side C++ sequencing of the entry points
 * above on device buffers:
 *   uavqp_solve_batch_device (the reference's equality problem) -> uavqp_corridor_from_cloud_device (boxes, attitude of that solve)
 *   -> at most max_rounds x (uavqp_solve_corridor_warm_device -- from the second round on warm_start = 2: working set carried over, previous
 *   polynomials as the starting point -- + uavqp_time_reallocate_device; the loop ends
 *   as soon as a re-allocation stretches nothing; if the cap is reached with durations still changing, one more solve makes the
 *   coefficients match d_times) -> uavqp_ellipsoid_check_grid_device on check_samples samples per trajectory (ONE time grid for the
 *   batch: dt = longest total duration / (check_samples - 1)) -> at most repair_rounds x (the boxes of every colliding trajectory
 *   are halved towards its waypoints -- in the last repair round collapsed onto them: the reference's equality rows --, warm re-solve,
 *   one re-allocation, re-solve if that stretched anything, check again).  A colliding trajectory with an interior waypoint whose box
 *   is degenerate (the cloud leaves no room around the searcher's waypoint) cannot be helped by narrower boxes: counted, not repaired.
 * Loop control is data dependent: the number of trajectories a round stretched reaches the host through a word of pinned,
 * host-coherent memory that the round's last kernel writes and the host polls (no copy, no event, no stream stop between rounds; round
 * 1, and later rounds while they still stretch more than n_traj / 64 trajectories, are enqueued before the previous count is looked
 * at, and so is the extra solve at the cap -- a round that turns out to be unnecessary works on an empty list and changes no byte).  From the second round on, re-allocation and compaction only visit the trajectories the previous
 * round re-solved.  The check needs no host round trip of its own: the longest duration stays on the device (the check kernel forms dt
 * from it), and ONE counter block read after the check carries hits, blocked waypoints, unsolved trajectories and that duration; the
 * call returns with the stream idle (SYNCHRONOUS).  The check itself does not test samples beyond a trajectory's end more than once
 * (they all are its end point and could only repeat the first one's verdict at a larger index).
 *   total_segments      sum_b M_b (the host knows it: it sized the buffers); uniform batches: n_traj * uniform_segments
 *   d_times             [total_segments] IN / OUT: stretched in place by the re-allocation (never shrunk)
 *   grid                uniform grid over d_obstacles with cell = check radius + 0.1 (uavqp_obstacle_grid_build_device), or NULL: built
 *                       and destroyed inside the call (a planner builds it once per map)
 *   d_corr_lo / d_corr_hi [total_segments + n_traj][3] OUT: the boxes of the final solve
 *   d_first_hit         [n_traj] OUT (may be NULL): first colliding sample of the final check, check_samples = collision-free
 *   params              uavqp_default_pipeline_params fills the launch-file values of the reference's kino-A* test
 *                       (test_kino_astar_searching.launch:49-57: robot 0.4 x 0.1 m, 7 m/s, 10 m/s^2); check_samples = 0 skips
 *                       the check and the repair; check_robot_r / _h > 0 check with another ellipsoid than the boxes were built with
 *   result              (may be NULL) rounds run, repair rounds run, trajectories still stretching at the cap, colliding at the first
 *                       check / of those with a blocked waypoint / colliding at the last check, trajectories not UAVQP_SOLVED, the dt
 *                       of the check grid. */
typedef struct uavqp_pipeline_params {
    int32_t struct_size;
    int32_t max_rounds;
    double robot_r, robot_h, h_max;
    double v_max, a_max;
    int32_t samples_per_seg;
    int32_t check_samples;
    double max_stretch;
    int32_t repair_rounds;
    int32_t reserved_;
    double check_robot_r, check_robot_h;
} uavqp_pipeline_params;
typedef struct uavqp_pipeline_result {
    int32_t rounds, repairs, still_stretching, colliding_before_repair, colliding_with_blocked_waypoints, colliding_after, unsolved, reserved_;
    double check_dt;
} uavqp_pipeline_result;
void uavqp_default_pipeline_params(uavqp_pipeline_params* out);
int uavqp_corridor_pipeline_device(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments, int total_segments,
                                   const int32_t* d_seg_offsets, const double* d_waypoints, double* d_times, const double* d_bc,
                                   const double* d_obstacles, int n_obs, const uavqp_grid* grid, const uavqp_pipeline_params* params,
                                   double* d_coeff_out, int32_t* d_status_out, double* d_corr_lo, double* d_corr_hi,
                                   int32_t* d_first_hit, uavqp_pipeline_result* result);
/* The same from HOST pointers (staged through device memory; times is updated in place; corr_lo / corr_hi / first_hit may be NULL):
 * what a C++ planner that keeps its paths in host memory calls -- TrajOptimizer::solvePipeline. */
int uavqp_corridor_pipeline_host(uavqp_ctx* ctx, int r, int n_traj, int uniform_segments, int max_segments, const int32_t* seg_offsets,
                                 const double* waypoints, double* times, const double* bc, const double* obstacles, int n_obs,
                                 const uavqp_pipeline_params* params, double* coeff_out, int32_t* status_out, double* corr_lo,
                                 double* corr_hi, int32_t* first_hit, uavqp_pipeline_result* result);

/* quadrotor_msgs/PolynomialTrajectory packer.  HOST function, no ctx, no device: turns ONE solved
 * trajectory (the [axis][segment][2r] slice of coeff_out that belongs to it) into the arrays of the message the rest of the
 * reference stack consumes -- src/simulator/utils/quadrotor_msgs/msg/PolynomialTrajectory.msg:1-28; the only consumer in the
 * reference is trajCallback, src/planner/traj_server/src/poly_traj_server.cpp:57-81, which reads num_order + 1 coefficients per
 * segment and axis from coef_x / coef_y / coef_z (ascending powers, segment-local time: the layout of coef_1d_,
 * minimum_control.cpp:186) and the segment durations from time[].  Nobody in the reference PRODUCES this message; this closes
 * the loop from the solver to poly_traj_server.
 *   coeff_traj  [3][n_seg][2r]   one trajectory of uavqp_solve_*'s coeff_out
 *   times       [n_seg]
 *   coef_x / coef_y / coef_z [n_seg * 2r], time_out [n_seg], order_out [n_seg] (uint32, = 2r - 1 each): caller-allocated
 *   num_order_out = 2r - 1, num_segment_out = n_seg (the message's scalar fields; trajectory_id, action = ACTION_ADD (1),
 *   start_yaw / final_yaw, mag_coeff, header.stamp are the caller's).
 * Returns UAVQP_ERR_INVALID_ARG for r not in {3, 4}, n_seg < 1, a null pointer, or a non-positive / non-finite duration
 * (poly_traj_server would walk off the end of such a trajectory). */
int uavqp_pack_polynomial_trajectory(int r, int n_seg, const double* coeff_traj, const double* times, double* coef_x, double* coef_y,
                                     double* coef_z, double* time_out, uint32_t* order_out, uint32_t* num_order_out,
                                     uint32_t* num_segment_out);

/* ---- Multi-GPU: contiguous shards of the batch, one process (or thread) and one ctx per GPU ----
 * Trajectories are independent QPs, so the solve itself needs no collective: every rank solves its own slice with the entry
 * points above.  The one exchange step is the all-gather of the solved coefficient shards over xGMI (RCCL).  The reference has
 * no multi-device code (its only remark on parallelism: test_minimum_jerk.cpp:73-74); nothing to mirror.
 *
 *   uavqp_shard_bounds          bounds[g] = floor(n_traj g / world), g = 0..world: rank g owns trajectories [bounds[g], bounds[g+1]).
 *   uavqp_shard_bounds_ragged   same, balanced by total SEGMENT count (work is proportional to M_b): bounds[g] is the first
 *                               trajectory whose segment offset reaches g / world of the total.  Host arithmetic on the host CSR
 *                               offsets; no device, no communicator.
 *   uavqp_comm_unique_id        fills a UAVQP_UNIQUE_ID_BYTES token on ONE rank (RCCL's ncclGetUniqueId); the caller ships it to
 *                               the other ranks by whatever it has (MPI, a file, a socket, torch.distributed's store).
 *   uavqp_comm_create           collective over all ranks: the ctx creates and owns the RCCL communicator (one per ctx; released
 *                               by uavqp_comm_destroy / uavqp_destroy).  world = 1 is valid (single-GPU self test).
 *   uavqp_comm_info             rank and world size of the ctx's communicator as RCCL reports them (ncclCommUserRank, ncclCommCount):
 *                               what bench.py records as allgather.rccl_world -- evidence that the exchange ran on a communicator of N ranks.
 *   uavqp_allgather_coeffs      device buffers, asynchronous on the ctx stream (ordered behind the solve): rank g contributes
 *                               counts[g] doubles from d_local; d_full receives the shards back to back in rank order
 *                               (sum_g counts[g] doubles).  Equal counts: one ncclAllGather; otherwise every rank sends its shard
 *                               to every peer directly (grouped ncclSend / ncclRecv: one hop per peer on the xGMI full mesh, no
 *                               padding).  d_local may alias its own slot of d_full (in place).
 *   uavqp_allgather_status      the same for the int32 status arrays (counts in trajectories).
 * RCCL is bound at run time (librccl.so.1); without it uavqp_comm_* return UAVQP_ERR_RCCL and uavqp_last_error() says why. */
#define UAVQP_UNIQUE_ID_BYTES 128
int uavqp_shard_bounds(int n_traj, int world, int32_t* bounds);
int uavqp_shard_bounds_ragged(const int32_t* seg_offsets, int n_traj, int world, int32_t* bounds);
int uavqp_comm_unique_id(void* id_out);
int uavqp_comm_create(uavqp_ctx* ctx, int rank, int world, const void* unique_id);
int uavqp_comm_info(const uavqp_ctx* ctx, int32_t* rank_out, int32_t* world_out);
int uavqp_comm_destroy(uavqp_ctx* ctx);
int uavqp_allgather_coeffs(uavqp_ctx* ctx, const double* d_local, const int64_t* counts, double* d_full);
int uavqp_allgather_status(uavqp_ctx* ctx, const int32_t* d_local, const int64_t* counts, int32_t* d_full);

/* hipGraph capture of a launch-bound inner loop: everything enqueued on the ctx stream between
 * uavqp_capture_begin and uavqp_capture_end (any number of uavqp_solve_batch_device calls with their
 * workspaces already sized by one eager call) becomes one executable graph; uavqp_graph_launch replays
 * it on the ctx stream.  No reference counterpart (the reference has no device queue). */
int uavqp_capture_begin(uavqp_ctx* ctx);
int uavqp_capture_end(uavqp_ctx* ctx, void** out_graph_exec);
int uavqp_graph_launch(uavqp_ctx* ctx, void* graph_exec);
int uavqp_graph_destroy(uavqp_ctx* ctx, void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* UAVQP_H_ */
