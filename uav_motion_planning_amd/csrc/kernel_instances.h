// kernel_instances.h -- which kernel instantiations live in which translation unit (round 6).
//
// libuavqp.so used to be ONE translation unit (uavqp.hip, 80-90 s of hipcc for every one-line change).  The register-heavy solver families
// are now compiled on their own (k_*.hip, `make -j`): each family file defines UAVQP_KERNEL_TU, includes its header and this file, and
// expands ITS list with UAVQP_INST = explicit instantiation definition; the host translation unit (uavqp.hip) expands EVERY list with
// UAVQP_INST = explicit instantiation declaration (`extern template`), so its launch sites -- unchanged hipLaunchKernelGGL calls -- bind
// to the family's kernels at link time and nothing is compiled twice.  An instantiation that is launched but missing from these lists is
// simply instantiated by the host translation unit, as before (slower to build, same library).
// Plain (non-template) kernels are emitted by the host translation unit only (#ifndef UAVQP_KERNEL_TU around their definitions).
#pragma once

#ifdef UAVQP_KERNEL_TU
#define UAVQP_INST template
#else
#define UAVQP_INST extern template
#endif

// ---- qp_twisted.h: solve_twisted_kernel<R, M, TILE, LPT>, the (R, M) pairs of find_twisted() x the four tile shapes
#define UAVQP_TWISTED_SHAPES(R_, M_)                                                      \
    UAVQP_INST __global__ void uavqp::solve_twisted_kernel<R_, M_, 4, 16>(uavqp::BatchArgs); \
    UAVQP_INST __global__ void uavqp::solve_twisted_kernel<R_, M_, 8, 8>(uavqp::BatchArgs);  \
    UAVQP_INST __global__ void uavqp::solve_twisted_kernel<R_, M_, 16, 2>(uavqp::BatchArgs); \
    UAVQP_INST __global__ void uavqp::solve_twisted_kernel<R_, M_, 32, 2>(uavqp::BatchArgs);
// (two translation units, k_twisted3.hip / k_twisted4.hip: 80 instantiations are the longest compile of the library)
#define UAVQP_TWISTED_PAIRS4(X) X(4, 2) X(4, 3) X(4, 4) X(4, 5) X(4, 6) X(4, 7) X(4, 8) X(4, 9) X(4, 10) X(4, 12)
#define UAVQP_TWISTED_PAIRS3(X) X(3, 2) X(3, 3) X(3, 4) X(3, 5) X(3, 6) X(3, 7) X(3, 8) X(3, 10) X(3, 12) X(3, 16)
#define UAVQP_INSTANCES_TWISTED3 UAVQP_TWISTED_PAIRS3(UAVQP_TWISTED_SHAPES)
#define UAVQP_INSTANCES_TWISTED4 UAVQP_TWISTED_PAIRS4(UAVQP_TWISTED_SHAPES)

// ---- qp_core_kernels.h / qp_generic2.h: the ragged solvers
#define UAVQP_GENERIC_R(R_)                                                                \
    UAVQP_INST __global__ void uavqp::solve_generic2_kernel<R_, true>(uavqp::BatchArgs);      \
    UAVQP_INST __global__ void uavqp::solve_generic2_kernel<R_, false>(uavqp::BatchArgs);     \
    UAVQP_INST __global__ void uavqp::solve_generic_kernel<R_, true, 3>(uavqp::BatchArgs);    \
    UAVQP_INST __global__ void uavqp::solve_generic_kernel<R_, false, 3>(uavqp::BatchArgs);   \
    UAVQP_INST __global__ void uavqp::solve_generic_kernel<R_, true, 1>(uavqp::BatchArgs);    \
    UAVQP_INST __global__ void uavqp::solve_generic_kernel<R_, false, 1>(uavqp::BatchArgs);
#define UAVQP_INSTANCES_GENERIC UAVQP_GENERIC_R(3) UAVQP_GENERIC_R(4)

// ---- qp_corridor.h: the exact corridor solve
#define UAVQP_INSTANCES_CORRIDOR                                                                                      \
    UAVQP_INST __global__ void uavqp::corridor_solve_kernel<3, true, UAVQP_CORRIDOR_WAVES_PER_CU>(uavqp::CorridorArgs);  \
    UAVQP_INST __global__ void uavqp::corridor_solve_kernel<3, false, UAVQP_CORRIDOR_WAVES_PER_CU>(uavqp::CorridorArgs); \
    UAVQP_INST __global__ void uavqp::corridor_solve_kernel<4, true, UAVQP_CORRIDOR_WAVES_PER_CU>(uavqp::CorridorArgs);  \
    UAVQP_INST __global__ void uavqp::corridor_solve_kernel<4, false, UAVQP_CORRIDOR_WAVES_PER_CU>(uavqp::CorridorArgs); \
    UAVQP_INST __global__ void uavqp::corridor_solve_kernel<4, true, 2>(uavqp::CorridorArgs);                            \
    UAVQP_INST __global__ void uavqp::corridor_solve_kernel<4, false, 2>(uavqp::CorridorArgs);

// ---- qp_corridor_dual.h: the position-space dual preludes
#define UAVQP_CORRIDOR_DUAL_R(R_)                                                                             \
    UAVQP_INST __global__ void uavqp::corridor_dual_kernel<R_, 8, 16>(uavqp::CorridorArgs, int, int, int);       \
    UAVQP_INST __global__ void uavqp::corridor_dual_kernel<R_, 16, 24>(uavqp::CorridorArgs, int, int, int);      \
    UAVQP_INST __global__ void uavqp::corridor_dual_kernel<R_, 16, 32>(uavqp::CorridorArgs, int, int, int);      \
    UAVQP_INST __global__ void uavqp::corridor_dual_mixed_kernel<R_>(uavqp::CorridorArgs, int, int);             \
    UAVQP_INST __global__ void uavqp::corridor_dual_wave_kernel<R_>(uavqp::CorridorArgs, int);                   \
    UAVQP_INST __global__ void uavqp::corridor_dual_wave2_kernel<R_>(uavqp::CorridorArgs, int);
#define UAVQP_INSTANCES_CORRIDOR_DUAL UAVQP_CORRIDOR_DUAL_R(3) UAVQP_CORRIDOR_DUAL_R(4)

// ---- qp_rows.h / qp_rows2.h: the general-rows solvers
#define UAVQP_ROWS_RK(R_, K_)                                                                  \
    UAVQP_INST __global__ void uavqp::rows_solve_kernel<R_, K_>(uavqp::RowsArgs);                 \
    UAVQP_INST __global__ void uavqp::rows_pair_kernel<R_, K_, true, false, false>(uavqp::Rows2Args);    \
    UAVQP_INST __global__ void uavqp::rows_pair_kernel<R_, K_, true, false, true>(uavqp::Rows2Args);     \
    UAVQP_INST __global__ void uavqp::rows_pair_kernel<R_, K_, true, true, false>(uavqp::Rows2Args);     \
    UAVQP_INST __global__ void uavqp::rows_pair_kernel<R_, K_, false, false, false>(uavqp::Rows2Args);   \
    UAVQP_INST __global__ void uavqp::rows_pair_kernel<R_, K_, false, false, true>(uavqp::Rows2Args);    \
    UAVQP_INST __global__ void uavqp::rows_pair_kernel<R_, K_, false, true, false>(uavqp::Rows2Args);
// (one translation unit per (R, K): k_rows31.hip ... k_rows42.hip -- five register-heavy kernels each)
#define UAVQP_INSTANCES_ROWS31 UAVQP_ROWS_RK(3, 1)
#define UAVQP_INSTANCES_ROWS32 UAVQP_ROWS_RK(3, 2)
#define UAVQP_INSTANCES_ROWS41 UAVQP_ROWS_RK(4, 1)
#define UAVQP_INSTANCES_ROWS42 UAVQP_ROWS_RK(4, 2)

// ---- qp_rows_dual.h: the starting set of the rows solve
#define UAVQP_ROWS_DUAL_RK(R_, K_) UAVQP_INST __global__ void uavqp::rows_dual_kernel<R_, K_>(uavqp::RowsDualArgs, int);
#define UAVQP_INSTANCES_ROWS_DUAL UAVQP_ROWS_DUAL_RK(3, 1) UAVQP_ROWS_DUAL_RK(3, 2) UAVQP_ROWS_DUAL_RK(4, 1) UAVQP_ROWS_DUAL_RK(4, 2)

// ---- obstacle_grid.h: boxes from the cloud, grid collision check
#define UAVQP_CLOUD_R(R_)                                                                 \
    UAVQP_INST __global__ void uavqp::cloud_window_kernel<R_, 2>(uavqp::CloudCorridorArgs);  \
    UAVQP_INST __global__ void uavqp::cloud_corridor_kernel<R_>(uavqp::CloudCorridorArgs);   \
    UAVQP_INST __global__ void uavqp::ellipsoid_grid_kernel<R_>(uavqp::EllipsoidGridArgs);
#define UAVQP_INSTANCES_CLOUD UAVQP_CLOUD_R(3) UAVQP_CLOUD_R(4)
