"""ctypes binding of the C ABI in include/uavqp.h (libuavqp.so, built in-tree by csrc/Makefile).

There is no CPU fallback: if the shared library is missing this module raises at import of the
symbol table, and every solve call raises UavqpError when no gfx950 device is usable.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# UAVQP_LIB_PATH: development aid for A/B runs of alternative builds of the same source (tools/); the product loads the in-tree library
LIB_PATH = os.environ.get("UAVQP_LIB_PATH") or os.path.join(_PKG, "libuavqp.so")

UAVQP_OK = 0
UAVQP_ERR_INVALID_ARG = -1
UAVQP_ERR_HIP = -2
UAVQP_ERR_NO_DEVICE = -3
UAVQP_ERR_ALLOC = -4
UAVQP_ERR_RCCL = -5
UAVQP_UNIQUE_ID_BYTES = 128

UAVQP_SOLVED = 1
UAVQP_MAX_ITER_REACHED = -2
UAVQP_PRIMAL_INFEASIBLE = -3
UAVQP_INVALID_INPUT = -10
UAVQP_NON_FINITE = -11

# every symbol include/uavqp.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = (
    "uavqp_version",
    "uavqp_last_error",
    "uavqp_create",
    "uavqp_destroy",
    "uavqp_set_stream",
    "uavqp_synchronize",
    "uavqp_set_variant",
    "uavqp_default_settings",
    "uavqp_set_settings",
    "uavqp_get_settings",
    "uavqp_solve_batch_device",
    "uavqp_solve_batch_host",
    "uavqp_solve_axis_host",
    "uavqp_solve_corridor_batch_device",
    "uavqp_solve_corridor_batch_host",
    "uavqp_solve_corridor_warm_device",
    "uavqp_solve_rows_batch_device",
    "uavqp_solve_rows_batch_host",
    "uavqp_time_reallocate_device",
    "uavqp_eval_batch_device",
    "uavqp_traj_length_device",
    "uavqp_ellipsoid_check_device",
    "uavqp_corridor_from_cloud_device",
    "uavqp_obstacle_grid_build_device",
    "uavqp_obstacle_grid_destroy",
    "uavqp_ellipsoid_check_grid_device",
    "uavqp_default_pipeline_params",
    "uavqp_corridor_pipeline_device",
    "uavqp_corridor_pipeline_host",
    "uavqp_pack_polynomial_trajectory",
    "uavqp_shard_bounds",
    "uavqp_shard_bounds_ragged",
    "uavqp_comm_unique_id",
    "uavqp_comm_create",
    "uavqp_comm_info",
    "uavqp_comm_destroy",
    "uavqp_allgather_coeffs",
    "uavqp_allgather_status",
    "uavqp_capture_begin",
    "uavqp_capture_end",
    "uavqp_graph_launch",
    "uavqp_graph_destroy",
)


class UavqpError(RuntimeError):
    pass


class Settings(ctypes.Structure):
    """uavqp_settings of include/uavqp.h (field order and types must match the header)."""
    _fields_ = [("struct_size", ctypes.c_int32), ("warm_start", ctypes.c_int32), ("eps_prim_inf", ctypes.c_double),
                ("max_iter", ctypes.c_int32), ("kernel_variant", ctypes.c_int32), ("ragged_window_sort", ctypes.c_int32),
                ("generic_lanes_per_traj", ctypes.c_int32), ("generic_waves_per_cu", ctypes.c_int32),
                ("corridor_pdas_rounds", ctypes.c_int32), ("corridor_initial_guess", ctypes.c_int32), ("rows_lanes_per_problem", ctypes.c_int32),
                ("corridor_pdas_rounds_warm", ctypes.c_int32), ("cloud_window", ctypes.c_int32), ("corridor_tail_shape", ctypes.c_int32),
                ("corridor_prelude_lanes", ctypes.c_int32), ("realloc_dead_band", ctypes.c_double),
                ("realloc_overshoot", ctypes.c_double)]


class PipelineParams(ctypes.Structure):
    """uavqp_pipeline_params of include/uavqp.h."""
    _fields_ = [("struct_size", ctypes.c_int32), ("max_rounds", ctypes.c_int32), ("robot_r", ctypes.c_double), ("robot_h", ctypes.c_double),
                ("h_max", ctypes.c_double), ("v_max", ctypes.c_double), ("a_max", ctypes.c_double), ("samples_per_seg", ctypes.c_int32),
                ("check_samples", ctypes.c_int32), ("max_stretch", ctypes.c_double), ("repair_rounds", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("check_robot_r", ctypes.c_double), ("check_robot_h", ctypes.c_double)]


class PipelineResult(ctypes.Structure):
    """uavqp_pipeline_result of include/uavqp.h."""
    _fields_ = [("rounds", ctypes.c_int32), ("repairs", ctypes.c_int32), ("still_stretching", ctypes.c_int32),
                ("colliding_before_repair", ctypes.c_int32), ("colliding_with_blocked_waypoints", ctypes.c_int32),
                ("colliding_after", ctypes.c_int32), ("unsolved", ctypes.c_int32), ("reserved_", ctypes.c_int32), ("check_dt", ctypes.c_double)]


def build(force=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", csrc])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UavqpError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The uavqp product path has no CPU fallback."
        )
    # torch wheels bundle their own libamdhip64 / libhsa-runtime64 (no SONAME).  Loaded first, the dynamic
    # loader resolves libuavqp.so's HIP dependency to that same copy (one HIP runtime per process, torch
    # streams and tensors are directly usable); loaded second, torch brings up a second HSA runtime and
    # fails with "No HIP GPUs are available".  So: torch first, whenever torch is present.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, dp, ip = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p
    L.uavqp_version.restype = ctypes.c_char_p
    L.uavqp_last_error.restype = ctypes.c_char_p
    L.uavqp_create.argtypes = [ctypes.POINTER(vp), i32]
    L.uavqp_destroy.argtypes = [vp]
    L.uavqp_set_stream.argtypes = [vp, vp]
    L.uavqp_synchronize.argtypes = [vp]
    L.uavqp_set_variant.argtypes = [vp, i32]
    L.uavqp_default_settings.argtypes = [ctypes.POINTER(Settings)]
    L.uavqp_default_settings.restype = None
    L.uavqp_set_settings.argtypes = [vp, ctypes.POINTER(Settings)]
    L.uavqp_get_settings.argtypes = [vp, ctypes.POINTER(Settings)]
    L.uavqp_solve_batch_device.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, ip]
    L.uavqp_solve_batch_host.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, ip]
    L.uavqp_solve_axis_host.argtypes = [vp, i32, i32, dp, dp, dp, dp, dp, dp, ctypes.POINTER(ctypes.c_int32)]
    L.uavqp_solve_corridor_batch_device.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, dp, dp, ip, ip]
    L.uavqp_solve_corridor_warm_device.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, dp, dp, ip, ip, vp, i32]
    L.uavqp_solve_corridor_batch_host.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, dp, dp, ip, ip]
    L.uavqp_solve_rows_batch_device.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, dp, i32, dp, ip, dp, dp, dp, ip, ip, vp]
    L.uavqp_solve_rows_batch_host.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, dp, i32, dp, ip, dp, dp, dp, ip, ip]
    L.uavqp_time_reallocate_device.argtypes = [vp, i32, i32, i32, ip, dp, dp, ctypes.c_double, ctypes.c_double, i32, ctypes.c_double, ip]
    L.uavqp_eval_batch_device.argtypes = [vp, i32, i32, i32, ip, dp, dp, i32, ctypes.c_double, ctypes.c_double, i32, dp]
    L.uavqp_traj_length_device.argtypes = [vp, i32, i32, i32, ip, dp, dp, ctypes.c_double, dp, dp, ip]
    L.uavqp_ellipsoid_check_device.argtypes = [vp, i32, i32, i32, ip, dp, dp, i32, ctypes.c_double, ctypes.c_double, dp, i32,
                                               ctypes.c_double, ctypes.c_double, ip, vp]
    L.uavqp_corridor_from_cloud_device.argtypes = [vp, i32, i32, i32, ip, i32, dp, dp, dp, dp, i32, ctypes.c_double, ctypes.c_double,
                                                   ctypes.c_double, dp, dp, dp]
    L.uavqp_obstacle_grid_build_device.argtypes = [vp, dp, i32, ctypes.c_double, ctypes.POINTER(vp)]
    L.uavqp_obstacle_grid_destroy.argtypes = [vp, vp]
    L.uavqp_ellipsoid_check_grid_device.argtypes = [vp, i32, i32, i32, ip, dp, dp, i32, ctypes.c_double, ctypes.c_double, vp,
                                                    ctypes.c_double, ctypes.c_double, ip, vp]
    L.uavqp_default_pipeline_params.argtypes = [ctypes.POINTER(PipelineParams)]
    L.uavqp_default_pipeline_params.restype = None
    L.uavqp_corridor_pipeline_device.argtypes = [vp, i32, i32, i32, i32, i32, ip, dp, dp, dp, dp, i32, vp, ctypes.POINTER(PipelineParams),
                                                 dp, ip, dp, dp, ip, ctypes.POINTER(PipelineResult)]
    L.uavqp_corridor_pipeline_host.argtypes = [vp, i32, i32, i32, i32, ip, dp, dp, dp, dp, i32, ctypes.POINTER(PipelineParams), dp, ip, dp, dp,
                                               ip, ctypes.POINTER(PipelineResult)]
    L.uavqp_pack_polynomial_trajectory.argtypes = [i32, i32, dp, dp, dp, dp, dp, dp, vp, vp, vp]
    L.uavqp_shard_bounds.argtypes = [i32, i32, ip]
    L.uavqp_shard_bounds_ragged.argtypes = [ip, i32, i32, ip]
    L.uavqp_comm_unique_id.argtypes = [vp]
    L.uavqp_comm_create.argtypes = [vp, i32, i32, vp]
    L.uavqp_comm_destroy.argtypes = [vp]
    L.uavqp_comm_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.uavqp_allgather_coeffs.argtypes = [vp, dp, vp, dp]
    L.uavqp_allgather_status.argtypes = [vp, ip, vp, ip]
    L.uavqp_capture_begin.argtypes = [vp]
    L.uavqp_capture_end.argtypes = [vp, ctypes.POINTER(vp)]
    L.uavqp_graph_launch.argtypes = [vp, vp]
    L.uavqp_graph_destroy.argtypes = [vp, vp]
    for name in SYMBOLS:
        getattr(L, name)  # AttributeError here = the library does not match the header
    _lib = L
    return L


def has_experiments():
    """True when libuavqp.so was built with -DUAVQP_EXPERIMENTS (`make -C csrc experiments`): the measured-slower cross-check kernels of
    cloud_grid2d.h (cloud_window = 2 / 3) and qp_corridor_lane.h (corridor_prelude_lanes = 1) exist; the default build refuses those settings."""
    return b"experiments" in lib().uavqp_version()


def check(rc, what):
    if rc != UAVQP_OK:
        msg = lib().uavqp_last_error().decode() if rc in (UAVQP_ERR_HIP, UAVQP_ERR_NO_DEVICE, UAVQP_ERR_RCCL, UAVQP_ERR_INVALID_ARG) else ""
        raise UavqpError(f"{what} failed with code {rc} {msg}")
