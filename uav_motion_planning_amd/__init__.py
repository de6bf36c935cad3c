"""uav_motion_planning_amd -- MI355X-native batched min-jerk / min-snap trajectory QP back-end.

Hot path of peiyu-cui/uav_motion_planning's traj_optimization::MinimumControl, rebuilt as hand-written
HIP for gfx950 behind a C ABI (include/uavqp.h).  This package is the Python host-side mirror of the
reference interface; the C++ facades live in uav_motion_planning_amd/cpp/.
"""
from ._lib import (UAVQP_INVALID_INPUT, UAVQP_MAX_ITER_REACHED, UAVQP_NON_FINITE, UAVQP_PRIMAL_INFEASIBLE, UAVQP_SOLVED, Settings, UavqpError, build, has_experiments, lib)  # noqa: F401
from .traj_optimizer import Context, MinimumControl, TrajOptimizer  # noqa: F401

__all__ = ["Context", "MinimumControl", "TrajOptimizer", "UavqpError", "Settings", "build", "lib", "has_experiments",
           "UAVQP_SOLVED", "UAVQP_MAX_ITER_REACHED", "UAVQP_PRIMAL_INFEASIBLE", "UAVQP_INVALID_INPUT", "UAVQP_NON_FINITE"]
