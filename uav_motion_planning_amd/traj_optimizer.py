"""Host-side mirror of the reference optimiser interface over the C ABI (include/uavqp.h).

  * MinimumControl  -- same method names / argument meaning / error behaviour as
    traj_optimization::MinimumControl (reference minimum_control.h:10-48, minimum_control.cpp:127-202):
    solve(pos_1d, bound_vel, bound_acc, time_vec) -> bool, getCoef1d(), reset().
  * TrajOptimizer   -- the batch facade named by BASELINE.json's north_star
    (setWaypoints / setTimeAllocation / solve / getPolyCoeff); no reference counterpart.
  * Context         -- thin RAII wrapper of uavqp_ctx for callers that already hold device buffers.

torch is used only for device memory and streams.  No CPU fallback: without libuavqp.so or without a
GPU every solve raises / returns False exactly as the reference does on solver failure.
"""
import ctypes

import numpy as np

from . import _lib


def _ptr(x):
    """Raw address of a numpy array / torch tensor / None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()  # torch tensor


class Context:
    """Owns one uavqp_ctx (one per host thread / device, as MinimumControl owns one OsqpEigen::Solver)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().uavqp_create(ctypes.byref(self._h), int(device)), "uavqp_create")
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().uavqp_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream_handle):
        """hip_stream_handle: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or None."""
        _lib.check(_lib.lib().uavqp_set_stream(self._h, ctypes.c_void_p(hip_stream_handle or 0)), "uavqp_set_stream")

    def set_variant(self, variant):
        _lib.check(_lib.lib().uavqp_set_variant(self._h, int(variant)), "uavqp_set_variant")

    def synchronize(self):
        _lib.check(_lib.lib().uavqp_synchronize(self._h), "uavqp_synchronize")

    def get_settings(self):
        """Current uavqp_settings of this ctx (a _lib.Settings ctypes struct)."""
        st = _lib.Settings()
        _lib.check(_lib.lib().uavqp_get_settings(self._h, ctypes.byref(st)), "uavqp_get_settings")
        return st

    def set_settings(self, **fields):
        """Update fields of the ctx's uavqp_settings (reference: minimum_control.cpp:160-162 sets warm_start,
        eps_prim_inf, max_iter on its OsqpEigen solver), e.g. set_settings(max_iter=1000, ragged_window_sort=0)."""
        st = self.get_settings()
        for k, v in fields.items():
            if not hasattr(st, k):
                raise AttributeError(f"uavqp_settings has no field {k!r}")
            setattr(st, k, v)
        _lib.check(_lib.lib().uavqp_set_settings(self._h, ctypes.byref(st)), "uavqp_set_settings")

    def eval_batch_device(self, r, n_traj, uniform_segments, seg_offsets, times, coeff, n_samples, t0, dt, what, out):
        """Batched PolyTraj::evaluatePos/Vel/Acc on the grid t0 + s*dt (device buffers, asynchronous)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_eval_batch_device(self._h, r, n_traj, uniform_segments, p(seg_offsets), p(times), p(coeff),
                                                n_samples, float(t0), float(dt), int(what), p(out))
        _lib.check(rc, "uavqp_eval_batch_device")

    def traj_length_device(self, r, n_traj, uniform_segments, seg_offsets, times, coeff, dt=0.01, length=None, mean_vel=None, n_samples=None):
        """Batched PolyTraj::getTraj + getLength + getMeanVel (poly_traj.hpp:175-207; dt = the reference's 0.01 s): device buffers."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_traj_length_device(self._h, r, n_traj, uniform_segments, p(seg_offsets), p(times), p(coeff), float(dt),
                                                 p(length), p(mean_vel), p(n_samples))
        _lib.check(rc, "uavqp_traj_length_device")

    def time_reallocate_device(self, r, n_traj, uniform_segments, seg_offsets, times, coeff, v_max, a_max,
                               samples_per_seg=16, max_stretch=1.5, changed=None):
        """One stretch-only time re-allocation step (device buffers, `times` updated in place)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_time_reallocate_device(self._h, r, n_traj, uniform_segments, p(seg_offsets), p(times), p(coeff),
                                                     float(v_max), float(a_max), int(samples_per_seg), float(max_stretch), p(changed))
        _lib.check(rc, "uavqp_time_reallocate_device")

    def ellipsoid_check_device(self, r, n_traj, uniform_segments, seg_offsets, times, coeff, n_samples, t0, dt,
                               obstacles, n_obs, robot_r, robot_h, first_hit, flags=None):
        """Batched KinoAstar::isCollisionFree over the samples of solved trajectories (device buffers)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_ellipsoid_check_device(self._h, r, n_traj, uniform_segments, p(seg_offsets), p(times), p(coeff),
                                                     n_samples, float(t0), float(dt), p(obstacles), int(n_obs),
                                                     float(robot_r), float(robot_h), p(first_hit), p(flags))
        _lib.check(rc, "uavqp_ellipsoid_check_device")

    def solve_corridor_device(self, r, n_traj, uniform_segments, max_segments, seg_offsets, waypoints, times, bc, corr_lo, corr_hi,
                              coeff_out, status_out, iters_out=None, active_set=None, warm_start=False):
        """Corridor-constrained solve on device buffers; active_set ([n_traj,3,2] int64/uint64 device tensor) carries the
        working set between the re-solves of an outer loop (warm_start=True reads it; warm_start=2 also starts the free positions from
        the polynomials found in coeff_out -- include/uavqp.h)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_solve_corridor_warm_device(self._h, r, n_traj, uniform_segments, max_segments, p(seg_offsets), p(waypoints),
                                                         p(times), p(bc), p(corr_lo), p(corr_hi), p(coeff_out), p(status_out),
                                                         p(iters_out), p(active_set), int(warm_start))
        _lib.check(rc, "uavqp_solve_corridor_warm_device")

    def solve_rows_device(self, r, n_traj, uniform_segments, max_segments, seg_offsets, waypoints, times, bc, corr_lo, corr_hi,
                          rows_per_segment, row_tau, row_deriv, row_lo, row_hi, coeff_out, status_out, iters_out=None, active_out=None):
        """Knot boxes (or None: waypoint equalities) + up to rows_per_segment general rows lo <= p_i^(d)(tau T_i) <= hi per segment and
        axis; exact dual active-set solve on device buffers (uavqp_solve_rows_batch_device)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_solve_rows_batch_device(self._h, r, n_traj, uniform_segments, max_segments, p(seg_offsets), p(waypoints),
                                                      p(times), p(bc), p(corr_lo), p(corr_hi), int(rows_per_segment), p(row_tau),
                                                      p(row_deriv), p(row_lo), p(row_hi), p(coeff_out), p(status_out), p(iters_out),
                                                      p(active_out))
        _lib.check(rc, "uavqp_solve_rows_batch_device")

    def corridor_from_cloud_device(self, r, n_traj, uniform_segments, seg_offsets, n_rows, waypoints, times, coeff,
                                   obstacles, n_obs, robot_r, robot_h, h_max, corr_lo, corr_hi, clearance=None):
        """Corridor boxes of every waypoint row from an obstacle cloud, robot ellipsoid of kino_astar.cpp:721-758
        (device buffers; coeff/times None = hover attitude)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_corridor_from_cloud_device(self._h, r, n_traj, uniform_segments, p(seg_offsets), int(n_rows),
                                                         p(waypoints), p(times), p(coeff), p(obstacles), int(n_obs),
                                                         float(robot_r), float(robot_h), float(h_max), p(corr_lo), p(corr_hi),
                                                         p(clearance))
        _lib.check(rc, "uavqp_corridor_from_cloud_device")

    def obstacle_grid_build(self, obstacles, n_obs, cell_size):
        """Uniform grid over a device point cloud; returns an opaque handle (free it with obstacle_grid_destroy)."""
        h = ctypes.c_void_p()
        rc = _lib.lib().uavqp_obstacle_grid_build_device(self._h, obstacles if isinstance(obstacles, int) or obstacles is None else _ptr(obstacles),
                                                         int(n_obs), float(cell_size), ctypes.byref(h))
        _lib.check(rc, "uavqp_obstacle_grid_build_device")
        return h

    def obstacle_grid_destroy(self, grid):
        _lib.check(_lib.lib().uavqp_obstacle_grid_destroy(self._h, grid), "uavqp_obstacle_grid_destroy")

    def ellipsoid_check_grid_device(self, r, n_traj, uniform_segments, seg_offsets, times, coeff, n_samples, t0, dt,
                                    grid, robot_r, robot_h, first_hit, flags=None):
        """ellipsoid_check_device with the candidates taken from an obstacle grid (identical flags, no exhaustive scan)."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_ellipsoid_check_grid_device(self._h, r, n_traj, uniform_segments, p(seg_offsets), p(times), p(coeff),
                                                          n_samples, float(t0), float(dt), grid, float(robot_r), float(robot_h),
                                                          p(first_hit), p(flags))
        _lib.check(rc, "uavqp_ellipsoid_check_grid_device")

    def corridor_pipeline_device(self, r, n_traj, uniform_segments, max_segments, total_segments, seg_offsets, waypoints, times, bc,
                                 obstacles, n_obs, coeff_out, status_out, corr_lo, corr_hi, first_hit=None, grid=None, **params):
        """uavqp_corridor_pipeline_device: BASELINE config 5 as one C-ABI call on device buffers (times is stretched in place).
        params: fields of uavqp_pipeline_params that differ from uavqp_default_pipeline_params.  Returns the uavqp_pipeline_result
        fields as a dict.  Synchronous."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        pp = _lib.PipelineParams()
        _lib.lib().uavqp_default_pipeline_params(ctypes.byref(pp))
        for k, v in params.items():
            if not hasattr(pp, k) or k in ("struct_size", "reserved_"):
                raise ValueError(f"unknown uavqp_pipeline_params field {k!r}")
            setattr(pp, k, v)
        res = _lib.PipelineResult()
        rc = _lib.lib().uavqp_corridor_pipeline_device(self._h, r, n_traj, uniform_segments, max_segments, int(total_segments), p(seg_offsets),
                                                       p(waypoints), p(times), p(bc), p(obstacles), int(n_obs), grid, ctypes.byref(pp),
                                                       p(coeff_out), p(status_out), p(corr_lo), p(corr_hi), p(first_hit), ctypes.byref(res))
        _lib.check(rc, "uavqp_corridor_pipeline_device")
        return {k: getattr(res, k) for k, _ in _lib.PipelineResult._fields_ if k != "reserved_"}

    # ---- multi-GPU: the ctx owns an RCCL communicator (include/uavqp.h, "Multi-GPU") ----
    @staticmethod
    def comm_unique_id():
        """RCCL rendezvous token (bytes, UAVQP_UNIQUE_ID_BYTES): call on ONE rank, ship it to the others."""
        buf = ctypes.create_string_buffer(_lib.UAVQP_UNIQUE_ID_BYTES)
        _lib.check(_lib.lib().uavqp_comm_unique_id(buf), "uavqp_comm_unique_id")
        return buf.raw

    def comm_create(self, rank, world, unique_id):
        """Collective over all ranks: create the communicator this ctx owns (world = 1 is a valid self test)."""
        assert len(unique_id) == _lib.UAVQP_UNIQUE_ID_BYTES
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.UAVQP_UNIQUE_ID_BYTES)
        _lib.check(_lib.lib().uavqp_comm_create(self._h, int(rank), int(world), buf), "uavqp_comm_create")
        self.rank, self.world = int(rank), int(world)

    def comm_info(self):
        """(rank, world) of this ctx's communicator as RCCL itself reports them (ncclCommUserRank / ncclCommCount)."""
        rk, wd = ctypes.c_int32(-1), ctypes.c_int32(-1)
        _lib.check(_lib.lib().uavqp_comm_info(self._h, ctypes.byref(rk), ctypes.byref(wd)), "uavqp_comm_info")
        return int(rk.value), int(wd.value)

    def comm_destroy(self):
        _lib.check(_lib.lib().uavqp_comm_destroy(self._h), "uavqp_comm_destroy")

    def allgather_coeffs(self, local, counts, full):
        """RCCL all-gather of float64 device shards on the ctx stream: rank g contributes counts[g] doubles of `local`,
        `full` receives them back to back in rank order (`local` may be the rank's own slice of `full`)."""
        c = (ctypes.c_int64 * len(counts))(*[int(x) for x in counts])
        _lib.check(_lib.lib().uavqp_allgather_coeffs(self._h, _ptr(local), c, _ptr(full)), "uavqp_allgather_coeffs")

    def allgather_status(self, local, counts, full):
        """The same for the int32 status arrays (counts in trajectories)."""
        c = (ctypes.c_int64 * len(counts))(*[int(x) for x in counts])
        _lib.check(_lib.lib().uavqp_allgather_status(self._h, _ptr(local), c, _ptr(full)), "uavqp_allgather_status")

    def capture_begin(self):
        """Start hipGraph capture of everything subsequently enqueued on the ctx stream."""
        _lib.check(_lib.lib().uavqp_capture_begin(self._h), "uavqp_capture_begin")

    def capture_end(self):
        g = ctypes.c_void_p()
        _lib.check(_lib.lib().uavqp_capture_end(self._h, ctypes.byref(g)), "uavqp_capture_end")
        return g

    def graph_launch(self, graph):
        _lib.check(_lib.lib().uavqp_graph_launch(self._h, graph), "uavqp_graph_launch")

    def graph_destroy(self, graph):
        _lib.check(_lib.lib().uavqp_graph_destroy(self._h, graph), "uavqp_graph_destroy")

    def solve_batch_device(self, r, n_traj, uniform_segments, max_segments, seg_offsets, waypoints, times, bc,
                           coeff_out, status_out=None):
        """All array arguments are device buffers (torch CUDA tensors or raw integer addresses). Asynchronous."""
        def p(x):
            return x if isinstance(x, int) or x is None else _ptr(x)
        rc = _lib.lib().uavqp_solve_batch_device(self._h, r, n_traj, uniform_segments, max_segments, p(seg_offsets),
                                                 p(waypoints), p(times), p(bc), p(coeff_out), p(status_out))
        _lib.check(rc, "uavqp_solve_batch_device")

    def solve_batch_host(self, r, seg_offsets, waypoints, times, bc, uniform_segments=0):
        """numpy in / numpy out (H2D + solve + D2H, synchronous).  Returns (coeff_flat, status)."""
        waypoints = np.ascontiguousarray(waypoints, dtype=np.float64)
        times = np.ascontiguousarray(times, dtype=np.float64)
        bc = np.ascontiguousarray(bc, dtype=np.float64)
        if uniform_segments > 0:
            n_traj = times.size // uniform_segments
            so = None
            total = n_traj * uniform_segments
            mmax = uniform_segments
        else:
            so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
            n_traj = so.size - 1
            total = int(so[-1]) if n_traj > 0 else 0
            mmax = int(np.max(np.diff(so))) if n_traj > 0 else 1
        assert waypoints.size == 3 * (total + n_traj), "waypoints must hold sum(M_b + 1) xyz rows"
        assert bc.size == n_traj * 2 * (r - 1) * 3
        coeff = np.zeros(3 * 2 * r * total, dtype=np.float64)
        status = np.zeros(n_traj, dtype=np.int32)
        rc = _lib.lib().uavqp_solve_batch_host(self._h, r, n_traj, uniform_segments, max(mmax, 1), _ptr(so),
                                               _ptr(waypoints), _ptr(times), _ptr(bc), _ptr(coeff), _ptr(status))
        _lib.check(rc, "uavqp_solve_batch_host")
        return coeff, status

    def solve_corridor_batch_host(self, r, seg_offsets, waypoints, times, bc, corr_lo, corr_hi, uniform_segments=0):
        """Corridor-constrained solve, numpy in / numpy out.  Returns (coeff_flat, status, iters)."""
        waypoints = np.ascontiguousarray(waypoints, dtype=np.float64)
        times = np.ascontiguousarray(times, dtype=np.float64)
        bc = np.ascontiguousarray(bc, dtype=np.float64)
        lo = np.ascontiguousarray(corr_lo, dtype=np.float64)
        hi = np.ascontiguousarray(corr_hi, dtype=np.float64)
        if uniform_segments > 0:
            n_traj = times.size // uniform_segments
            so, total, mmax = None, n_traj * uniform_segments, uniform_segments
        else:
            so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
            n_traj = so.size - 1
            total = int(so[-1]) if n_traj > 0 else 0
            mmax = int(np.max(np.diff(so))) if n_traj > 0 else 1
        assert waypoints.size == 3 * (total + n_traj) == lo.size == hi.size
        coeff = np.zeros(3 * 2 * r * total, dtype=np.float64)
        status = np.zeros(n_traj, dtype=np.int32)
        iters = np.zeros(n_traj, dtype=np.int32)
        rc = _lib.lib().uavqp_solve_corridor_batch_host(self._h, r, n_traj, uniform_segments, max(mmax, 1), _ptr(so),
                                                        _ptr(waypoints), _ptr(times), _ptr(bc), _ptr(lo), _ptr(hi),
                                                        _ptr(coeff), _ptr(status), _ptr(iters))
        _lib.check(rc, "uavqp_solve_corridor_batch_host")
        return coeff, status, iters

    def solve_rows_batch_host(self, r, seg_offsets, waypoints, times, bc, corr_lo, corr_hi, rows_per_segment, row_tau, row_deriv,
                              row_lo, row_hi, uniform_segments=0):
        """Knot boxes (or None, None: waypoint equalities) + general rows, numpy in / numpy out (uavqp_solve_rows_batch_host).
        row_tau / row_deriv [segments][K], row_lo / row_hi [segments][K][3].  Returns (coeff_flat, status, iters)."""
        waypoints = np.ascontiguousarray(waypoints, dtype=np.float64)
        times = np.ascontiguousarray(times, dtype=np.float64)
        bc = np.ascontiguousarray(bc, dtype=np.float64)
        lo = None if corr_lo is None else np.ascontiguousarray(corr_lo, dtype=np.float64)
        hi = None if corr_hi is None else np.ascontiguousarray(corr_hi, dtype=np.float64)
        tau = np.ascontiguousarray(row_tau, dtype=np.float64)
        drv = np.ascontiguousarray(row_deriv, dtype=np.int32)
        rlo = np.ascontiguousarray(row_lo, dtype=np.float64)
        rhi = np.ascontiguousarray(row_hi, dtype=np.float64)
        if uniform_segments > 0:
            n_traj = times.size // uniform_segments
            so, total, mmax = None, n_traj * uniform_segments, uniform_segments
        else:
            so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
            n_traj = so.size - 1
            total = int(so[-1]) if n_traj > 0 else 0
            mmax = int(np.max(np.diff(so))) if n_traj > 0 else 1
        K = int(rows_per_segment)
        assert waypoints.size == 3 * (total + n_traj) and tau.size == total * K == drv.size and rlo.size == 3 * total * K == rhi.size
        coeff = np.zeros(3 * 2 * r * total, dtype=np.float64)
        status = np.zeros(n_traj, dtype=np.int32)
        iters = np.zeros(n_traj, dtype=np.int32)
        rc = _lib.lib().uavqp_solve_rows_batch_host(self._h, r, n_traj, uniform_segments, max(mmax, 1), _ptr(so), _ptr(waypoints), _ptr(times),
                                                    _ptr(bc), _ptr(lo), _ptr(hi), K, _ptr(tau), _ptr(drv), _ptr(rlo), _ptr(rhi),
                                                    _ptr(coeff), _ptr(status), _ptr(iters))
        _lib.check(rc, "uavqp_solve_rows_batch_host")
        return coeff, status, iters

    def solve_axis_host(self, r, pos_1d, bound_vel, bound_acc, time_vec, bound_jerk=None):
        pos = np.ascontiguousarray(pos_1d, dtype=np.float64)
        bv = np.ascontiguousarray(bound_vel, dtype=np.float64)
        ba = np.ascontiguousarray(bound_acc, dtype=np.float64)
        bj = None if bound_jerk is None else np.ascontiguousarray(bound_jerk, dtype=np.float64)
        tv = np.ascontiguousarray(time_vec, dtype=np.float64)
        n_seg = tv.size
        coef = np.zeros(2 * r * max(n_seg, 0), dtype=np.float64)
        st = ctypes.c_int32(0)
        rc = _lib.lib().uavqp_solve_axis_host(self._h, r, n_seg, _ptr(pos), _ptr(bv), _ptr(ba), _ptr(bj), _ptr(tv),
                                              _ptr(coef), ctypes.byref(st))
        return rc, st.value, coef


class MinimumControl:
    """Drop-in mirror of traj_optimization::MinimumControl (reference minimum_control.h:10-48).

    solve() keeps the reference's contract: inputs are not modified, the return value is a bool,
    on failure the previously stored coefficients are kept (minimum_control.cpp:173-184), one axis per
    call, coefficients in ascending powers per segment (minimum_control.cpp:186).  `order` selects
    r = 3 (min-jerk, the reference) or r = 4 (min-snap extension; bound_jerk defaults to zero).
    """

    def __init__(self, order=3, device=0):
        assert order in (3, 4)
        self._r = order
        self._device = device
        self._ctx = None  # created on first solve (the reference builds its OSQP workspace in solve(), too)
        self._coef_1d = np.zeros(0)

    def solve(self, pos_1d, bound_vel, bound_acc, time_vec, bound_jerk=None):
        pos = np.asarray(pos_1d, dtype=np.float64)
        tv = np.asarray(time_vec, dtype=np.float64)
        # reference H8: pos_1d.size() < 2 indexes out of range there; here it is a clean failure
        if pos.ndim != 1 or tv.ndim != 1 or tv.size < 1 or pos.size != tv.size + 1:
            print("solver init failed!")
            return False
        if self._ctx is None:
            self._ctx = Context(self._device)  # raises UavqpError without libuavqp.so / without a GPU
            # the three settings the reference passes to its solver (minimum_control.cpp:160-162)
            self._ctx.set_settings(warm_start=1, eps_prim_inf=1e-3, max_iter=1000)
        rc, st, coef = self._ctx.solve_axis_host(self._r, pos, bound_vel, bound_acc, tv, bound_jerk)
        if rc != _lib.UAVQP_OK:
            print("solver init failed!")
            return False
        if st != _lib.UAVQP_SOLVED:
            print("solver solve failed!")
            return False
        self._coef_1d = coef
        return True

    def reset(self):
        """minimum_control.cpp:194-197: coef_1d_.setZero()."""
        self._coef_1d = np.zeros_like(self._coef_1d)

    def getCoef1d(self):
        """minimum_control.cpp:199-202: returns a copy."""
        return self._coef_1d.copy()


class TrajOptimizer:
    """Batch facade with the north-star method names; all trajectories and axes in one device pass.

    setWaypoints(xyz, wp_offsets)    xyz [sum(M_b+1)][3]; wp_offsets[n_traj+1] (or None + uniform count)
    setTimeAllocation(T)             T [sum M_b]
    setBoundary(bc)                  bc [n_traj][2][r-1][3]; default: all zero (test_minimum_jerk.cpp:59-63)
    setCorridor(lo, hi)              optional boxes [sum(M_b+1)][3] replacing the interior-waypoint equalities
                                     (north-star extension; None, None restores the reference's equality rows)
    setRows(K, tau, deriv, lo, hi)   optional general rows lo <= p^(deriv)(tau T) <= hi per segment, slot and axis (K = 1 or 2 slots;
                                     tau / deriv [sum M_b][K], lo / hi [sum M_b][K][3]; K = 0 removes them) -- the same method as the
                                     C++ facade's (cpp/traj_optimizer.h)
    solve() -> bool                  True iff every trajectory solved (statuses in .status)
    getPolyCoeff()                   flat float64 array, trajectory b at 3*2r*seg_offsets[b], [axis][seg][2r]
    """

    def __init__(self, order=4, device=0):
        assert order in (3, 4)
        self._r = order
        self._device = device
        self._ctx = None
        self._wp = self._T = self._bc = self._so = None
        self._lo = self._hi = None
        self._rows = None
        self._coef = np.zeros(0)
        self.status = np.zeros(0, dtype=np.int32)
        self.iterations = np.zeros(0, dtype=np.int32)

    def setWaypoints(self, xyz, wp_offsets=None, n_waypoints=None):
        self._wp = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        if wp_offsets is not None:
            wo = np.asarray(wp_offsets, dtype=np.int64)
            n_traj = wo.size - 1
            self._so = (wo - np.arange(n_traj + 1)).astype(np.int32)  # waypoint offsets -> segment offsets
        else:
            assert n_waypoints is not None and n_waypoints >= 2 and self._wp.shape[0] % n_waypoints == 0
            n_traj = self._wp.shape[0] // n_waypoints
            self._so = (np.arange(n_traj + 1) * (n_waypoints - 1)).astype(np.int32)

    def setTimeAllocation(self, T):
        self._T = np.ascontiguousarray(T, dtype=np.float64).ravel()

    def setBoundary(self, bc):
        self._bc = np.ascontiguousarray(bc, dtype=np.float64)

    def setCorridor(self, lo, hi):
        if lo is None or hi is None:
            self._lo = self._hi = None
            return
        self._lo = np.ascontiguousarray(lo, dtype=np.float64).reshape(-1, 3)
        self._hi = np.ascontiguousarray(hi, dtype=np.float64).reshape(-1, 3)

    def setRows(self, rows_per_segment, tau=None, deriv=None, lo=None, hi=None):
        if not rows_per_segment:
            self._rows = None
            return
        K = int(rows_per_segment)
        self._rows = (K, np.ascontiguousarray(tau, dtype=np.float64).reshape(-1, K), np.ascontiguousarray(deriv, dtype=np.int32).reshape(-1, K),
                      np.ascontiguousarray(lo, dtype=np.float64).reshape(-1, K, 3), np.ascontiguousarray(hi, dtype=np.float64).reshape(-1, K, 3))

    def solve(self):
        if self._wp is None or self._T is None:
            return False
        n_traj = self._so.size - 1
        if self._T.size != int(self._so[-1]):
            return False
        if self._lo is not None and (self._lo.shape != self._wp.shape or self._hi.shape != self._wp.shape):
            return False
        bc = self._bc if self._bc is not None else np.zeros((n_traj, 2, self._r - 1, 3))
        if self._ctx is None:
            self._ctx = Context(self._device)  # raises UavqpError without libuavqp.so / without a GPU
            # the three settings the reference passes to its solver (minimum_control.cpp:160-162)
            self._ctx.set_settings(warm_start=1, eps_prim_inf=1e-3, max_iter=1000)
        if self._rows is not None:
            K, tau, drv, rlo, rhi = self._rows
            if tau.shape[0] != self._T.size:
                return False
            self._coef, self.status, self.iterations = self._ctx.solve_rows_batch_host(
                self._r, self._so, self._wp, self._T, bc, self._lo, self._hi, K, tau, drv, rlo, rhi)
        elif self._lo is not None:
            self._coef, self.status, self.iterations = self._ctx.solve_corridor_batch_host(
                self._r, self._so, self._wp, self._T, bc, self._lo, self._hi)
        else:
            self._coef, self.status = self._ctx.solve_batch_host(self._r, self._so, self._wp, self._T, bc)
        return bool(np.all(self.status == _lib.UAVQP_SOLVED))

    def getPolyCoeff(self, traj=None):
        if traj is None:
            return self._coef.copy()
        s0, s1 = int(self._so[traj]), int(self._so[traj + 1])
        nc = 2 * self._r
        return self._coef[3 * nc * s0:3 * nc * s1].reshape(3, s1 - s0, nc).copy()
