"""BASELINE config 5 as one call -- a BINDING of the C-ABI entry uavqp_corridor_pipeline_device (include/uavqp.h; the sequencing
itself is host C++ in csrc/uavqp_pipeline.h, the same code a C++ planner reaches through TrajOptimizer::solvePipeline):
plain solve -> corridor boxes from the obstacle cloud (robot ellipsoid of KinoAstar::isCollisionFree, kino_astar.cpp:721-758,
attitude of that solve) -> at most `max_rounds` x (corridor-constrained solve with the working set carried over + time
re-allocation) -> SE(3) collision check of the result against a uniform grid over the cloud -> repair of what it flags (boxes
halved towards the searcher's waypoints, last resort: the reference's own equality problem), at most `repair_rounds` times.
The reference has no such loop (constant 1.0 s allocation, test_minimum_jerk.cpp:65-71): parity is per inner solve (SURVEY.md
section 8-a').  torch is used here for ONE thing: allocating the output buffers on the caller's device."""


def corridor_pipeline_device(ctx, r, seg_offsets, waypoints, times, bc, obstacles, max_segments, robot_r=0.4, robot_h=0.1,
                             h_max=0.8, v_max=7.0, a_max=10.0, max_rounds=5, samples_per_seg=16, max_stretch=2.0,
                             check_samples=100, grid=None, repair_rounds=2, check_robot=None, out=None):
    """All array arguments are torch CUDA tensors on the ctx's device (float64 / int32), ragged layout of include/uavqp.h:
    seg_offsets [n+1] int32, waypoints [sum(M+1), 3], times [sum M] (UPDATED IN PLACE by the re-allocation), bc [n, 2, r-1, 3],
    obstacles [n_obs, 3].  robot / limit defaults: test_kino_astar_searching.launch:49-57.
    check_robot = (r, h) of the ellipsoid used by the final check when it differs from the one the boxes were built with (a safety
    margin; the tests use it to force the repair path).  grid: handle of Context.obstacle_grid_build (cell = check radius + 0.1), or
    None = built inside the call.  out: dict of pre-allocated output tensors (coeff, status, corr_lo, corr_hi, first_hit) to reuse.
    Returns dict(coeff, status, corr_lo, corr_hi (the boxes of the final solve), first_hit (of the final check, check_samples = none),
    collision_free (bool per trajectory), colliding_before_repair, colliding_with_blocked_waypoints (of those: trajectories with a
    waypoint the cloud leaves no room around -- not repairable by narrower boxes), repairs, rounds, still_stretching, check_dt,
    check_samples, all_solved); `collision_free` is formed on first access (see _PipelineResult)."""
    import torch
    n = seg_offsets.numel() - 1
    rows = waypoints.reshape(-1, 3).shape[0]
    dev = waypoints.device
    n_obs = 0 if obstacles is None else obstacles.reshape(-1, 3).shape[0]
    total_seg = times.numel()
    out = out or {}
    coeff = out.get("coeff") if out.get("coeff") is not None else torch.zeros(total_seg * 3 * 2 * r, dtype=torch.float64, device=dev)
    status = out.get("status") if out.get("status") is not None else torch.zeros(n, dtype=torch.int32, device=dev)
    lo = out.get("corr_lo") if out.get("corr_lo") is not None else torch.zeros((rows, 3), dtype=torch.float64, device=dev)
    hi = out.get("corr_hi") if out.get("corr_hi") is not None else torch.zeros((rows, 3), dtype=torch.float64, device=dev)
    first_hit = out.get("first_hit") if out.get("first_hit") is not None else torch.full((n,), int(check_samples), dtype=torch.int32, device=dev)
    chk = (0.0, 0.0) if check_robot is None else check_robot
    res = ctx.corridor_pipeline_device(r, n, 0, int(max_segments), total_seg, seg_offsets, waypoints, times, bc, obstacles, n_obs, coeff, status,
                                       lo, hi, first_hit, grid=grid, robot_r=robot_r, robot_h=robot_h, h_max=h_max, v_max=v_max, a_max=a_max,
                                       max_rounds=int(max_rounds), samples_per_seg=int(samples_per_seg), max_stretch=max_stretch,
                                       check_samples=int(check_samples), repair_rounds=int(repair_rounds), check_robot_r=float(chk[0]),
                                       check_robot_h=float(chk[1]))
    return _PipelineResult(coeff=coeff, status=status, corr_lo=lo, corr_hi=hi, first_hit=first_hit, check_samples=int(check_samples),
                           colliding_before_repair=res["colliding_before_repair"], colliding_with_blocked_waypoints=res["colliding_with_blocked_waypoints"],
                           repairs=res["repairs"], rounds=res["rounds"], still_stretching=res["still_stretching"], check_dt=res["check_dt"],
                           all_solved=res["unsolved"] == 0)


class _PipelineResult(dict):
    """The result dict.  `collision_free` (first_hit >= check_samples: a torch comparison = one more launch) is a LAZY key: it is a member of
    the dict for every access path (`in`, get, keys / items / values, iteration, dict(res), **res) and is formed the first time any of them
    touches it.  `check_samples` (the sample count of the final check; = "no hit" in first_hit) is a documented entry of the result."""
    _LAZY = "collision_free"

    def _materialise(self):
        if not dict.__contains__(self, self._LAZY):
            dict.__setitem__(self, self._LAZY, dict.__getitem__(self, "first_hit") >= dict.__getitem__(self, "check_samples"))

    def __missing__(self, key):
        if key == self._LAZY:
            self._materialise()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return key == self._LAZY or dict.__contains__(self, key)

    def get(self, key, default=None):
        if key == self._LAZY:
            return self[key]
        return dict.get(self, key, default)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + (0 if dict.__contains__(self, self._LAZY) else 1)

    def copy(self):
        self._materialise()
        return dict(self)
