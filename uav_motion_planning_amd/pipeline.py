"""BASELINE config 5 as one call: plain solve -> corridor boxes from the obstacle cloud (robot ellipsoid of
KinoAstar::isCollisionFree, kino_astar.cpp:721-758, attitude of that solve) -> corridor-constrained solve and time
re-allocation, at most `max_rounds` times with the working set carried from round to round -> SE(3) collision check of the
result against a uniform grid over the cloud -> REPAIR: the boxes only bound the knots, and they were built with the attitude of
the first (equality) solve, so the check of the final polynomials is the arbiter; the boxes of every trajectory it flags are
halved towards the searcher's waypoints and that trajectory is re-solved, at most `repair_rounds` times (the last resort --
zero-width boxes -- is the reference's own equality problem); what still collides is reported in `collision_free`.
Everything stays in device buffers; this is host-side sequencing of the
C-ABI calls of include/uavqp.h only (the reference has no such loop: the constant 1.0 s allocation of
test_minimum_jerk.cpp:65-71 is the starting point, parity is per inner solve -- SURVEY.md section 8-a')."""
import numpy as np

from . import _lib


def corridor_pipeline_device(ctx, r, seg_offsets, waypoints, times, bc, obstacles, max_segments, robot_r=0.4, robot_h=0.1,
                             h_max=0.8, v_max=7.0, a_max=10.0, max_rounds=5, samples_per_seg=16, max_stretch=2.0,
                             check_samples=100, grid=None, repair_rounds=2, check_robot=None):
    """All array arguments are torch CUDA tensors on the ctx's device (float64 / int32), ragged layout of include/uavqp.h:
    seg_offsets [n+1] int32, waypoints [sum(M+1), 3], times [sum M] (UPDATED IN PLACE by the re-allocation), bc [n, 2, r-1, 3],
    obstacles [n_obs, 3].  robot / limit defaults: test_kino_astar_searching.launch:49-57.
    check_robot = (r, h) of the ellipsoid used by the final check when it differs from the one the boxes were built with (a safety
    margin; the tests use it to force the repair path).
    Returns dict(coeff, status, corr_lo, corr_hi (the boxes of the final solve), first_hit (of the final check, check_samples = none),
    collision_free (bool per trajectory), colliding_before_repair, colliding_with_blocked_waypoints (of those: trajectories with a
    waypoint the cloud leaves no room around -- not repairable by narrower boxes), repairs, rounds, still_stretching, iterations)."""
    import torch
    n = seg_offsets.numel() - 1
    rows = waypoints.reshape(-1, 3).shape[0]
    dev = waypoints.device
    n_obs = 0 if obstacles is None else obstacles.reshape(-1, 3).shape[0]
    total_seg = times.numel()
    coeff = torch.zeros(total_seg * 3 * 2 * r, dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    iters = torch.zeros(n, dtype=torch.int32, device=dev)
    changed = torch.zeros(n, dtype=torch.int32, device=dev)
    lo = torch.zeros((rows, 3), dtype=torch.float64, device=dev)
    hi = torch.zeros((rows, 3), dtype=torch.float64, device=dev)
    active = torch.zeros((n, 3, 2), dtype=torch.int64, device=dev)
    first_hit = torch.zeros(n, dtype=torch.int32, device=dev)

    ctx.solve_batch_device(r, n, 0, max_segments, seg_offsets, waypoints, times, bc, coeff, status)
    ctx.corridor_from_cloud_device(r, n, 0, seg_offsets, rows, waypoints, times, coeff, obstacles, n_obs, robot_r, robot_h, h_max, lo, hi)
    rounds, it_hist = 0, []
    for rnd in range(max_rounds):
        ctx.solve_corridor_device(r, n, 0, max_segments, seg_offsets, waypoints, times, bc, lo, hi, coeff, status, iters, active, rnd > 0)
        rounds += 1
        it_hist.append(iters.clone())
        ctx.time_reallocate_device(r, n, 0, seg_offsets, times, coeff, v_max, a_max, samples_per_seg, max_stretch, changed)
        ctx.synchronize()
        if int((changed > 0).sum().item()) == 0:     # the last solve already belongs to the final durations
            break
    else:
        # cap reached with durations changed by the last re-allocation: one more solve so that coeff matches `times`
        ctx.solve_corridor_device(r, n, 0, max_segments, seg_offsets, waypoints, times, bc, lo, hi, coeff, status, iters, active, True)
        it_hist.append(iters.clone())
    chk_r, chk_h = (robot_r, robot_h) if check_robot is None else check_robot
    own_grid = grid is None
    if own_grid:
        grid = ctx.obstacle_grid_build(obstacles, n_obs, chk_r + 0.1)
    repairs, colliding_before, hopeless = 0, None, 0
    try:
        seg_cnt = (seg_offsets[1:] - seg_offsets[:-1]).long()
        traj_of_seg = torch.repeat_interleave(torch.arange(n, device=dev), seg_cnt)
        traj_of_row = torch.repeat_interleave(torch.arange(n, device=dev), seg_cnt + 1)
        wp_rows = waypoints.reshape(-1, 3)
        while True:
            t_tot = torch.zeros(n, dtype=torch.float64, device=dev)
            t_tot.index_add_(0, traj_of_seg, times)
            dt = float(t_tot.max().item()) / max(1, check_samples - 1)
            ctx.ellipsoid_check_grid_device(r, n, 0, seg_offsets, times, coeff, check_samples, 0.0, dt, grid, chk_r, chk_h, first_hit)
            ctx.synchronize()
            hit = first_hit < check_samples
            if colliding_before is None:
                colliding_before = int(hit.sum().item())
                # a trajectory with a waypoint the cloud leaves no room around (degenerate box: the searcher's waypoint itself is
                # within the robot's reach of an obstacle) cannot be helped by narrower boxes
                roomy = torch.ones(n, dtype=torch.bool, device=dev)
                tight_rows = ((hi - lo).min(dim=1).values <= 0.0)
                interior = torch.ones(rows, dtype=torch.bool, device=dev)
                interior[(seg_offsets[:-1].long() + torch.arange(n, device=dev))] = False
                interior[(seg_offsets[1:].long() + torch.arange(n, device=dev))] = False
                roomy[traj_of_row[tight_rows & interior]] = False
                hopeless = int((hit & ~roomy).sum().item())
            hit = hit & roomy
            n_hit = int(hit.sum().item())
            if n_hit == 0 or repairs >= repair_rounds:
                break
            # halve the boxes of the flagged trajectories towards their waypoints (last round: the waypoint equalities), re-solve
            # warm-started, re-allocate once (the durations only ever stretch) and solve again if that changed anything
            shrink = 0.0 if repairs + 1 == repair_rounds else 0.5
            rows_hit = hit[traj_of_row].unsqueeze(1)
            lo.copy_(torch.where(rows_hit, wp_rows - shrink * (wp_rows - lo), lo))
            hi.copy_(torch.where(rows_hit, wp_rows + shrink * (hi - wp_rows), hi))
            ctx.solve_corridor_device(r, n, 0, max_segments, seg_offsets, waypoints, times, bc, lo, hi, coeff, status, iters, active, True)
            ctx.time_reallocate_device(r, n, 0, seg_offsets, times, coeff, v_max, a_max, samples_per_seg, max_stretch, changed)
            ctx.synchronize()
            if int((changed > 0).sum().item()) > 0:
                ctx.solve_corridor_device(r, n, 0, max_segments, seg_offsets, waypoints, times, bc, lo, hi, coeff, status, iters, active, True)
            it_hist.append(iters.clone())
            repairs += 1
    finally:
        if own_grid:
            ctx.obstacle_grid_destroy(grid)
    return dict(coeff=coeff, status=status, corr_lo=lo, corr_hi=hi, first_hit=first_hit, collision_free=first_hit >= check_samples,
                colliding_before_repair=colliding_before, colliding_with_blocked_waypoints=hopeless, repairs=repairs, rounds=rounds,
                still_stretching=int((changed > 0).sum().item()), iterations=it_hist, check_dt=dt,
                all_solved=bool((status == _lib.UAVQP_SOLVED).all().item()))
