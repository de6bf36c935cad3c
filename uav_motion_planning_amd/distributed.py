"""Multi-GPU sharding of a trajectory batch: one process per GPU, torch.distributed (RCCL on ROCm).

The path shards trivially -- trajectories are independent QPs: every rank
solves a contiguous slice of the batch with no data-path collective; the single exchange step is the
all-gather of the solved coefficient shards (north star: RCCL all-gather over xGMI).  Equal shards use
all_gather_into_tensor; ragged shards are padded to the largest shard (payload is small next to HBM
bandwidth: 1536 B per 8-segment snap trajectory).

The same code runs on the gloo backend (CPU tensors) for the world_size-2 tests in tests/.
"""
import numpy as np


def shard_bounds(n_traj, world):
    """Contiguous equal-count partition: rank g owns [bounds[g], bounds[g+1])."""
    return [(n_traj * g) // world for g in range(world + 1)]


def shard_bounds_ragged(seg_offsets, world):
    """Contiguous partition balanced by total segment count (work is proportional to M_b)."""
    so = np.asarray(seg_offsets, dtype=np.int64)
    n_traj = so.size - 1
    total = int(so[-1])
    bounds = [0]
    for g in range(1, world):
        target = total * g / world
        b = int(np.searchsorted(so, target, side="left"))
        b = min(max(b, bounds[-1]), n_traj)
        bounds.append(b)
    bounds.append(n_traj)
    return bounds


def local_slice(batch, lo, hi):
    """Cut trajectories [lo, hi) out of a batch dict (workloads.py layout); offsets are re-based to 0."""
    so = np.asarray(batch["seg_offsets"], dtype=np.int64)
    s0, s1 = int(so[lo]), int(so[hi])
    wp = np.asarray(batch["waypoints"]).reshape(-1, 3)
    return dict(r=batch["r"], M=batch.get("M", 0),
                seg_offsets=(so[lo:hi + 1] - s0).astype(np.int32),
                waypoints=wp[s0 + lo:s1 + hi].copy(),
                times=np.asarray(batch["times"]).reshape(-1)[s0:s1].copy(),
                bc=np.asarray(batch["bc"]).reshape(so.size - 1, -1)[lo:hi].copy())


def allgather_coeffs(local_coeff, shard_numels, group=None):
    """All-gather the flat coefficient shards of every rank.

    local_coeff   1-D float64 torch tensor (CUDA for nccl/RCCL, CPU for gloo), this rank's shard
    shard_numels  list of shard lengths for all ranks (known on every rank from the partition)
    Returns the concatenated full coefficient tensor on the caller's device.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(shard_numels) == world and local_coeff.numel() == shard_numels[dist.get_rank(group)]
    mx = max(shard_numels)
    if all(n == mx for n in shard_numels):
        full = torch.empty(world * mx, dtype=local_coeff.dtype, device=local_coeff.device)
        dist.all_gather_into_tensor(full, local_coeff.contiguous(), group=group)
        return full
    padded = torch.zeros(mx, dtype=local_coeff.dtype, device=local_coeff.device)
    padded[:local_coeff.numel()] = local_coeff
    full = torch.empty(world * mx, dtype=local_coeff.dtype, device=local_coeff.device)
    dist.all_gather_into_tensor(full, padded, group=group)
    return torch.cat([full[g * mx:g * mx + shard_numels[g]] for g in range(world)])


def solve_sharded(batch, solve_local, group=None, device=None):
    """The whole multi-GPU step of SURVEY.md section 8-e on every rank: cut this rank's contiguous shard out of the
    batch (balanced by segment count), solve it, all-gather coefficients and statuses.

    batch        workloads.py layout (seg_offsets, waypoints, times, bc, r), identical on every rank
    solve_local  callable(shard) -> (coeff float64 [3*2r*segments of the shard], status int32 [trajectories of the shard]);
                 on a GPU rank:  lambda s: ctx.solve_batch_host(s["r"], s["seg_offsets"], s["waypoints"], s["times"],
                                                                 s["bc"].reshape(-1, 2, s["r"] - 1, 3))
    device       torch device of the gathered tensors (cuda:<local rank> for nccl = RCCL, cpu for gloo)
    Returns (coeff, status) for the WHOLE batch in batch order, as torch tensors on `device`."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r = int(batch["r"])
    so = np.asarray(batch["seg_offsets"], dtype=np.int64)
    bounds = shard_bounds_ragged(so, world)
    shard = local_slice(batch, bounds[rank], bounds[rank + 1])
    coef, st = solve_local(shard)
    dev = torch.device("cpu") if device is None else device
    numels = [3 * 2 * r * int(so[bounds[g + 1]] - so[bounds[g]]) for g in range(world)]
    counts = [bounds[g + 1] - bounds[g] for g in range(world)]
    full = allgather_coeffs(torch.as_tensor(np.ascontiguousarray(coef), dtype=torch.float64).to(dev), numels, group)
    status = allgather_coeffs(torch.as_tensor(np.ascontiguousarray(st), dtype=torch.int32).to(dev), counts, group)
    return full, status
