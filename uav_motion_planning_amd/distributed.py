"""Multi-GPU sharding of a trajectory batch: one process per GPU.

The path shards trivially -- trajectories are independent QPs: every rank solves a contiguous slice of the batch with no
data-path collective; the single exchange step is the all-gather of the solved coefficient shards (north star: RCCL
all-gather over xGMI).  The partition and the collective are entry points of the C ABI (include/uavqp.h:
uavqp_shard_bounds[_ragged], uavqp_comm_create, uavqp_allgather_coeffs / _status), so a C++ planner shards exactly like this
module does; here they are driven from Python, with torch.distributed used only for the rendezvous (shipping RCCL's
unique id) -- or, when the ctx has no communicator (the world_size-2 gloo tests on CPU), as the stand-in collective.

Everything stays in device tensors: shards are VIEWS of the batch tensors, the solve writes straight into the rank's slot of
the full output tensor, the all-gather is in place.  The reference has no multi-device code (test_minimum_jerk.cpp:73-74 is
its only remark on parallelism).
"""
import ctypes

import numpy as np

from . import _lib


def shard_bounds(n_traj, world):
    """Contiguous equal-count partition (uavqp_shard_bounds): rank g owns [bounds[g], bounds[g+1])."""
    out = (ctypes.c_int32 * (world + 1))()
    _lib.check(_lib.lib().uavqp_shard_bounds(int(n_traj), int(world), out), "uavqp_shard_bounds")
    return list(out)


def shard_bounds_ragged(seg_offsets, world):
    """Contiguous partition balanced by total segment count -- work is proportional to M_b (uavqp_shard_bounds_ragged)."""
    so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
    out = (ctypes.c_int32 * (world + 1))()
    _lib.check(_lib.lib().uavqp_shard_bounds_ragged(so.ctypes.data, int(so.size - 1), int(world), out), "uavqp_shard_bounds_ragged")
    return list(out)


def local_slice(batch, lo, hi):
    """Trajectories [lo, hi) of a batch as VIEWS (numpy arrays or torch tensors on any device; workloads.py layout); only the
    re-based CSR offsets are new.  batch["seg_offsets"] must be host data (numpy): the partition is host arithmetic."""
    so = np.asarray(batch["seg_offsets"], dtype=np.int64)
    s0, s1 = int(so[lo]), int(so[hi])
    n = so.size - 1
    wp = batch["waypoints"].reshape(-1, 3)
    bc = batch["bc"].reshape(n, -1)
    return dict(r=batch["r"], M=batch.get("M", 0),
                seg_offsets=(so[lo:hi + 1] - s0).astype(np.int32),
                waypoints=wp[s0 + lo:s1 + hi],
                times=batch["times"].reshape(-1)[s0:s1],
                bc=bc[lo:hi])


def create_comm(ctx, group=None):
    """Give `ctx` its RCCL communicator over the ranks of a torch.distributed group: rank 0 draws the unique id
    (uavqp_comm_unique_id), the group ships it (an object broadcast: works on gloo and nccl alike), every rank calls
    uavqp_comm_create.  torch.distributed is only the courier here."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ctx.comm_create(rank, world, box[0])
    return ctx


def allgather_shards(local, counts, full, ctx=None, group=None):
    """All-gather of per-rank shards into `full` (shards back to back in rank order; `local` may be this rank's slice of it).
    With a ctx that owns a communicator: uavqp_allgather_coeffs / _status (RCCL, on the ctx stream).  Without one:
    torch.distributed on the tensors' device (gloo on CPU in the tests)."""
    import torch
    if ctx is not None:
        if local.dtype == torch.float64:
            ctx.allgather_coeffs(local, counts, full)
        else:
            ctx.allgather_status(local, counts, full)
        return full
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert len(counts) == world and local.numel() == counts[rank]
    mx = max(counts)
    if all(c == mx for c in counts):
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        return full
    padded = torch.zeros(mx, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    tmp = torch.empty(world * mx, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(tmp, padded, group=group)
    off = 0
    for g in range(world):
        full[off:off + counts[g]] = tmp[g * mx:g * mx + counts[g]]
        off += counts[g]
    return full


def solve_sharded(batch, solve_local, rank, world, ctx=None, group=None):
    """The whole multi-GPU step of SURVEY.md section 8-e on every rank: this rank's contiguous shard (balanced by segment
    count) as views of the batch tensors, solved into its slot of the full output tensors, then the in-place all-gather of
    coefficients and statuses.  No host round trip.

    batch        workloads.py layout, identical on every rank: seg_offsets (numpy, host), waypoints / times / bc (torch
                 tensors on the rank's device, or CPU tensors for the gloo tests), r
    solve_local  callable(shard, coeff_out, status_out): solves the shard (views, see local_slice) INTO the two output
                 views; on a GPU rank e.g.
                     lambda s, c, st: ctx.solve_batch_device(s["r"], len(s["seg_offsets"]) - 1, 0, mmax, d_offsets(s), s["waypoints"],
                                                             s["times"], s["bc"], c, st)
    ctx          a Context that owns a communicator (create_comm): RCCL through the C ABI; None: torch.distributed collective
    Returns (coeff, status, bounds) -- full-batch tensors in batch order on the batch's device."""
    import torch
    r = int(batch["r"])
    so = np.asarray(batch["seg_offsets"], dtype=np.int64)
    bounds = shard_bounds_ragged(so, world)
    dev = batch["times"].device
    coeff = torch.zeros(3 * 2 * r * int(so[-1]), dtype=torch.float64, device=dev)
    status = torch.zeros(so.size - 1, dtype=torch.int32, device=dev)
    c_counts = [3 * 2 * r * int(so[bounds[g + 1]] - so[bounds[g]]) for g in range(world)]
    s_counts = [bounds[g + 1] - bounds[g] for g in range(world)]
    c_off, s_off = sum(c_counts[:rank]), sum(s_counts[:rank])
    c_loc = coeff[c_off:c_off + c_counts[rank]]
    s_loc = status[s_off:s_off + s_counts[rank]]
    if s_counts[rank] > 0:
        solve_local(local_slice(batch, bounds[rank], bounds[rank + 1]), c_loc, s_loc)
    allgather_shards(c_loc, c_counts, coeff, ctx, group)
    allgather_shards(s_loc, s_counts, status, ctx, group)
    return coeff, status, bounds
