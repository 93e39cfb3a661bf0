"""Object sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

Objects are independent networks with independent rays, gradients and Adam state, so
the K objects are split into contiguous blocks and NO data-path collective is needed.  Two couplings remain:
render_rays.py:89-94 (if ANY object of the batch has an empty mask, that loss term is zero for ALL objects: a
pair of flags that must be global) and the replicated background network (its rays are split over the ranks:
global mask counts before the step, the gradient SUM after it).  An iteration therefore needs exactly TWO
collectives (train.ShardedIteration): one 4-int SUM before the step (`pack_pre`: the two flags as counts of
objects with an empty mask + the background's two mask counts) and one fp32 SUM after it (background gradient
with its four loss terms appended), the latter on RCCL's own stream under the object kernel."""
import os
from typing import Tuple

import torch


def _active(group=None) -> bool:
    """Collectives run when a process group spans more than one rank -- or, with OBJNERF_DIST_SELFTEST=1, also on
    a single-rank group (lets one GPU exercise the RCCL code path: `torchrun --nproc-per-node 1 bench.py`)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("OBJNERF_DIST_SELFTEST") == "1"


def shard_objects(K: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the K objects owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(K, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_flags(local_flags: torch.Tensor, group=None) -> torch.Tensor:
    """MAX-reduce the two early-return flags over all ranks (in place)."""
    import torch.distributed as dist
    if _active(group):
        dist.all_reduce(local_flags, op=dist.ReduceOp.MAX, group=group)
    return local_flags


def total_loss(local_terms: torch.Tensor, color_scaling=5.0, opacity_scaling=10.0, feat_scaling=5.0,
               group=None) -> torch.Tensor:
    """Scalar batch loss = sum over ALL objects of (depth + cs*colour + os*opacity + fs*feature),
    loss.py:79,99-101; logging only."""
    import torch.distributed as dist
    w = torch.tensor([1.0, color_scaling, opacity_scaling, feat_scaling], device=local_terms.device)
    t = (local_terms * w).sum().reshape(1)
    if _active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce (RCCL when the tensor is on a GPU); no-op for a single process."""
    import torch.distributed as dist
    if _active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_rays(R: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice of the R rays of ONE replicated network (the background) owned by `rank`."""
    return shard_objects(R, world, rank)


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world size) of this process; (0, 1) without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """In-place broadcast from `src` (replicated networks start from identical weights); no-op for one process."""
    import torch.distributed as dist
    if _active(group):
        dist.broadcast(t, src=src, group=group)
    return t


def pack_pre(obj_flags, bg_counts, device) -> torch.Tensor:
    """The pre-step exchange of an iteration as ONE int32[4] buffer: [objects with an empty label-1 mask, objects
    with an empty label!=2 mask (this rank's 0/1 flags: after the SUM, > 0 means the early return of
    render_rays.py:89-94 applies everywhere), background n(label==1), background n(label!=2) of this rank's ray
    slice].  Either part may be None (no foreground objects on this rank / no background network)."""
    pre = torch.zeros(4, dtype=torch.int32, device=device)
    if obj_flags is not None:
        pre[0:2] = obj_flags.reshape(2)
    if bg_counts is not None:
        pre[2:4] = bg_counts.reshape(-1)[:2]
    return pre


def unpack_pre(pre: torch.Tensor):
    """-> (global object flags int32[2], global background counts int32[1,2], background flags int32[2])."""
    gflags = (pre[0:2] > 0).to(torch.int32)
    bg_counts = pre[2:4].reshape(1, 2).contiguous()
    bg_flags = (pre[2:4] == 0).to(torch.int32)
    return gflags, bg_counts, bg_flags


def allreduce_sum_async(t: torch.Tensor, group=None):
    """SUM all-reduce that returns a work handle (or None when no group is active).  On a GPU the collective runs on
    the backend's own stream after the work queued on the CURRENT stream; `wait()` makes the current stream wait
    for it -- everything launched in between overlaps the transfer."""
    import torch.distributed as dist
    if not _active(group):
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_max_(t: torch.Tensor, group=None) -> torch.Tensor:
    import torch.distributed as dist
    if _active(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t
