"""Object sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

Objects are independent networks with independent rays, gradients and Adam state, so
the K objects are split into contiguous blocks and NO data-path collective is needed.  The one
coupling is render_rays.py:89-94: if ANY object of the batch has an empty mask, that loss term is zero
for ALL objects -- a pair of flags that must be global (2-int all_reduce(MAX))."""
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
