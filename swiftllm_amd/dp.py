"""Request-sharded data parallelism across the GPUs of one node.

The reference has no multi-GPU support at all (README.md:28-30, 54). The MI355X plan:
one process per GPU, each a full replica with its OWN KV pool and block manager; a request lives on
one replica for its whole life, so there is NO data-path collective and no RCCL traffic. What the
ranks share is bookkeeping only — who takes which requests, a barrier around timed regions, and the
reduction of per-rank counters — done over a `gloo` group on the host.
"""
import datetime
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_units: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `num_units` requests owned by `rank`
    (the first num_units % world_size ranks take one extra)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(num_units, world_size)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard(items: Sequence, rank: int, world_size: int) -> List:
    b, e = shard_bounds(len(items), rank, world_size)
    return list(items[b:e])


def least_loaded(outstanding_tokens: Sequence[int]) -> int:
    """Online routing rule: a new request goes to the replica with the fewest outstanding tokens
    (ties -> lowest rank). Requests never migrate afterwards."""
    best = 0
    for i, v in enumerate(outstanding_tokens):
        if v < outstanding_tokens[best]:
            best = i
    return best


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torchrun exports them; (0, 0, 1) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_control_group(timeout_s: int = 600) -> bool:
    """Join the host-side control group (gloo) when launched with WORLD_SIZE > 1."""
    _, _, world = env_rank_world()
    if world <= 1:
        return False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return True


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_job(local_units: float, local_seconds: float) -> Tuple[float, float]:
    """Whole-job totals: (sum of units over ranks, max of seconds over ranks)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(local_units), float(local_seconds)
    units = torch.tensor([float(local_units)], dtype=torch.float64)
    secs = torch.tensor([float(local_seconds)], dtype=torch.float64)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dist.all_reduce(secs, op=dist.ReduceOp.MAX)
    return float(units), float(secs)


def gather_lists(local: list) -> List[list]:
    """Every rank's list, in rank order (e.g. per-replica token streams)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    return out
