"""Multi-GPU model of the env hot path: a contiguous batch split, one process (and one q1env_t handle) per GPU,
NO collective on the data path.  Envs are independent (no cross-env term anywhere in reference phys.py / env.py),
so the only things ranks ever exchange are the barrier / timing reduction of a benchmark and, optionally, a
host-side gather of results.  The counter RNG is keyed by the GLOBAL env index so results do not depend on the
number of shards."""
from typing import Callable, List, Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """(start, count) of rank's contiguous slice of `total` envs; the remainder goes to the first ranks."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def shard_plan(total: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(total, r, world) for r in range(world)]


def make_shard_env(config, rank: int, world: int, factory: Callable, **kw):
    """Build this rank's env over its slice: factory(config_with_local_num_envs, env_index_base=start, **kw)."""
    import dataclasses
    start, count = shard_range(config.num_envs, rank, world)
    if count == 0:
        raise ValueError(f"rank {rank} of {world} owns no env out of {config.num_envs}")
    return factory(dataclasses.replace(config, num_envs=count), env_index_base=start, **kw), start, count


def max_over_ranks(seconds: float, device=None) -> float:
    """Elapsed time of the slowest rank (what a sharded benchmark must report)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local: np.ndarray, dst: int = 0):
    """Host-side concatenation of per-shard result rows in global env order on rank `dst` (None elsewhere)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, parts, dst=dst)
    return np.concatenate(parts, axis=0) if parts is not None else None
