"""Multi-GPU sharding of independent environments.

Environment instances never interact, so the path shards with NO data-path collective:
rank r of W owns the contiguous global environment ids [r*N/W, (r+1)*N/W), replicates the
network descriptor / MOER table / constants, and steps its shard with its own engine.  The only
collective is an all-gather of a small metrics vector per report interval (RCCL over xGMI on
GPU ranks — latency-bound at a few hundred bytes — or gloo on CPU ranks in tests).
"""
from __future__ import annotations

import numpy as np

from .hostio import to_host

METRIC_NAMES = ('profit', 'carbon_cost', 'excess_charge', 'env_steps', 'episodes_finished',
                'envs_with_status')


def shard_range(global_envs: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous env-id range of ``rank`` (sizes differ by at most one)."""
    assert 0 <= rank < world
    lo = (global_envs * rank) // world
    hi = (global_envs * (rank + 1)) // world
    return lo, hi


def shard_seeds(base_seed: int | None, rank: int, world: int, global_envs: int) -> list[int | None]:
    """Per-environment reset seeds of a shard: global env i gets ``base_seed + i`` (the gymnasium
    VectorEnv convention), independent of how many ranks the job runs on."""
    lo, hi = shard_range(global_envs, rank, world)
    return [None if base_seed is None else base_seed + i for i in range(lo, hi)]


class WorldMismatch(RuntimeError):
    """The launcher's world (RANK / WORLD_SIZE) contradicts what the caller asked for."""


def resolve_world(requested_gpus: int, environ) -> tuple[int, int, int, bool]:
    """``(rank, local_rank, world, must_spawn)`` for a job asked to run on ``requested_gpus`` GPUs of one node.

    * launched by ``torch.distributed.run`` (WORLD_SIZE set): the world must be exactly ``requested_gpus``
      — a silent one-GPU run labelled as N GPUs is the failure this guards against;
    * plain ``python ... --gpus N`` with N > 1: the caller has to start N ranks itself (``must_spawn``);
    * N = 1: single process."""
    if requested_gpus < 1:
        raise WorldMismatch(f'--gpus must be >= 1, got {requested_gpus}')
    ws = environ.get('WORLD_SIZE')
    if ws is None:
        return 0, 0, (requested_gpus if requested_gpus > 1 else 1), requested_gpus > 1
    world = int(ws)
    if world != requested_gpus:
        raise WorldMismatch(f'--gpus {requested_gpus} but the launcher started WORLD_SIZE={world} ranks; '
                            f'refusing to report a {world}-rank run as {requested_gpus} GPUs')
    rank = int(environ.get('RANK', '0'))
    local_rank = int(environ.get('LOCAL_RANK', str(rank)))
    if not (0 <= rank < world):
        raise WorldMismatch(f'RANK={rank} outside WORLD_SIZE={world}')
    return rank, local_rank, world, False


def all_gather_vector(local: np.ndarray, device=None) -> np.ndarray:
    """All-gathers a small float64 vector; returns ``[W, len]`` (identity without a process group)."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return local[None, :].copy()
    t = torch.as_tensor(local, dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return to_host(torch.stack(out))


def metrics_vector(metrics: dict[str, float]) -> np.ndarray:
    return np.array([metrics[k] for k in METRIC_NAMES], dtype=np.float64)


def all_gather_metrics(local: np.ndarray, device=None):
    """All-gathers the per-rank metrics vector; returns ``(per_rank[W, 6], total[6])``.
    Works with any initialised ``torch.distributed`` backend (nccl = RCCL on GPU, gloo on CPU);
    without an initialised process group it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local[None, :].copy(), local.copy()
    t = torch.as_tensor(local, dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    per_rank = to_host(torch.stack(out))
    return per_rank, per_rank.sum(axis=0)


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
