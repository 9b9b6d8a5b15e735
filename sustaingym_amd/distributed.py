"""Multi-GPU sharding of independent environments.

Environment instances never interact, so the path shards with NO data-path collective:
rank r of W owns the contiguous global environment ids [r*N/W, (r+1)*N/W), replicates the
network descriptor / MOER table / constants, and steps its shard with its own engine.  The only
collective is an all-gather of a small metrics vector per report interval (RCCL over xGMI on
GPU ranks — latency-bound at a few hundred bytes — or gloo on CPU ranks in tests).
"""
from __future__ import annotations

import numpy as np

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
    per_rank = torch.stack(out).cpu().numpy()
    return per_rank, per_rank.sum(axis=0)


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
