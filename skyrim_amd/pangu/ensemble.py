"""Member-parallel ensembles: one process per GPU, members sharded round-robin over ranks.

The reference has no counterpart for this (its ``GlobalEnsemble`` runs *models* sequentially on one
GPU and averages them, /root/reference/skyrim/core/models/ensemble.py:51-67,73-108); the semantics kept
here are "mean over a new member dimension".  The rollout of a member needs no communication at all;
the only exchange is the reduction that forms the ensemble mean / spread of a saved step:

    mean   = all_reduce(sum_local(x)) / M
    spread = sqrt(all_reduce(sum_local((x - mean)^2)) / M)          (two-pass: no cancellation)

``torch.distributed`` with backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def member_shard(n_members: int, rank: int, world_size: int) -> list[int]:
    """Members owned by ``rank`` (round-robin: 50 members on 8 ranks -> 7,7,6,6,6,6,6,6)."""
    return list(range(rank, n_members, world_size))


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def ensemble_mean_spread(local_states: list[torch.Tensor], n_members: int) -> tuple[torch.Tensor, torch.Tensor]:
    """Ensemble mean and spread (population std) of states sharded over ranks.

    ``local_states``: this rank's member states (same shape each; may be empty on a rank that owns
    no member).  Every rank returns the full mean / spread.
    """
    if local_states:
        s = torch.stack(local_states).sum(0)
    else:
        raise ValueError("every rank must own at least one member (use world_size <= n_members)")
    if _world() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    mean = s / n_members
    sq = torch.zeros_like(mean)
    for x in local_states:
        sq += (x - mean) ** 2
    if _world() > 1:
        dist.all_reduce(sq, op=dist.ReduceOp.SUM)
    return mean, (sq / n_members).sqrt()


def rollout_members(step_fn, initial_states: list[torch.Tensor], n_steps: int) -> list[torch.Tensor]:
    """Advance each local member ``n_steps`` times with ``step_fn(x) -> x_next`` (device resident)."""
    out = []
    for x in initial_states:
        for _ in range(n_steps):
            x = step_fn(x)
        out.append(x)
    return out
