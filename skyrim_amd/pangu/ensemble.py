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


def perturbed_member(x0: torch.Tensor, std: torch.Tensor, member: int, scale: float = 1e-3) -> torch.Tensor:
    """Member ``i`` of a perturbed-initial-condition ensemble: x0 + scale * std_c * N(0,1; seed 1000+i)
    (SURVEY.md 8(d), config 5).  Member 0 is the unperturbed control.  Deterministic per device type."""
    if member == 0:
        return x0.clone()
    gen = torch.Generator(device=x0.device).manual_seed(1000 + member)
    noise = torch.randn(x0.shape, generator=gen, device=x0.device, dtype=x0.dtype)
    return x0 + scale * std.to(x0.device, x0.dtype)[:, None, None] * noise


class MemberParallelEnsemble:
    """N-member ensemble of one model, members sharded round-robin over the ranks of the default process group
    (one process per GPU).  ``step_fn`` advances one member by one model step on this rank's device."""

    def __init__(self, step_fn, n_members: int, channel_std: torch.Tensor, perturb_scale: float = 1e-3):
        self.step_fn = step_fn
        self.n_members = n_members
        self.std = channel_std
        self.scale = perturb_scale
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = _world()
        if self.world > n_members:
            raise ValueError("more ranks than ensemble members")
        self.members = member_shard(n_members, self.rank, self.world)

    def run(self, x0: torch.Tensor, n_steps: int, gather: bool = False) -> dict:
        """Roll every local member ``n_steps`` forward.  Returns the ensemble mean and spread of the final states
        (identical on every rank) and, with ``gather``, all member states ordered by member id (all-gather)."""
        ics = [perturbed_member(x0, self.std, m, self.scale) for m in self.members]
        finals = rollout_members(self.step_fn, ics, n_steps)
        mean, spread = ensemble_mean_spread(finals, self.n_members)
        out = {"mean": mean, "spread": spread, "local_members": self.members, "local_states": finals}
        if gather:
            per = (self.n_members + self.world - 1) // self.world
            local = torch.stack(finals + [torch.zeros_like(finals[0])] * (per - len(finals)))
            if self.world > 1:
                buf = [torch.empty_like(local) for _ in range(self.world)]
                dist.all_gather(buf, local)
            else:
                buf = [local]
            states = [None] * self.n_members
            for r in range(self.world):
                for j, m in enumerate(member_shard(self.n_members, r, self.world)):
                    states[m] = buf[r][j]
            out["members"] = torch.stack(states)
        return out
