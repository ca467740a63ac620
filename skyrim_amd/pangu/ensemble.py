"""Member-parallel ensembles: one process per GPU, members sharded round-robin over ranks.

The reference has no counterpart for this (its ``GlobalEnsemble`` runs *models* sequentially on one
GPU and averages them, /root/reference/skyrim/core/models/ensemble.py:51-67,73-108); the semantics kept
here are "mean over a new member dimension".  The rollout of a member needs no communication at all;
the only exchange is the reduction that forms the ensemble mean / spread of a saved step:

    mean   = sum_over_ranks(sum_local(x)) / M
    spread = sqrt(sum_over_ranks(sum_local((x - mean)^2)) / M)          (two-pass: no cancellation -- the members of a perturbed-
                                                                         IC ensemble differ by 1e-3 of a standard deviation)

``sum_over_ranks`` is either a ring ``all_reduce`` or -- the default on the 8-GPU xGMI full mesh, SURVEY.md 5 -- an
``all_gather`` of the per-rank partial sums followed by a local reduction: every rank sends its 286 MB partial to its 7 peers
over 7 links at once (1.9 ms at 153 GB/s per link) instead of pushing 2 (N-1)/N of it around a ring (3.3 ms), and the result is
bit-identical on every rank (same summation order).  BASELINE configs[4] (50 members on 8 GPUs, "all-gather of outputs"): members
round-robin 7,7,6,6,6,6,6,6; per SAVED step one reduction for mean / spread and optionally one all-gather of the member states.

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


def _group_is_up():
    """A process group exists -- possibly of ONE rank: the collectives are then issued all the same (a rank's data lands on itself),
    which is how the RCCL path is exercised on a 1-GPU box (tests/test_rccl_gpu.py).  Without a group nothing is communicated."""
    return dist.is_available() and dist.is_initialized()


def sum_over_ranks(t: torch.Tensor, how: str = "allgather") -> torch.Tensor:
    """Sum of one tensor per rank, on every rank, in place.  "allgather": all_gather of the partials + local sum in rank order
    (direct exchange over the xGMI full mesh; identical bits on every rank); "allreduce": ring all-reduce."""
    w = _world()
    if not _group_is_up():
        return t
    if how == "allreduce":
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
    if how != "allgather":
        raise ValueError("how is 'allgather' or 'allreduce'")
    parts = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(parts, t.contiguous()[None])          # concatenation form along dim 0 (the one every backend implements)
    torch.sum(parts, dim=0, out=t)
    return t


def ensemble_mean_spread(local_states: list[torch.Tensor], n_members: int, how: str = "allgather") -> tuple[torch.Tensor, torch.Tensor]:
    """Ensemble mean and spread (population std) of states sharded over ranks.

    ``local_states``: this rank's member states (same shape each).  Every rank returns the full mean / spread.
    """
    if not local_states:
        raise ValueError("every rank must own at least one member (use world_size <= n_members)")
    s = torch.stack(local_states).sum(0) if len(local_states) > 1 else local_states[0].clone()
    mean = sum_over_ranks(s, how) / n_members
    sq = torch.zeros_like(mean)
    for x in local_states:
        sq += (x - mean) ** 2
    return mean, (sum_over_ranks(sq, how) / n_members).sqrt()


def gather_members(local_states: list[torch.Tensor], n_members: int) -> torch.Tensor:
    """All member states on every rank, ordered by member id: (n_members, ...).  One all-gather of ceil(M / world) slots per rank
    (ranks owning one member less send a zero slot)."""
    w = _world()
    per = (n_members + w - 1) // w
    local = torch.stack(local_states + [torch.zeros_like(local_states[0])] * (per - len(local_states)))
    if _group_is_up():
        buf = torch.empty((w * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, local)
        buf = buf.view((w, per) + tuple(local.shape[1:]))
    else:
        buf = local[None]
    states = [None] * n_members
    for r in range(w):
        for j, m in enumerate(member_shard(n_members, r, w)):
            states[m] = buf[r, j]
    return torch.stack(states)


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

    def steps(self, x0: torch.Tensor, n_steps: int, save_every: int = 1, gather: bool = False, how: str = "allgather"):
        """Generator over the SAVED steps of the ensemble rollout (BASELINE configs[4]): every local member advances ``save_every``
        model steps without any communication, then one reduction forms the ensemble mean / spread of that lead time and, with
        ``gather``, one all-gather collects the member states.  Yields dicts {"step", "mean", "spread", "local_states"[, "members"]}."""
        states = [perturbed_member(x0, self.std, m, self.scale) for m in self.members]
        for k in range(1, n_steps + 1):
            states = [self.step_fn(x) for x in states]
            if k % save_every == 0 or k == n_steps:
                mean, spread = ensemble_mean_spread(states, self.n_members, how)
                out = {"step": k, "mean": mean, "spread": spread, "local_members": self.members, "local_states": states}
                if gather:
                    out["members"] = gather_members(states, self.n_members)
                yield out

    def run(self, x0: torch.Tensor, n_steps: int, gather: bool = False, how: str = "allgather") -> dict:
        """Roll every local member ``n_steps`` forward.  Returns the ensemble mean and spread of the final states
        (identical on every rank) and, with ``gather``, all member states ordered by member id (all-gather)."""
        last = None
        for last in self.steps(x0, n_steps, save_every=n_steps, gather=gather, how=how):
            pass
        return last
