"""Member-parallel ensemble logic on CPU: world_size-2 gloo processes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skyrim_amd.pangu.ensemble import (MemberParallelEnsemble, ensemble_mean_spread, member_shard, perturbed_member,
                                       rollout_members)


def test_member_shard_round_robin():
    sizes = [len(member_shard(50, r, 8)) for r in range(8)]
    assert sizes == [7, 7, 6, 6, 6, 6, 6, 6]
    allm = sorted(m for r in range(8) for m in member_shard(50, r, 8))
    assert allm == list(range(50))


def test_mean_spread_single_process():
    xs = [torch.full((3, 4), float(i)) for i in range(5)]
    mean, spread = ensemble_mean_spread(xs, 5)
    assert torch.allclose(mean, torch.full((3, 4), 2.0))
    assert torch.allclose(spread, torch.full((3, 4), 2.0 ** 0.5))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_members, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(0)
        base = 1e5 + torch.randn(4, 6, 8, generator=gen)              # large mean, small spread (cancellation trap)
        members = member_shard(n_members, rank, world)
        ics = [base + 1.0 * m for m in members]
        outs = rollout_members(lambda x: x * 1.0 + 0.5, ics, 3)       # fake "step"
        mean, spread = ensemble_mean_spread(outs, n_members)
        q.put((rank, mean.numpy(), spread.numpy()))        # by value: the worker may exit before the parent reads
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_matches_single_process():
    world, n_members = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_members, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    gen = torch.Generator().manual_seed(0)
    base = 1e5 + torch.randn(4, 6, 8, generator=gen)
    allm = torch.stack(rollout_members(lambda x: x * 1.0 + 0.5, [base + 1.0 * m for m in range(n_members)], 3))
    ref_mean, ref_spread = allm.double().mean(0), allm.double().std(0, unbiased=False)
    got = [(r, torch.from_numpy(m), torch.from_numpy(s)) for r, m, s in got]
    for rank, mean, spread in got:
        assert torch.allclose(mean.double(), ref_mean, rtol=1e-6)
        assert torch.allclose(spread.double(), ref_spread, rtol=5e-3)
    assert torch.equal(got[0][1], got[1][1])


def test_perturbed_members_are_seeded_and_small():
    x0 = torch.ones(3, 8, 16) * 100.0
    std = torch.tensor([1.0, 10.0, 100.0])
    a, b = perturbed_member(x0, std, 3), perturbed_member(x0, std, 3)
    assert torch.equal(a, b) and torch.equal(perturbed_member(x0, std, 0), x0)
    d = (perturbed_member(x0, std, 4) - x0).flatten(1).std(1)
    assert torch.allclose(d, 1e-3 * std, rtol=0.3)


def _ens_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0 = torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6)
        ens = MemberParallelEnsemble(lambda x: 0.5 * x + 1.0, 5, torch.tensor([1.0, 2.0]), perturb_scale=0.1)
        out = ens.run(x0, 2, gather=True)
        # per saved step (BASELINE configs[4]): both reduction forms agree, the gathered members of step 1 are the rolled ICs
        saved = list(ens.steps(x0, 2, save_every=1, gather=True, how="allreduce"))
        assert [s_["step"] for s_ in saved] == [1, 2] and torch.allclose(saved[1]["mean"], out["mean"], rtol=1e-6)
        assert torch.allclose(saved[1]["spread"], out["spread"], rtol=1e-4, atol=1e-6) and torch.equal(saved[1]["members"], out["members"])
        assert saved[0]["members"].shape == (5, 2, 4, 6)
        q.put((rank, out["mean"].numpy(), out["spread"].numpy(), out["members"].numpy(), out["local_members"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_member_parallel_ensemble_two_ranks_equals_one_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ens_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    x0 = torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6)
    single = MemberParallelEnsemble(lambda x: 0.5 * x + 1.0, 5, torch.tensor([1.0, 2.0]), perturb_scale=0.1).run(x0, 2, gather=True)
    assert got[0][4] == [0, 2, 4] and got[1][4] == [1, 3]
    for rank, mean, spread, members, _ in got:
        mean, spread, members = torch.from_numpy(mean), torch.from_numpy(spread), torch.from_numpy(members)
        assert torch.allclose(mean, single["mean"], rtol=1e-6)
        assert torch.allclose(spread, single["spread"], rtol=1e-4, atol=1e-6)
        assert torch.equal(members, single["members"])
