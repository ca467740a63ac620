"""Member-parallel ensemble logic on CPU: world_size-2 gloo processes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skyrim_amd.pangu.ensemble import ensemble_mean_spread, member_shard, rollout_members


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
        q.put((rank, mean, spread))
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
    for rank, mean, spread in got:
        assert torch.allclose(mean.double(), ref_mean, rtol=1e-6)
        assert torch.allclose(spread.double(), ref_spread, rtol=5e-3)
    assert torch.equal(got[0][1], got[1][1])
