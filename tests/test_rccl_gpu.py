"""The RCCL leg of the multi-GPU paths on ONE GPU (VERDICT r2 "Missing #5"): a 1-rank ``nccl`` process group -- ``backend="nccl"`` IS RCCL on
ROCm -- through exactly the calls ``bench.py --gpus N`` and the sharded GraphCast engine make at N > 1.  With one rank every collective copies
a rank's data onto itself, so the results must equal the no-group results; what the test proves is that the RCCL branch loads, accepts the
tensors (device, dtype, contiguity, the concatenating all-gather form) and runs on the HIP stream.  No scaling claim is made here."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_ensemble_reductions_and_member_gather_over_rccl(nccl_group):
    from skyrim_amd.pangu.ensemble import MemberParallelEnsemble, ensemble_mean_spread, gather_members
    assert nccl_group.get_backend() == "nccl" and nccl_group.get_world_size() == 1
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [torch.randn(69, 49, 192, device="cuda", generator=g) for _ in range(3)]
    stack = torch.stack(xs)
    for how in ("allgather", "allreduce"):
        mean, spread = ensemble_mean_spread(xs, 3, how)
        assert torch.allclose(mean, stack.mean(0), atol=1e-6) and torch.allclose(spread, stack.std(0, unbiased=False), atol=1e-5)
    got = gather_members(xs, 3)
    assert got.shape == (3, 69, 49, 192) and torch.equal(got, stack)
    ens = MemberParallelEnsemble(lambda x: x * 0.5 + 1.0, 3, torch.ones(69), perturb_scale=1e-2)
    out = ens.run(xs[0], 2, gather=True)
    assert out["members"].shape == (3, 69, 49, 192) and torch.allclose(out["mean"], out["members"].mean(0), atol=1e-6)
    t = torch.ones(1 << 20, device="cuda")
    nccl_group.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t.sum()) == float(1 << 20)


def test_sharded_graphcast_step_through_the_default_collectives(nccl_group, monkeypatch):
    """``GraphcastEngine(shard=(0, 1))`` with SKGC_EXERCISE_COLLECTIVES=1: the all-reduce of the grid->mesh aggregate and the per-layer
    all-gather of the mesh-node latents run through torch.distributed's default group (RCCL) and change nothing."""
    from skyrim_amd.graphcast.engine import GraphcastEngine
    from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states
    cfg = GraphcastConfig(n_lat=33, n_lon=64, splits=2, latent=32, steps=3)
    p = init_synthetic(cfg, 0)
    x0, x1 = (t.cuda() for t in synthetic_states(cfg, 0))
    f = forcings(cfg, 1000.0).cuda()
    plain = GraphcastEngine(cfg, "cuda:0")
    plain.load_params(p)
    want = plain.step(x0, x1, f)
    monkeypatch.setenv("SKGC_EXERCISE_COLLECTIVES", "1")
    eng = GraphcastEngine(cfg, "cuda:0", shard=(0, 1))
    assert eng.exercise and eng.shard_mesh
    eng.load_params(p)
    eng.profiling = True
    got = eng.step(x0, x1, f)
    stats = {s["name"]: s for s in eng.profile_read()}
    assert stats["exchange"]["launches"] == 1 + cfg.steps                    # one all-reduce + one all-gather per processor layer
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
