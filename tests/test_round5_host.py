"""Round-5 host-side pieces that need no GPU: the compact bench line, GraphCast's per-stage FLOP accounting, the sigma-unit error of the Pangu
oracle, and the out-of-distribution state generators of the numerics tests."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def test_compact_line_keeps_the_contract_and_fits_3_kb(tmp_path, monkeypatch, capsys):
    import bench
    full = json.loads((ROOT / "profiles" / "r05_bench_pangu.json").read_text())
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= 3000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert line["config"]["workload"] and "model" not in line["config"]
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["parity"]["full_size"]["grid"] == "721x1440" and set(line["models"]) == {"sfno", "graphcast"}
    # a multi-GPU line has no CPU baseline / parity / models: the compact form must not invent them
    slim = {k: v for k, v in full.items() if k not in ("cpu_baseline", "parity", "models", "pcie_inclusive", "predict_inclusive")}
    assert not ({"cpu_baseline", "parity", "models"} & set(bench.compact_line(slim)))
    # emit(): the LAST stdout line is the compact JSON; the full record goes next to bench.py
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(tmp_path / "bench.py"))
    bench.emit(full)
    out = capsys.readouterr().out.strip().splitlines()
    assert json.loads(out[-1])["detail"] == "bench_detail.json" and len(out[-1]) <= 3000
    assert json.loads((tmp_path / "bench_detail.json").read_text())["ms_per_step"] == full["ms_per_step"]


def test_graphcast_flops_per_stage_add_up_to_the_step_totals():
    from skyrim_amd.graphcast.spec import GraphcastConfig, flops_per_stage, flops_per_step, flops_per_step_executed
    cfg = GraphcastConfig()
    counts = (cfg, 721 * 1440, 40962, 327660, 1618746, 3 * 721 * 1440)
    pub, exe = flops_per_stage(*counts), flops_per_stage(*counts, executed=True)
    assert set(pub) == set(exe) == {"embed", "encoder", "processor", "decoder", "output"}
    assert abs(sum(pub.values()) - flops_per_step(*counts)) < 1e3 and abs(sum(exe.values()) - flops_per_step_executed(*counts)) < 1e3
    assert 25e12 < sum(pub.values()) < 27e12 and 14e12 < sum(exe.values()) < 16e12
    assert pub["embed"] == exe["embed"] and pub["output"] == exe["output"]          # no edge MLP in them: nothing to take apart
    for k in ("encoder", "processor", "decoder"):
        assert exe[k] < pub[k]
    assert abs(pub["processor"] / 16 - 751.6e9) / 751.6e9 < 0.01                     # the per-layer figure of VERDICT r4


def test_sigma_error_does_not_flatter_offset_channels():
    from oracle import pangu_oracle as O
    std = torch.tensor([1.3e3, 1.0])
    ref = torch.stack([torch.full((4, 8), 1.0e5), torch.zeros(4, 8)])               # "msl": mean 1e5, sigma 1.3e3; a zero-mean channel
    ref[0, 0, 0] += 1.3e3
    ref[1, 0, 0] = 1.0
    y = ref.clone()
    y[0, 1, 1] += 13.0                                                               # 1 % of sigma
    y[1, 1, 1] += 0.01
    rel, sig = O.per_channel_rel_err(y, ref), O.per_channel_sigma_err(y, ref, std)
    assert abs(rel[0].item() - 13.0 / 101300.0) < 1e-9 and abs(sig[0].item() - 0.01) < 1e-6      # 1.3e-4 against 1e-2: the offset flatters the first
    assert abs(rel[1].item() - 0.01) < 1e-9 and abs(sig[1].item() - 0.01) < 1e-9


def test_out_of_distribution_states_are_what_their_names_say():
    import _states
    from skyrim_amd.pangu.spec import PanguGeometry, channel_stats, smooth_noise
    g = PanguGeometry(49, 192)
    mean, std = channel_stats()
    rough = {}
    for kind in _states.KINDS:
        u = _states.unit_field(g, kind)
        assert u.shape == (69, 49, 192) and torch.isfinite(u).all()
        assert u.flatten(1).mean(1).abs().max() < 1e-4 and (u.flatten(1).std(1) - 1).abs().max() < 1e-3
        rough[kind] = (u[:, :, 1:] - u[:, :, :-1]).std().item()                      # east-west roughness
        x = _states.state(g, kind)
        assert torch.allclose(x.flatten(1).mean(1), mean, rtol=1e-3, atol=1e-3 * float(std.max()))
    base = smooth_noise(g, 0)
    rough["calibration"] = (base[:, :, 1:] - base[:, :, :-1]).std().item()
    assert rough["smooth3"] > 1.5 * rough["calibration"] > 1.5 * rough["smooth31"]   # rougher and smoother than the calibration family
    assert rough["powerlaw"] < rough["calibration"]
    zonal = _states.unit_field(g, "meridional").mean(2)                              # a zonal-mean profile that is most of the variance
    assert zonal.var(1).mean() > 0.5
    with pytest.raises(ValueError):
        _states.unit_field(g, "nope")


def test_gain_one_params_scale_only_embedding_and_recovery():
    import _states
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    p = init_synthetic(PanguGeometry(49, 192), 0)
    q = _states.gain_one_params(p)
    changed = sorted(k for k in p if not torch.equal(p[k], q[k]))
    assert changed and all(k.startswith(("embed.conv", "recover.conv")) and k.endswith("weight") for k in changed)
    assert torch.allclose(q["embed.conv.weight"], 3.5 * p["embed.conv.weight"])
