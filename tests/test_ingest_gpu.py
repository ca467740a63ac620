"""SURVEY.md 8 rows f2 / f3 / f4 on the HIP path ("close the (f) rows with GPU tests"):

  f2  weights INGESTED from files of the published formats -- a Pangu-shaped ``.onnx`` (+ reviewed ``.map.json``), a modulus-named
      SFNO ``weights.tar`` package directory, a haiku-keyed GraphCast ``params.npz`` directory -- named by ``SKYRIM_*_WEIGHTS``, run
      through the reference API on the GPU and compared with the oracle on the SAME tensors
      (/root/reference/skyrim/core/models/pangu.py:45-46, fourcastnet_v2.py:36-37, graphcast.py:51-54);
  f3  the ``forecast`` CLI -> HIP path -> files -> values, and a restart from the saved file
      (/root/reference/skyrim/forecast.py:19-56, tests/core/test_skyrim.py:6-10 is the reference's own bar; utils.py:24-27);
  f4  ``Skyrim("pangu", "fourcastnet_v2")``: the multi-model mean over the common channels on real engines
      (/root/reference/skyrim/core/models/ensemble.py:51-67), and the whole-grid wind speed on the device.

The files are written here from ``init_synthetic`` (no checkpoint is obtainable in this environment); what is tested is the path
file -> loader -> slot layout -> prepared planes -> kernels."""
import datetime
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T0 = datetime.datetime(2024, 5, 13, 18, 0)


def test_pangu_onnx_file_through_the_reference_api(tmp_path, monkeypatch):
    from oracle import pangu_oracle as O
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    from tests.test_onnx_weights import pangu_like_onnx
    g = PanguGeometry(49, 192)
    params = init_synthetic(g, 3)
    path = tmp_path / "pangu_weather_6.onnx"
    truth = pangu_like_onnx(path, g, params)
    mapping = {s: [n, "T" if n.startswith("onnx::MatMul") else "id"] for s, n in truth.items()}
    Path(str(path) + ".map.json").write_text(json.dumps({"mapping": mapping}))
    monkeypatch.setenv("SKYRIM_PANGU_WEIGHTS", str(path))
    monkeypatch.delenv("SKYRIM_SYNTHETIC_WEIGHTS", raising=False)              # the file, not the seeded stand-in
    m = PanguModel(ic_source="synthetic", geom=g)
    pred = m.predict_one_step(T0)
    ic = torch.from_numpy(np.array(pred.values[0]))
    err = O.per_channel_rel_err(torch.from_numpy(np.array(pred.values[1])), O.forward(params, ic)).max().item()
    assert err < 7e-4, err
    # the load-time guard judged the default plan on THESE weights and the loop records what it runs with (PanguEngine._guard)
    assert m.model.term_plan == 0x66F and m.model.engine.guard_report and m.model.engine.guard_report[-1][1] < 5e-4, m.model.engine.guard_report
    # without the reviewed mapping the loader refuses (the automatic shape-and-order mapping is never applied silently)
    Path(str(path) + ".map.json").unlink()
    with pytest.raises(ValueError, match="no slot mapping given"):
        PanguModel(ic_source="synthetic", geom=g)


def test_sfno_package_directory_through_the_reference_api(tmp_path, monkeypatch):
    from oracle import sfno_oracle as S
    from skyrim_amd.core.models.fourcastnet_v2 import FourcastnetV2Model
    from skyrim_amd.sfno.spec import SfnoConfig, init_synthetic
    cfg = SfnoConfig(n_lat=33, n_lon=64, in_chans=5, out_chans=5, embed_dim=16, num_layers=3, scale_factor=2)
    p = init_synthetic(cfg, 0)
    sd = {"module.encoder.0.weight": p["encoder.fc1.weight"][:, :, None, None], "module.encoder.0.bias": p["encoder.fc1.bias"],
          "module.encoder.2.weight": p["encoder.fc2.weight"][:, :, None, None], "module.pos_embed": p["pos_embed"][None],
          "module.decoder.0.weight": p["decoder.fc1.weight"][:, :, None, None], "module.decoder.0.bias": p["decoder.fc1.bias"],
          "module.decoder.2.weight": p["decoder.fc2.weight"][:, :, None, None]}
    for i in range(cfg.num_layers):
        b = f"module.blocks.{i}."
        sd.update({b + "norm0.weight": p[f"blocks.{i}.norm0.weight"], b + "norm0.bias": p[f"blocks.{i}.norm0.bias"],
                   b + "filter.filter.weight": torch.view_as_complex(p[f"blocks.{i}.filter.weight"].contiguous()),
                   b + "inner_skip.weight": p[f"blocks.{i}.inner_skip.weight"][:, :, None, None], b + "inner_skip.bias": p[f"blocks.{i}.inner_skip.bias"],
                   b + "norm1.weight": p[f"blocks.{i}.norm1.weight"], b + "norm1.bias": p[f"blocks.{i}.norm1.bias"],
                   b + "mlp.fwd.0.weight": p[f"blocks.{i}.mlp.fc1.weight"][:, :, None, None], b + "mlp.fwd.0.bias": p[f"blocks.{i}.mlp.fc1.bias"],
                   b + "mlp.fwd.2.weight": p[f"blocks.{i}.mlp.fc2.weight"][:, :, None, None], b + "mlp.fwd.2.bias": p[f"blocks.{i}.mlp.fc2.bias"]})
    pkg = tmp_path / "fcnv2_sm"
    pkg.mkdir()
    torch.save({"model_state": sd}, pkg / "weights.tar")
    np.save(pkg / "global_means.npy", p["norm.mean"].reshape(1, -1, 1, 1).numpy())
    np.save(pkg / "global_stds.npy", p["norm.std"].reshape(1, -1, 1, 1).numpy())
    monkeypatch.setenv("SKYRIM_SFNO_WEIGHTS", str(pkg))
    monkeypatch.delenv("SKYRIM_SYNTHETIC_WEIGHTS", raising=False)
    m = FourcastnetV2Model(ic_source="synthetic", cfg=cfg)
    pred = m.predict_one_step(T0)
    ic = torch.from_numpy(np.array(pred.values[0]))
    err = S.per_channel_rel_err(torch.from_numpy(np.array(pred.values[1])), S.forward(p, ic, cfg)).max().item()
    assert err < 1e-4, err


def test_graphcast_haiku_directory_through_the_reference_api(tmp_path, monkeypatch):
    from oracle import graphcast_graph as GG
    from oracle import graphcast_oracle as G
    from skyrim_amd.core.models.base import GlobalModel
    from skyrim_amd.core.models.graphcast import GraphcastModel
    from skyrim_amd.graphcast import checkpoint as GC
    from skyrim_amd.graphcast.spec import GraphcastConfig, init_synthetic, param_spec
    cfg = GraphcastConfig(n_lat=33, n_lon=64, splits=2, latent=32, steps=3)
    p = init_synthetic(cfg, 0)
    hk = {}
    for mlp in sorted({s.rsplit(".", 2)[0] for s, _ in param_spec(cfg) if s.endswith(".fc1.weight")}):
        for part, key in GC.haiku_keys(mlp).items():
            slot = f"{mlp}.{part}"
            if slot in p:
                hk[key] = (p[slot].T if (part.endswith("weight") and p[slot].dim() == 2) else p[slot]).numpy()
    # deepmind's mesh-node embedder sees [zeros(grid feature width) | 3 structural features]
    hk[GC.haiku_keys("embed.mesh")["fc1.weight"]] = np.concatenate([np.zeros((cfg.grid_in - 3, cfg.latent), np.float32), p["embed.mesh.fc1.weight"].T.numpy()])
    pkg = tmp_path / "graphcast_operational"
    pkg.mkdir()
    np.savez(pkg / "params.npz", model_config=np.zeros(1), **{"params:" + ":".join(k.rsplit("/", 1)): v for k, v in hk.items()})
    np.savez(pkg / "stats.npz", mean=p["norm.mean"].numpy(), std=p["norm.std"].numpy(), diff_std=p["norm.diff_std"].numpy(), static=p["static"].numpy())
    monkeypatch.setenv("SKYRIM_GRAPHCAST_WEIGHTS", str(pkg))
    monkeypatch.delenv("SKYRIM_SYNTHETIC_WEIGHTS", raising=False)
    from skyrim_amd.datasource import get_initial_condition_for_model
    m = GraphcastModel(ic_source="synthetic", cfg=cfg)
    da = GlobalModel.forecast(m, T0, n_steps=1)                                    # the generic TimeLoop drive: two history levels in, one step out
    x = get_initial_condition_for_model(m.model, m.data_source, T0)[0].cpu()       # (2, C, lat, lon): states at t - 6 h and t
    x0, x1 = x[0], x[1]
    f = m.model.forcing(T0).float().cpu()
    ref = G.forward(p, GG.build(cfg.n_lat, cfg.n_lon, cfg.splits), x0, x1, f)
    got = torch.from_numpy(np.array(da.values[-1]))
    assert np.array_equal(np.array(da.values[0]), x1.numpy())
    assert G.increment_rel_err(got, ref, x1).max().item() < 1e-3


def test_forecast_cli_on_the_hip_path_and_restart_from_its_file(tmp_path, monkeypatch):
    """``forecast -m pangu -ic gfs -l 12 -o DIR`` (click; the tests' SKYRIM_SYNTHETIC_IC=1 puts the seeded stand-in behind "gfs") -> PanguModel on the GPU -> two netCDF files; values against the oracle;
    then ``predict_one_step`` restarted from the second file on a FRESH model (reference utils.py:24-27)."""
    from click.testing import CliRunner
    from oracle import pangu_oracle as O
    from skyrim_amd import forecast as cli
    from skyrim_amd.core.models import MODELS
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.labeled import open_dataarray
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    g = PanguGeometry(49, 192)
    params = init_synthetic(g, 0)

    class ToyPangu(PanguModel):
        def __init__(self, *a, **kw):
            super().__init__(*a, geom=g, params=params, **kw)

    monkeypatch.setitem(MODELS, "pangu", ToyPangu)
    res = CliRunner().invoke(cli.main, ["-m", "pangu", "-ic", "gfs", "-d", "20240513", "-t", "1800", "-l", "12", "-o", str(tmp_path)])
    assert res.exit_code == 0, res.output + repr(res.exception)
    files = sorted(tmp_path.rglob("*.nc"), key=lambda f: f.name.split("__")[2])          # by start time ("file" sorts before "synthetic")
    assert [f.name.split("__")[0] for f in files] == ["pangu", "pangu"] and files[0].name.endswith("20240513_18:00__20240514_00:00.nc")
    first, second = open_dataarray(str(files[0])), open_dataarray(str(files[1]))
    ic = torch.from_numpy(np.array(first.values[0]))
    want = O.rollout(params, ic, 3)
    assert O.per_channel_rel_err(torch.from_numpy(np.array(first.values[1])), want[0]).max().item() < 7e-4
    assert O.per_channel_rel_err(torch.from_numpy(np.array(second.values[1])), want[1]).max().item() < 1e-3
    assert files[0].name.startswith("pangu__synthetic__") and files[1].name.startswith("pangu__file__")      # never stamped "gfs"
    fresh = ToyPangu(ic_source="synthetic")
    nxt = fresh.predict_one_step(T0 + datetime.timedelta(hours=12), initial_condition=files[1])
    assert fresh.model.io_counters == {"state_uploads": 1, "resident_hits": 0}
    assert O.per_channel_rel_err(torch.from_numpy(np.array(nxt.values[1])), want[2]).max().item() < 1e-3


def test_multi_model_mean_on_real_engines_and_device_wind_speed():
    """``Skyrim("pangu", "fourcastnet_v2").predict`` = mean over the models of their common channels (ensemble.py:51-67), each model on
    its HIP engine; and ``GlobalPrediction.wind_speed_field`` on the GPU against the host form."""
    from skyrim_amd.core import Skyrim
    from skyrim_amd.core.models import MODELS
    from skyrim_amd.core.models.base import GlobalPrediction
    from skyrim_amd.core.models.fourcastnet_v2 import FourcastnetV2Model
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.pangu.spec import PanguGeometry
    from skyrim_amd.sfno.spec import SfnoConfig
    import pytest as _pytest
    g = PanguGeometry(49, 192)
    cfg = SfnoConfig(n_lat=49, n_lon=192, embed_dim=32, num_layers=3, scale_factor=2)        # 73 channels on Pangu's toy grid

    class P(PanguModel):
        def __init__(self, *a, **kw):
            super().__init__(*a, geom=g, **kw)

    class F(FourcastnetV2Model):
        def __init__(self, *a, **kw):
            super().__init__(*a, cfg=cfg, **kw)

    mp = _pytest.MonkeyPatch()
    try:
        mp.setitem(MODELS, "pangu", P)
        mp.setitem(MODELS, "fourcastnet_v2", F)
        s = Skyrim("pangu", "fourcastnet_v2", ic_source="synthetic")
        pred, _ = s.predict("20240513", "1800", lead_time=6, save=False)
        a, _ = P(ic_source="synthetic").rollout(T0, n_steps=1, save=False)
        b, _ = F(ic_source="synthetic").rollout(T0, n_steps=1, save=False)
    finally:
        mp.undo()
    common = [c for c in a.channel.values.tolist() if c in set(b.channel.values.tolist())]
    assert len(common) > 40 and pred.prediction.channel.values.tolist() == common
    want = 0.5 * (a.sel(channel=common).values + b.sel(channel=common).values)
    assert pred.prediction.values.shape == want.shape and np.allclose(pred.prediction.values, want, rtol=1e-6, atol=1e-6)
    gp = GlobalPrediction(a, model_name="pangu")
    host = gp.wind_speed_field(1000, n_step=1)
    dev = gp.wind_speed_field(1000, n_step=1, device="cuda:0")
    assert dev.is_cuda and np.allclose(np.asarray(host), dev.cpu().numpy(), rtol=1e-6)


def test_ensemble_rollout_on_real_engines_writes_the_mean_files_and_frees_each_member(tmp_path, monkeypatch):
    """``GlobalEnsemble.rollout`` to the reference's contract (ensemble.py:86-128) on the HIP engines: one member on the GPU at a time, released
    in a ``finally`` (``GlobalModel.release_model``: ``skpangu_destroy`` / arenas dropped / allocator emptied -- ``torch.cuda.memory_allocated()``
    is back at its baseline after EACH member), the per-step ensemble-mean files named ``{a_b}__{src}__{t0}__{t1}.nc`` and holding the mean of
    the members' files of that step."""
    from skyrim_amd.core.models.ensemble import GlobalEnsemble, GlobalEnsemblePrediction
    from skyrim_amd.labeled import open_dataarray
    from skyrim_amd.pangu.spec import PanguGeometry
    from skyrim_amd.sfno.spec import SfnoConfig
    g = PanguGeometry(49, 192)
    cfg = SfnoConfig(n_lat=49, n_lon=192, embed_dim=32, num_layers=3, scale_factor=2)
    ens = GlobalEnsemble(["pangu", "fourcastnet_v2"], ic_source="synthetic", model_kwargs={"pangu": dict(geom=g), "fourcastnet_v2": dict(cfg=cfg)})
    import gc
    gc.collect()                                  # engines of earlier tests still waiting for the collector would be freed mid-test by release_model's own gc
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    base, seen, peak = torch.cuda.memory_allocated(), [], []
    release = GlobalEnsemble._release_model

    def spy(self):
        peak.append(torch.cuda.memory_allocated())
        release(self)
        seen.append(torch.cuda.memory_allocated())

    monkeypatch.setattr(GlobalEnsemble, "_release_model", spy)
    mean, paths = ens.rollout(T0, n_steps=2, save=True, save_config={"output_dir": str(tmp_path)})
    assert len(seen) == 2 and all(p > base + (1 << 20) for p in peak), (base, peak)          # a member holds device memory while it runs ...
    assert all(s <= base for s in seen), (base, seen)                                            # ... and none once it is released
    assert [p.name for p in paths] == ["fourcastnet_v2_pangu__synthetic__20240513_18:00__20240514_00:00.nc",
                                       "fourcastnet_v2_pangu__file__20240514_00:00__20240514_06:00.nc"]
    assert all(p.parent == tmp_path / "fourcastnet_v2_pangu" and p.exists() for p in paths)
    for s, p in enumerate(paths):
        got, members = open_dataarray(p), [open_dataarray(q) for q in ens.member_paths[s::2]]
        common = got.channel.values.tolist()
        assert len(common) > 40 and common == ens.common_channels
        want = 0.5 * (members[0].sel(channel=common).values + members[1].sel(channel=common).values)
        assert np.allclose(got.values, want, rtol=1e-6, atol=1e-6)
    assert np.allclose(open_dataarray(paths[-1]).values, mean.values, rtol=1e-6, atol=1e-6)
    assert GlobalEnsemblePrediction(paths[-1]).prediction.shape == mean.shape


def test_bswap32_kernel_against_numpy():
    """skio_bswap32 (include/skyrim_io.h): every word byte-reversed, vector body and scalar tail, unaligned starts, in place."""
    from skyrim_amd import deliver as D
    side = torch.cuda.Stream()
    for n, off in ((1, 0), (3, 0), (4, 0), (1027, 0), (1027, 1), (1 << 20, 0), ((1 << 20) + 5, 3)):
        src = torch.randn(n + off, device="cuda")[off:]
        dst = torch.empty(n + off, device="cuda")[off:]
        side.wait_stream(torch.cuda.current_stream())
        D.bswap32(src, dst, side)
        side.synchronize()
        want = src.cpu().numpy().byteswap()
        assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32)), (n, off)
    x = torch.randn(4099, device="cuda")
    want = x.cpu().numpy().byteswap()
    D.bswap32(x, x, torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy().view(np.uint32), want.view(np.uint32))
    with pytest.raises(ValueError):
        D.bswap32(x, x[:10], torch.cuda.current_stream())
    with pytest.raises(ValueError):
        D.bswap32(x.double(), x.double(), torch.cuda.current_stream())


def _rollouts_with_and_without_the_image(m, tmp_path, monkeypatch, n_steps=3):
    """``m.rollout(save=True)`` twice from one initial condition -- host swap (SKYRIM_SAVE_BE=0), then the big-endian image -- with the toy
    payloads forced through the large-payload writers; returns {mode: (final values, paths)} after checking which writer ran."""
    from skyrim_amd import ncio
    monkeypatch.setattr(ncio, "FAST_PAYLOAD_BYTES", 0)
    calls = {"image": 0, "host": 0}
    image_w, host_w = ncio._image_payload_write, ncio._parallel_payload_write
    monkeypatch.setattr(ncio, "_image_payload_write", lambda *a, **k: (calls.__setitem__("image", calls["image"] + 1), image_w(*a, **k))[1])
    monkeypatch.setattr(ncio, "_parallel_payload_write", lambda *a, **k: (calls.__setitem__("host", calls["host"] + 1), host_w(*a, **k))[1])
    ic = m.predict_one_step(T0)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SKYRIM_SAVE_BE", mode)
        before = dict(calls)
        pred, paths = m.rollout(T0, n_steps=n_steps, save=True, save_config={"output_dir": str(tmp_path / mode), "forecast_id": "x"}, initial_condition=ic)
        out[mode] = (np.array(pred.values), [Path(p) for p in paths])
        assert (calls["image"] - before["image"], calls["host"] - before["host"]) == ((n_steps, 0) if mode == "1" else (0, n_steps))
        image = pred.__dict__["_image"]
        assert (image is not None) == (mode == "1")
        if image is not None:                        # steps after the first borrow the state they start from out of the previous image
            assert len(image.parts) == 2 and [p.array.shape[0] for p in image.parts] == [1, 1]
            assert np.array_equal(np.asarray(image.array, dtype=np.float32), pred.values)
    monkeypatch.delenv("SKYRIM_SAVE_BE")
    assert np.array_equal(out["0"][0], out["1"][0])
    for a, b in zip(out["0"][1], out["1"][1]):
        assert a.name == b.name and a.read_bytes() == b.read_bytes()
    return ic


def test_saving_rollout_delivers_the_bytes_of_its_files(tmp_path, monkeypatch):
    """``rollout(save=True)`` (reference base.py:134-143 -> common.py:144): the per-step netCDF files written from the big-endian image
    produced in HBM are byte-identical to the files of the host-swapping path (SKYRIM_SAVE_BE=0); intermediate steps bring only the
    image of the NEW state to the host (the state they start from is the previous image's), the state stays resident, and a reader of
    their ``values`` still gets the numbers."""
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.core.models.utils import run_basic_inference
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    g = PanguGeometry(49, 192)
    m = PanguModel(ic_source="synthetic", geom=g, params=init_synthetic(g, 0))
    ic = _rollouts_with_and_without_the_image(m, tmp_path, monkeypatch)
    # an intermediate step as the rollout asks for it: nothing but the image crosses to the host, the numbers appear on first read
    both = run_basic_inference(m.model, 1, m.data_source, T0, ic, deliver="both")
    only = run_basic_inference(m.model, 1, m.data_source, T0, ic, deliver="be")
    assert len(only.__dict__["_image"].parts) == 1                                     # (``ic`` was not resident: the whole pair came over)
    uploads = m.model.io_counters["state_uploads"]
    assert only.__dict__["_ready"] is not None and only.__dict__["_image"] is not None
    nxt = run_basic_inference(m.model, 1, m.data_source, T0, only, deliver="be")      # fed back without a read: resident, no upload
    assert m.model.io_counters["state_uploads"] == uploads and only.__dict__["_ready"] is not None
    parts = nxt.__dict__["_image"].parts
    assert len(parts) == 2 and parts[0].array.base is not None and np.shares_memory(parts[0].array, only.__dict__["_image"].parts[-1].array)
    assert np.array_equal(only.values, both.values) and only.__dict__["_ready"] is None
    assert np.array_equal(np.asarray(both.__dict__["_image"].array, dtype=np.float32), both.values)
    ref = run_basic_inference(m.model, 1, m.data_source, T0, both)
    assert np.array_equal(nxt.values, ref.values) and np.array_equal(nxt.values[0], only.values[1])


def test_saving_rollout_image_on_the_sfno_wrapper_and_graphcast_keeps_its_own_loop(tmp_path, monkeypatch):
    """The same byte-for-byte comparison through FourcastnetV2Model.  GraphcastModel drives its stepper itself, as the reference's wrapper
    does (graphcast.py:93-142: own ``_predict_one_step`` / ``rollout``, synchronous saves): no image there, and the public
    ``predict_one_step`` of the base class still reaches its TimeLoop."""
    from skyrim_amd.core.models.fourcastnet_v2 import FourcastnetV2Model
    from skyrim_amd.core.models.graphcast import GraphcastModel
    from skyrim_amd.graphcast.spec import GraphcastConfig
    from skyrim_amd.sfno.spec import SfnoConfig
    sfno = FourcastnetV2Model(ic_source="synthetic", cfg=SfnoConfig(n_lat=49, n_lon=192, embed_dim=32, num_layers=3, scale_factor=2))
    _rollouts_with_and_without_the_image(sfno, tmp_path / "sfno", monkeypatch)
    gc = GraphcastModel(ic_source="synthetic", cfg=GraphcastConfig(n_lat=33, n_lon=64, splits=2, latent=32, steps=2))
    one = gc.predict_one_step(T0)
    assert one.values.shape[0] == 2 and one.__dict__.get("_image") is None and np.isfinite(one.values).all()
    pred, paths = gc.rollout(T0, n_steps=2, save=True, save_config={"output_dir": str(tmp_path / "gc")})
    assert len(paths) == 2 and all(Path(p).exists() for p in paths) and pred.__dict__.get("_image") is None


def test_rollout_reads_a_steps_finite_check_with_its_copy_not_between_steps(tmp_path):
    """``rollout`` never waits for the GPU between two steps: the non-finite flag of a step travels to the host behind the state and is read
    by whoever reads the numbers.  A step that overflows still stops the rollout with the TimeLoop's FloatingPointError -- from the save
    thread (no file, no partial file for that step) or at the end of a rollout that saves nothing; ``predict_one_step`` raises at once."""
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    g = PanguGeometry(49, 192)
    m = PanguModel(ic_source="synthetic", geom=g, params=init_synthetic(g, 0))
    ic = m.predict_one_step(T0)
    good, calls = m.model.engine.step, []

    def step(x, *a, **k):
        calls.append(1)
        y = good(x, *a, **k)
        if len(calls) == 2:
            y[3, 5, 7] = float("inf")
        return y
    m.model.engine.step = step
    try:
        with pytest.raises(FloatingPointError, match="after step 1"):
            m.rollout(T0, n_steps=4, save=True, save_config={"output_dir": str(tmp_path), "forecast_id": "x"}, initial_condition=ic)
        files = sorted(p.name for p in (tmp_path / "x").iterdir())
        assert len(files) == 1 and files[0].endswith("20240513_18:00__20240514_00:00.nc"), files       # the first step's file, nothing of the second
        calls.clear()
        calls.append(1)                                                                                 # the FIRST step of this rollout overflows
        with pytest.raises(FloatingPointError, match="after step 1"):
            m.rollout(T0, n_steps=3, save=False, initial_condition=ic)
        calls.clear()
        calls.append(1)
        with pytest.raises(FloatingPointError, match="after step 1"):
            m.predict_one_step(T0, initial_condition=ic)
    finally:
        m.model.engine.step = good
    pred, _ = m.rollout(T0, n_steps=2, save=False, initial_condition=ic)                                # and the model still works
    assert np.isfinite(pred.values).all()
