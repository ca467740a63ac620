"""CPU tests of the host-side mirror of the reference API (skyrim.core / skyrim.common), driven by a
fake TimeLoop exactly like the reference's own tests/core/test_base.py (BoringModel)."""
import datetime
from pathlib import Path

import numpy as np
import pytest
import torch

from skyrim_amd import common
from skyrim_amd.core import Skyrim
from skyrim_amd.core import models as models_pkg
from skyrim_amd.core.models.base import GlobalModel, GlobalPrediction, GlobalPredictionRollout, adjust_lead_time
from skyrim_amd.core.models.ensemble import GlobalEnsemble
from skyrim_amd.core.models.utils import perturb_initial_conditions, run_basic_inference
from skyrim_amd.datasource import SyntheticDataSource, get_data_source
from skyrim_amd.labeled import DataArray, open_dataarray
from skyrim_amd.pangu.spec import CHANNELS, PanguGeometry

GEOM = PanguGeometry(9, 96)


class BoringTimeLoop:
    """CPU stand-in with the TimeLoop protocol: each 6-h step adds 1 to every field."""
    n_history_levels = 1
    time_step = datetime.timedelta(hours=6)
    device = torch.device("cpu")

    def __init__(self, channels):
        self.in_channel_names = list(channels)
        self.out_channel_names = list(channels)
        self.geom = GEOM
        self.grid = GEOM

    def __call__(self, time, x, restart=None):
        assert x.shape == (1, 1, len(self.in_channel_names), GEOM.n_lat, GEOM.n_lon)
        state = x[:, 0].clone()
        yield time, state, restart
        while True:
            state = state + 1.0
            time = time + self.time_step
            yield time, state, restart


class BoringGlobalModel(GlobalModel):
    model_name = "boring"
    channels = ["u1000", "v1000", "t2m"]

    def __init__(self, *args, **kwargs):
        super().__init__(self.model_name, *args, **kwargs)

    def build_model(self):
        return BoringTimeLoop(self.channels)

    @property
    def time_step(self):
        return self.model.time_step

    @property
    def in_channel_names(self):
        return self.model.in_channel_names

    @property
    def out_channel_names(self):
        return self.model.out_channel_names


class OtherBoringModel(BoringGlobalModel):
    model_name = "other"
    channels = ["t2m", "u1000", "msl"]


@pytest.fixture()
def boring_registry(monkeypatch):
    monkeypatch.setitem(models_pkg.MODELS, "boring", BoringGlobalModel)
    monkeypatch.setitem(models_pkg.MODELS, "other", OtherBoringModel)


T0 = datetime.datetime(2024, 5, 13, 18, 0)


@pytest.mark.parametrize("lead,want", [(0, 6), (5, 6), (6, 6), (13, 12), (24, 24), (167, 162)])
def test_adjust_lead_time(lead, want):
    assert adjust_lead_time(lead, 6) == want          # base.py:13-15


def test_ic_source_selection_and_validation():
    m = BoringGlobalModel(ic_source="ifs")           # reference tests/core/test_base.py:24-27 (source by name)
    assert isinstance(m.data_source, SyntheticDataSource) and m.data_source.channel_names == m.in_channel_names
    with pytest.raises(ValueError):
        BoringGlobalModel(ic_source="nope")
    with pytest.raises(ValueError):
        get_data_source(["t2m"], "nope")


def test_predict_one_step_layout():
    m = BoringGlobalModel(ic_source="gfs")
    da = m.predict_one_step(T0)
    assert set(da.dims) == {"time", "channel", "lat", "lon"}          # reference tests/core/test_graphcast.py:14
    assert da.dims == ("time", "channel", "lat", "lon") and da.shape == (2, 3, 9, 96)
    assert da.channel.values.tolist() == m.out_channel_names          # test_graphcast.py:22
    assert list(da.time.values) == [np.datetime64(T0, "ns"), np.datetime64(T0 + datetime.timedelta(hours=6), "ns")]
    assert da.lat.values[0] == 90.0 and da.lat.values[-1] == -90.0 and da.lon.values[0] == 0.0
    assert np.allclose(da.values[1], da.values[0] + 1.0)              # entry 0 echoes the initial state
    assert da.values.dtype == np.float32


def test_rollout_saves_one_file_per_step(tmp_path):
    m = BoringGlobalModel(ic_source="gfs")
    cfg = {"output_dir": str(tmp_path), "file_type": "netcdf"}
    pred, paths = m.rollout(T0, n_steps=3, save=True, save_config=cfg)
    assert len(paths) == 3 and "forecast_id" in cfg                   # rollout mutates save_config (base.py:129-130)
    names = [Path(p).name for p in paths]
    assert names[0] == "boring__synthetic__20240513_18:00__20240514_00:00.nc"      # never stamped with a source that was not read
    assert names[1] == "boring__file__20240514_00:00__20240514_06:00.nc"      # source flips to "file" (base.py:144)
    assert all(Path(p).parent.name == cfg["forecast_id"] for p in paths)
    assert pred.shape == (2, 3, 9, 96)
    first = m.predict_one_step(T0)
    assert np.allclose(pred.values[1], first.values[0] + 3.0)         # state fed back step to step
    back = open_dataarray(paths[-1])
    assert back.shape == (2, 3, 9, 96) and np.allclose(back.values, pred.values)
    # restart from a saved step (utils.py:24-27): path as initial condition
    again = run_basic_inference(m.model, 1, m.data_source, T0, x=paths[-1])
    assert np.allclose(again.values[0], pred.values[1])
    roll = GlobalPredictionRollout(paths)
    assert len(roll.surface_wind_speed(10.0, 20.0)) == 3


def test_forecast_keeps_every_step_and_selects_channels():
    m = BoringGlobalModel(ic_source="cds")
    da = m.forecast(T0, n_steps=4, channels=["t2m"])
    assert da.shape == (5, 1, 9, 96) and da.channel.values.tolist() == ["t2m"]
    assert np.allclose(da.values[4] - da.values[0], 4.0)


def test_skyrim_facade(boring_registry, tmp_path):
    assert "pangu" in Skyrim.list_available_models()
    with pytest.raises(ValueError, match=r"Invalid model name\(s\): \['nope'\]"):
        Skyrim("nope")                                                # skyrim.py:20-22
    s = Skyrim("boring", ic_source="gfs")
    pred, paths = s.predict("20240513", "1800", lead_time=13, save=True, save_config={"output_dir": str(tmp_path), "file_type": "netcdf"})
    assert isinstance(pred, GlobalPrediction) and len(paths) == 2     # 13 h -> 12 h -> 2 steps
    assert pred.prediction.shape == (2, 3, 9, 96)
    pred2, paths2 = s.predict("20240513", "1800", lead_time=6)
    assert paths2 == [] and pred2.prediction.time.values[-1] == np.datetime64("2024-05-14T00:00")
    fc = s.forecast(datetime.datetime(2024, 5, 13, 18, 0, 33, 7), n_steps=2)
    assert fc.shape == (3, 3, 9, 96) and fc.time.values[0] == np.datetime64("2024-05-13T18:00")


def test_global_prediction_accessors():
    m = BoringGlobalModel(ic_source="cds")
    gp = GlobalPrediction(m.predict_one_step(T0), model_name="boring")
    v = gp.prediction.values
    assert gp.point(90.0, 0.0, "t2m", n_step=1) == pytest.approx(v[1, 2, 0, 0])
    assert gp.point(21.0, -3.75, "t2m", n_step=0) == pytest.approx(v[0, 2, 3, 95])      # negative lon wraps, nearest lat
    u, w = gp.point_wind_uv(0.0, 180.0, 1000)
    assert gp.wind_speed(0.0, 180.0, 1000) == pytest.approx((u ** 2 + w ** 2) ** 0.5)
    assert gp.surface_wind_speed(0.0, 180.0) == gp.wind_speed(0.0, 180.0, 1000)
    assert gp.slice(channel="u1000").shape == (2, 9, 96)
    assert gp.slice(lat=slice(90, 0), lon=slice(0, 90)).shape[2:] == (5, 25)
    with pytest.raises(ValueError):
        GlobalPrediction(123)


def test_save_forecast_local_netcdf_and_zarr(tmp_path):
    # mirrors reference tests/test_common.py:31-53 (path parts + shape round trip)
    m = BoringGlobalModel(ic_source="cds")
    pred = m.predict_one_step(T0)
    fid = common.generate_forecast_id()
    assert len(fid) == 10 and fid != common.generate_forecast_id()
    p = common.save_forecast(pred, "test_model", T0, T0 + datetime.timedelta(hours=6), "cds",
                             config={"forecast_id": fid, "file_type": "netcdf", "output_dir": str(tmp_path)})
    assert Path(p).exists() and fid in p and "test_model" in p and p.endswith(".nc")
    assert open_dataarray(p).shape == (2, 3, 9, 96)
    z = common.save_forecast(pred, "test_model", T0, T0, "cds", config={"forecast_id": fid + "z", "file_type": "zarr", "output_dir": str(tmp_path)})
    z2 = common.save_forecast(pred, "test_model", T0, T0, "file", config={"forecast_id": fid + "z", "file_type": "zarr", "output_dir": str(tmp_path)})
    assert z == z2 and (Path(z) / ".zmetadata").exists()
    assert open_dataarray(z).shape == (4, 3, 9, 96)                   # appended along time
    only = common.save_forecast(pred, "m", T0, T0, "cds", config={"output_dir": str(tmp_path), "filter_vars": ["t2m"]})
    assert only.endswith(".nc") and open_dataarray(only).shape == (2, 1, 9, 96)   # default file_type stays netcdf locally
    with pytest.raises(ValueError):
        common.save_forecast(pred, "m", T0, T0, "cds", config={"output_dir": str(tmp_path), "file_type": "grib"})
    with pytest.raises(NotImplementedError):
        common.save_forecast(pred, "m", T0, T0, "cds", config={"output_dir": "s3://bucket/x"})
    assert common.generate_filename("pangu", T0, T0 + datetime.timedelta(hours=6), "gfs") == "pangu__gfs__20240513_18:00__20240514_00:00.nc"


def test_multi_model_ensemble_mean_over_common_channels(boring_registry):
    ens = Skyrim("boring", "other", ic_source="cds").model
    assert isinstance(ens, GlobalEnsemble)
    mean, paths = ens.rollout(T0, n_steps=1, save=False)
    assert mean.channel.values.tolist() == ["u1000", "t2m"] and mean.shape == (2, 2, 9, 96) and paths == []
    with pytest.raises(ValueError):
        GlobalEnsemble(["boring", "missing"])


def test_ensemble_rollout_saves_the_per_step_mean_files_and_releases_each_member(boring_registry, tmp_path, monkeypatch):
    """The reference's contract (ensemble.py:86-128): members one at a time, each released in a ``finally``; with save=True the per-step
    ENSEMBLE-MEAN files ``{a_b}/{a_b}__{src}__{t0}__{t1}.nc`` (names sorted) are written and returned; ``GlobalEnsemblePrediction`` opens them."""
    from skyrim_amd.core.models.ensemble import GlobalEnsemblePrediction
    released = []
    monkeypatch.setattr(BoringGlobalModel, "release_model", lambda self: released.append(self.model_name))
    ens = GlobalEnsemble(["other", "boring"], ic_source="cds")
    mean, paths = ens.rollout(T0, n_steps=2, save=True, save_config={"output_dir": str(tmp_path)})
    assert released == ["other", "boring"] and ens._model is None
    assert [p.name for p in paths] == ["boring_other__synthetic__20240513_18:00__20240514_00:00.nc", "boring_other__file__20240514_00:00__20240514_06:00.nc"]
    assert all(p.parent == tmp_path / "boring_other" and p.exists() for p in paths)
    assert len(ens.member_paths) == 4 and all(Path(p).exists() for p in ens.member_paths)
    for s, p in enumerate(paths):
        got = open_dataarray(p)
        members = [open_dataarray(q) for q in ens.member_paths[s::2]]
        want = 0.5 * (members[0].sel(channel=["t2m", "u1000"]).values + members[1].sel(channel=["t2m", "u1000"]).values)
        assert got.channel.values.tolist() == ["t2m", "u1000"] and np.allclose(got.values, want)
    assert np.allclose(open_dataarray(paths[-1]).values, mean.values)
    gp = GlobalEnsemblePrediction(paths[-1])
    assert gp.point(90.0, 0.0, "t2m", n_step=1) == pytest.approx(mean.values[1, 0, 0, 0])
    # a member that fails is still released, and nothing is swallowed
    monkeypatch.setattr(OtherBoringModel, "rollout", lambda self, **kw: (_ for _ in ()).throw(RuntimeError("boom")))
    released.clear()
    with pytest.raises(RuntimeError, match="boom"):
        GlobalEnsemble(["boring", "other"], ic_source="cds").rollout(T0, n_steps=1, save=False)
    assert released == ["boring", "other"]
    with pytest.raises(ValueError, match="netcdf"):
        ens.rollout(T0, n_steps=1, save=True, save_config={"output_dir": str(tmp_path), "file_type": "zarr"})


def test_ensemble_mean_aligns_reversed_latitudes():
    """xr.concat aligns by label (reference ensemble.py:64); a member delivered south-to-north (GraphCast) is turned round, another grid refused."""
    lat = np.linspace(90, -90, 5)
    a = DataArray(np.arange(10, dtype=np.float32).reshape(1, 1, 5, 2), ["time", "channel", "lat", "lon"], dict(time=[T0], channel=["t2m"], lat=lat, lon=[0.0, 180.0]))
    b = DataArray(a.values[:, :, ::-1].copy(), ["time", "channel", "lat", "lon"], dict(time=[T0], channel=["t2m"], lat=lat[::-1].copy(), lon=[0.0, 180.0]))
    m = GlobalEnsemble(["pangu"])._ensemble_predictions([a, b])
    assert np.array_equal(m.values, a.values) and np.array_equal(m.lat.values, lat)
    c = DataArray(a.values, ["time", "channel", "lat", "lon"], dict(time=[T0], channel=["t2m"], lat=lat * 0.5, lon=[0.0, 180.0]))
    with pytest.raises(ValueError, match="different lat"):
        GlobalEnsemble(["pangu"])._ensemble_predictions([a, c])


def test_estimate_pressure_hpa():
    from skyrim_amd.core.models.utils import estimate_pressure_hpa
    assert estimate_pressure_hpa(0.0) == pytest.approx(1013.25)
    assert estimate_pressure_hpa(1500.0) == pytest.approx(845.6, abs=0.2) and estimate_pressure_hpa(5500.0) == pytest.approx(505.4, abs=0.5)


def test_release_model_calls_the_time_loops_release():
    m = BoringGlobalModel(ic_source="cds")
    seen = []
    m.model.release = lambda: seen.append("released")
    m.release_model()
    assert seen == ["released"] and m.model is None


def test_perturb_initial_conditions():
    da = DataArray(np.zeros((1, 2, 9, 96), np.float32), ["time", "channel", "lat", "lon"],
                   dict(time=[T0], channel=["t2m", "msl"], lat=GEOM.lat, lon=GEOM.lon))
    perturb_initial_conditions(da, "msl", 44.0, -7.0, 5.0)
    assert da.values[0, 1, 2, 94] == 5.0 and da.values.sum() == 5.0


def test_synthetic_source_is_deterministic_and_era5_sized():
    src = SyntheticDataSource(CHANNELS, GEOM)
    a, b = src[T0], src[T0]
    assert a.shape == (69, 9, 96) and np.array_equal(a, b) and not np.array_equal(a, src[T0 + datetime.timedelta(hours=6)])
    assert 2.0e5 > a[CHANNELS.index("z500")].mean() > 4.0e4 and 330 > a[CHANNELS.index("t2m")].mean() > 230


def test_forecast_cli_mirrors_reference_options(boring_registry, tmp_path, monkeypatch):
    """/root/reference/skyrim/forecast.py:59-101: same option names and defaults; tests/core/test_skyrim.py:6-10 asserts exit code 0."""
    from click.testing import CliRunner
    from skyrim_amd import common, forecast
    monkeypatch.setattr(common, "AVAILABLE_MODELS", common.AVAILABLE_MODELS + ["boring"])
    opts = {p.name: p for p in forecast.main.params}
    assert set(opts) == {"model_name", "date", "time", "lead_time", "list_models", "initial_conditions", "output_dir", "filter_vars", "modal"}
    assert opts["model_name"].default == "pangu" and opts["lead_time"].default == 6 and opts["initial_conditions"].default == "gfs"
    assert opts["time"].default == "0000" and "-lm" in opts["list_models"].opts
    res = CliRunner().invoke(forecast.main, ["--list_models"])
    assert res.exit_code == 0 and "pangu" in res.output
    paths = forecast.run_forecast("boring", "20240513", "1800", 12, False, "gfs", str(tmp_path), "t2m")
    assert len(paths) == 2 and all(Path(p).exists() for p in paths)
    assert open_dataarray(paths[0]).channel.values.tolist() == ["t2m"]
    res = CliRunner().invoke(forecast.main, ["--modal"])
    assert res.exit_code != 0


# ---- round 2: explicit opt-ins, restart from a given state, GraphCast wrapper semantics ---------------------------------- #
def test_network_sources_refuse_silent_substitution(monkeypatch, tmp_path):
    monkeypatch.delenv("SKYRIM_SYNTHETIC_IC", raising=False)
    monkeypatch.delenv("SKYRIM_IC_DIR", raising=False)
    for src in ("gfs", "cds", "ifs"):
        with pytest.raises(RuntimeError, match="SKYRIM_SYNTHETIC_IC"):
            BoringGlobalModel(ic_source=src)
    assert BoringGlobalModel(ic_source="synthetic").source_label == "synthetic"          # the explicit name always works
    # a local archive in the saved-forecast layout stands in for the fetcher, and then the files carry the real source name
    m = BoringGlobalModel(ic_source="synthetic")
    ic = m.predict_one_step(T0)
    (tmp_path / "gfs").mkdir()
    ic.to_netcdf(tmp_path / "gfs" / (T0 + datetime.timedelta(hours=6)).strftime("%Y%m%d_%H%M.nc"))
    monkeypatch.setenv("SKYRIM_IC_DIR", str(tmp_path))
    g = BoringGlobalModel(ic_source="gfs")
    t1 = T0 + datetime.timedelta(hours=6)
    pred, paths = g.rollout(t1, n_steps=1, save=True, save_config={"output_dir": str(tmp_path / "out")})
    assert Path(paths[0]).name.startswith("boring__gfs__") and np.allclose(pred.values[0], ic.values[1])
    with pytest.raises(FileNotFoundError):
        g.predict_one_step(T0)


def test_missing_weights_raise_unless_opted_in(monkeypatch):
    from skyrim_amd import weights
    monkeypatch.delenv("SKYRIM_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.delenv("SKYRIM_X_WEIGHTS", raising=False)
    with pytest.raises(RuntimeError, match="SKYRIM_X_WEIGHTS"):
        weights.resolve("SKYRIM_X_WEIGHTS", lambda p: {"from": p}, lambda: {"synthetic": True}, "x")
    monkeypatch.setenv("SKYRIM_SYNTHETIC_WEIGHTS", "1")
    assert weights.resolve("SKYRIM_X_WEIGHTS", lambda p: {"from": p}, lambda: {"synthetic": True}, "x") == {"synthetic": True}
    monkeypatch.setenv("SKYRIM_X_WEIGHTS", "/w.pt")
    assert weights.resolve("SKYRIM_X_WEIGHTS", lambda p: {"from": p}, lambda: {"synthetic": True}, "x") == {"from": "/w.pt"}
    guard = weights.FiniteGuard("hint")
    guard.push(torch.ones(3), 1)
    guard.push(torch.tensor([1.0, float("inf")]), 2)            # step 1 was fine
    with pytest.raises(FloatingPointError, match="after step 2: hint"):
        guard.push(torch.ones(3), 3)


class OverflowingTimeLoop(BoringTimeLoop):
    """A TimeLoop whose step ``bad_step`` produces an inf, guarded the way PanguTimeLoop guards its states (deferred flag)."""

    def __init__(self, channels, bad_step):
        super().__init__(channels)
        self.bad_step = bad_step

    def __call__(self, time, x, restart=None):
        from skyrim_amd import weights
        state = x[:, 0].clone()
        yield time, state, restart
        guard, k = weights.FiniteGuard("fp16 planes overflowed"), 0
        try:
            while True:
                k += 1
                state = state + (float("inf") if k == self.bad_step else 1.0)
                time = time + self.time_step
                guard.push(state, k)
                yield time, state, restart
        finally:
            guard.check()


def test_finite_guard_fires_through_predict_one_step_rollout_and_forecast():
    """The deferred non-finite check must reach the LAST yielded state: run_basic_inference stops the generator at k == n, and with
    n = 1 (predict_one_step, rollout) the last state is the only one."""
    class M(BoringGlobalModel):
        bad = 1

        def build_model(self):
            return OverflowingTimeLoop(self.channels, self.bad)

    m = M(ic_source="synthetic")
    with pytest.raises(FloatingPointError, match="after step 1: fp16 planes overflowed"):
        m.predict_one_step(T0)
    with pytest.raises(FloatingPointError, match="after step 1"):
        m.rollout(T0, n_steps=2, save=False)
    M.bad = 3
    m3 = M(ic_source="synthetic")
    assert m3.forecast(T0, n_steps=2).shape[0] == 3                 # steps 1, 2 are fine
    with pytest.raises(FloatingPointError, match="after step 3"):
        m3.forecast(T0, n_steps=3)                                  # the final state of forecast(n) is checked too


def test_resident_state_is_tied_to_the_delivered_array():
    """core/models/utils.py ResidentState (CPU tensors stand in for the device copies): only the very array that was delivered, still
    read-only, maps back to the resident states; a copy, a writable array or too short a history do not."""
    from skyrim_amd.core.models.utils import ResidentState, perturb_initial_conditions
    host = np.zeros((2, 3, 4, 5), np.float32)
    s0, s1 = torch.zeros(1, 3, 4, 5), torch.ones(1, 3, 4, 5)
    rs = ResidentState(host, [s0, s1])
    da = DataArray(host, ["time", "channel", "lat", "lon"], dict(channel=["a", "b", "c"], lat=np.arange(4.0), lon=np.arange(5.0)))
    assert not host.flags.writeable and da.values is host
    assert rs.tensor_for(da, 1).shape == (1, 1, 3, 4, 5) and torch.equal(rs.tensor_for(da, 1)[0, 0], s1[0])
    assert torch.equal(rs.tensor_for(da, 2)[0, 0], s0[0]) and rs.tensor_for(da, 3) is None
    assert rs.tensor_for(da.copy(), 1) is None
    edited = perturb_initial_conditions(da, "b", 2.0, 3.0, 9.0)          # copies on demand instead of writing through the read-only array
    assert edited.values is not host and edited.values[-1, 1, 2, 3] == 9.0 and host.sum() == 0.0 and rs.tensor_for(edited, 1) is None


def test_a_delivered_array_waits_for_its_copy_only_when_its_numbers_are_read():
    """labeled.DataArray(ready=...): run_basic_inference hands over a pinned buffer whose device-to-host copy may still be in flight; shape,
    dims, coordinates and the resident-state lookup never wait, the first read of ``values`` (by anyone: a derived array, numpy, a
    writer) waits exactly once; assigning new values drops the wait."""
    from skyrim_amd.core.models.utils import ResidentState
    calls = []
    host = np.arange(2 * 3 * 4 * 5, dtype=np.float32).reshape(2, 3, 4, 5)
    mk = lambda: DataArray(host, ["time", "channel", "lat", "lon"], dict(channel=["a", "b", "c"], lat=np.arange(4.0), lon=np.arange(5.0)),  # noqa: E731
                           ready=lambda: calls.append(1))
    da = mk()
    rs = ResidentState(host, [torch.zeros(1, 3, 4, 5)])
    assert da.shape == (2, 3, 4, 5) and da.ndim == 4 and da.size == 120 and da.dtype == np.float32 and da.channel.tolist() == ["a", "b", "c"]
    assert "DataArray" in repr(da) and rs.tensor_for(da, 1) is not None and calls == []
    assert da.values is host and calls == [1] and da.values is host and calls == [1]
    for read in (lambda d: d.sel(channel="b"), lambda d: np.asarray(d), lambda d: d.isel(time=-1), lambda d: d.copy(), lambda d: d.mean("time"),
                 lambda d: d.assign_coords(lat=np.arange(4.0) + 1).values):
        calls.clear()
        read(mk())
        assert calls == [1], read
    calls.clear()
    d = mk()
    d.values = host.copy()
    assert d.values is not host and calls == []


def test_predict_one_step_accepts_a_pathlib_path(tmp_path):
    m = BoringGlobalModel(ic_source="synthetic")
    first, paths = m.rollout(T0, n_steps=1, save=True, save_config={"output_dir": str(tmp_path)})
    nxt = m.predict_one_step(T0 + datetime.timedelta(hours=6), initial_condition=Path(paths[0]))
    assert np.allclose(nxt.values[1], first.values[1] + 1.0)


def test_rollout_from_a_given_initial_condition_and_no_id_leak(tmp_path):
    m = BoringGlobalModel(ic_source="synthetic")
    first, paths = m.rollout(T0, n_steps=2, save=True, save_config={"output_dir": str(tmp_path)})
    t2 = T0 + datetime.timedelta(hours=12)
    # reference base.py:127 TODO: continue from a saved step (path) or from an in-memory prediction
    for ic in (paths[-1], first):
        cont, p2 = m.rollout(t2, n_steps=1, save=True, save_config={"output_dir": str(tmp_path)}, initial_condition=ic)
        assert np.allclose(cont.values[0], first.values[1]) and np.allclose(cont.values[1], first.values[1] + 1.0)
        assert Path(p2[0]).name.startswith("boring__file__")
    # no shared default: two calls without a config draw different forecast ids
    _, a = m.rollout(T0, n_steps=1, save=True, save_config={"output_dir": str(tmp_path)})
    _, b = m.rollout(T0, n_steps=1, save=True, save_config={"output_dir": str(tmp_path)})
    assert Path(a[0]).parent != Path(b[0]).parent
    import inspect
    assert inspect.signature(GlobalModel.rollout).parameters["save_config"].default is None
    assert inspect.signature(Skyrim.predict).parameters["save_config"].default is None


def test_ensemble_of_nothing_is_a_value_error():
    with pytest.raises(ValueError):
        GlobalEnsemble(["pangu"])._ensemble_predictions([])


def test_wind_speed_field_matches_point_accessor():
    m = BoringGlobalModel(ic_source="synthetic")
    gp = GlobalPrediction(m.predict_one_step(T0), model_name="boring")
    f = gp.wind_speed_field(1000, n_step=1)
    assert f.shape == (9, 96) and f[4, 48] == pytest.approx(gp.wind_speed(0.0, 180.0, 1000))


def _cpu_graphcast_model():
    """GraphcastModel over the real GraphcastTimeLoop stepper / dataset code with the HIP engine replaced by `+1 per step`."""
    from skyrim_amd.core.models.graphcast import GraphcastModel
    from skyrim_amd.graphcast import timeloop as TL
    from skyrim_amd.graphcast.spec import CHANNELS as GC, GraphcastConfig

    cfg = GraphcastConfig(n_lat=9, n_lon=16)

    class FakeEngine:
        state_shape = (83, 9, 16)
        device = torch.device("cpu")

        def step(self, prev, cur, forcing):
            assert forcing.shape == (15, 9, 16)
            return cur + 1.0

    loop = object.__new__(TL.GraphcastTimeLoop)
    loop.cfg, loop.engine = cfg, FakeEngine()
    loop.in_channel_names = loop.out_channel_names = list(GC)
    loop.grid = TL.Grid(list(np.linspace(90.0, -90.0, 9)), list(np.arange(16) * 22.5))
    loop.stepper = TL._Stepper(loop)
    lat = torch.deg2rad(torch.linspace(90.0, -90.0, 9, dtype=torch.float64))[:, None]
    loop._sin_lat, loop._cos_lat = torch.sin(lat), torch.cos(lat)
    loop._lon = torch.deg2rad(torch.arange(16, dtype=torch.float64) * 22.5)[None, :]

    class M(GraphcastModel):
        def build_model(self):
            return loop

    src = SyntheticDataSource(GC, state_fn=lambda seed: torch.arange(83.0)[:, None, None] * 100 + torch.arange(9.0)[None, :, None] + torch.zeros(83, 9, 16))
    m = M(ic_source="synthetic")
    m.data_source = src
    return m, loop, cfg


def test_graphcast_wrapper_channel_map_order_and_forecast_only_flip(tmp_path):
    """/root/reference/skyrim/core/models/graphcast.py: _to_global_da emits CHANNEL_MAP order (:29-41, q first, t2m before u10m),
    forecast flips latitude back to 90..-90 (:138), rollout does not (:163-177)."""
    from skyrim_amd.core.models.graphcast import CHANNEL_MAP
    from skyrim_amd.graphcast.spec import CHANNELS as GC
    m, loop, cfg = _cpu_graphcast_model()
    levels = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
    want = [f"{c}{l}" for _, c in CHANNEL_MAP[:6] for l in levels] + [c for _, c in CHANNEL_MAP[6:]]
    assert want[:2] == ["q50", "q100"] and want[-5:] == ["t2m", "u10m", "v10m", "msl", "tp06"] and sorted(want) == sorted(GC)
    fc = m.forecast(T0, n_steps=3)
    assert fc.dims == ("time", "channel", "lat", "lon") and fc.shape == (4, 83, 9, 16)
    assert fc.channel.values.tolist() == want
    assert fc.lat.values[0] == 90.0 and fc.lat.values[-1] == -90.0                       # flipped back
    assert list(fc.time.values) == [np.datetime64(T0 + i * datetime.timedelta(hours=6), "ns") for i in range(4)]
    # values: synthetic field = 100 * (index in CHANNELS) + row index (row 0 = 90N); each step adds 1
    z500, t2m = GC.index("z500"), GC.index("t2m")
    assert fc.sel(channel="z500").values[0, 2, 5] == 100.0 * z500 + 2 and fc.sel(channel="t2m").values[3, 7, 0] == 100.0 * t2m + 7 + 3
    sub = m.forecast(T0, n_steps=1, channels=["t2m", "q50"])
    assert sub.channel.values.tolist() == ["t2m", "q50"] and sub.shape == (2, 2, 9, 16)
    pred, paths = m.rollout(T0, n_steps=2, save=True, save_config={"output_dir": str(tmp_path)})
    assert pred.lat.values[0] == -90.0 and pred.channel.values.tolist() == want          # the stepper's ascending latitudes
    assert pred.sel(channel="z500").values[1, 0, 0] == 100.0 * z500 + 8 + 2               # row 0 = 90S = source row 8
    assert list(pred.time.values) == [np.datetime64(T0 + i * datetime.timedelta(hours=6), "ns") for i in (1, 2)]
    back = open_dataarray(paths[0])
    assert back.lat.values[0] == -90.0 and back.shape == (2, 83, 9, 16) and Path(paths[0]).name.startswith("graphcast__synthetic__")
    # restart from the saved file: channel order and latitude direction are read off its coordinates
    cont, _ = m.rollout(T0 + datetime.timedelta(hours=12), n_steps=1, save=False, initial_condition=paths[-1])
    assert np.allclose(cont.values[0], pred.values[1]) and np.allclose(cont.values[1], pred.values[1] + 1.0)


def test_graphcast_stepper_protocol_and_time_loop_protocol_agree():
    m, loop, cfg = _cpu_graphcast_model()
    x = torch.stack([torch.zeros(83, 9, 16), torch.ones(83, 9, 16)])[None]
    state = loop.stepper.initialize(x, T0)
    assert state[0] == T0 and state[2].dtype == np.uint32
    ds = state[1]
    assert ds["geopotential"].dims == ("batch", "time", "level", "lat", "lon") and ds["2m_temperature"].dims == ("batch", "time", "lat", "lon")
    assert ds["geopotential"].lat.values[0] == -90.0 and ds["geopotential"].level.values.tolist()[0] == 50
    state2, out = loop.stepper.step(state)
    assert state2[0] == T0 + datetime.timedelta(hours=6) and out.shape == (1, 83, 9, 16) and float(out.mean()) == 2.0
    it = loop(T0, x)
    t0, y0, _ = next(it)
    t1, y1, _ = next(it)
    assert t0 == T0 and float(y0.mean()) == 1.0 and t1 == state2[0] and torch.equal(y1, out)
    da = run_basic_inference(loop, 2, None, T0, x=None if False else DataArray(x[0].numpy(), ["time", "channel", "lat", "lon"],
                             dict(time=[T0 - datetime.timedelta(hours=6), T0], channel=loop.in_channel_names, lat=loop.grid.lat, lon=loop.grid.lon)))
    assert da.shape == (3, 83, 9, 16) and np.allclose(da.values[2], 3.0)


def test_graphcast_device_forcings_match_the_spec_closed_form():
    from skyrim_amd.graphcast.spec import forcings
    from skyrim_amd.graphcast.timeloop import _EPOCH
    m, loop, cfg = _cpu_graphcast_model()
    t = datetime.datetime(2024, 5, 13, 18, 0)
    want = forcings(cfg, (t - _EPOCH).total_seconds() / 3600.0)
    assert torch.allclose(loop.forcing(t), want, atol=1e-6)
