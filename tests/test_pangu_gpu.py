"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU oracle.

Tolerances (floating point; stated per the north star): per-channel max|y - ref| / max|ref| <= 1e-3.
Modes (skyrim_amd/pangu/engine.py PRECISIONS): hi/lo split GEMMs with fp16 single-term attention --
"f16x1m" (DEFAULT since round 5: term plan 0x66F -- proj / fc1 / fc2 of every block with ONE fp16 weight plane; the coarse layers 2 / 3 read the
activation operands' hi plane only, ONE MFMA term, and their QKV is one term; layers 1 / 4 keep two terms; weights rounded with error
feedback (compensated) on the built-in calibration state; 2.0 .. 3.05e-4 measured over the full-size 24-h rollout -> asserted <= 3.5e-4),
"f16x2m" (round 4's default, 0x6F: two terms everywhere; 1.4 .. 1.7e-4 -> <= 3e-4), "f16x2c" (0x66: layers 1 / 4 at three terms),
"f16x3q" (3 terms; observed ~1e-4 -> asserted <= 3e-4), "f16x3", "bf16x3" (3 terms everywhere; ~8e-5 -> <= 3e-4).  (The single-plane
"f16" probe -- slower than the default since round 5 and outside the bar -- was deleted in round 6.)  Other plans: PanguEngine(g, "f16x3q", term_plan=...).
Stage-level tests use max-abs / max-abs-ref.
"""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pangu_oracle as O
from skyrim_amd.pangu.engine import DEFAULT_PRECISION
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state

pytestmark = pytest.mark.gpu

STAGE_TOL = {"f16x1m": 1.5e-3, "f16x2m": 1.5e-3, "f16x2c": 1.5e-3, "bf16x3": 5e-4, "f16x3": 5e-4, "f16x3q": 5e-4}
# the default mode's asserted error per step (STEP_TOL[DEFAULT_PRECISION]).  Round 5: the one-term coarse layers, measured at 721x1440 over the 24-h
# rollout at 2.50 / 2.78 / 2.75 / 2.84e-4; the figure is deterministic for a given build but moves with anything that nudges the compensated rounding's
# decisions -- 2.0 .. 3.05e-4 across three choices of what the rounding is fitted on (tools/r5_x1m_full.py), 1.95e-4 (step 1) after a kernel
# clean-up that changed no arithmetic on paper -- hence 3.5e-4, three times inside the bar; toy grid 1.5 - 1.7e-4.  The two-term plan (round 4's
# default) sits at 1.4 .. 1.7e-4 and is held to 3e-4.
DEF_TOL = 3.5e-4
STEP_TOL = {"f16x1m": DEF_TOL, "f16x2m": 3e-4, "f16x2c": 3e-4, "bf16x3": 3e-4, "f16x3": 3e-4, "f16x3q": 3e-4}   # 3-term modes: 3x inside the bar


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.fixture(scope="module")
def ref(toy):
    g, params, x = toy
    taps = {}
    y = O.forward(params, x, taps=taps)
    return taps, y


# the default run: the default mode and three terms on fp16 planes; SKYRIM_TEST_ALL_MODES=1 adds the other names (bf16x3: one step below)
MODES = [DEFAULT_PRECISION, "f16x3q"] + ([m for m in STEP_TOL if m not in (DEFAULT_PRECISION, "f16x3q")] if os.environ.get("SKYRIM_TEST_ALL_MODES") else [])


@pytest.fixture(scope="module", params=MODES)
def eng(request, toy):
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    e = PanguEngine(g, request.param, "cuda:0")
    e.load_params(params)
    return e


def test_native_library_is_the_path_that_runs(eng):
    import ctypes
    assert isinstance(eng.lib, ctypes.CDLL) and "libskyrim_pangu.so" in eng.lib._name
    with open("/proc/self/maps") as f:
        assert "libskyrim_pangu.so" in f.read()


def test_patch_embed(eng, toy, ref):
    g, params, x = toy
    assert rel(eng.patch_embed(x.cuda()), ref[0]["embed"]) < STAGE_TOL[eng.precision]


@pytest.mark.parametrize("layer,i", [(1, 0), (1, 1), (2, 0), (2, 1), (2, 5), (3, 4), (4, 1)])
def test_earth_specific_block(eng, toy, ref, layer, i):
    """Window gather (pad/roll/partition), QKV, earth-specific bias + shift mask, softmax, PV, proj,
    LayerNorm, window reverse + crop, residual, MLP -- one block, plain and rolled, both resolutions."""
    g, params, x = toy
    torch.manual_seed(layer * 10 + i)
    # the block's TRUE input (the oracle's tap of the stage before it): a calibrated term plan carries biases fitted to the activations each
    # block sees in the network, so a block is tested on those (block 5 of a layer fed the layer's entry state is a different distribution)
    before = {(1, 0): "embed", (2, 0): "down", (3, 0): "layer2.block5", (4, 0): "up"}
    xin = ref[0][before.get((layer, i), f"layer{layer}.block{i - 1}")].contiguous()
    want = O.earth_block(O._block_params(params, layer, i), xin, O.Geometry(g.n_lat, g.n_lon).res(layer), O.HEADS[layer - 1], i % 2 == 1)
    got = eng.block(layer, i, xin.cuda())
    assert rel(got, want) < STAGE_TOL[eng.precision]


def test_downsample_upsample_recover(eng, toy, ref):
    taps, y = ref
    tol = STAGE_TOL[eng.precision]
    assert rel(eng.downsample(taps["layer1.block1"].cuda()), taps["down"]) < tol
    assert rel(eng.upsample(taps["layer3"].cuda()), taps["up"]) < tol
    got = eng.patch_recover(taps["layer1.block1"].cuda(), taps["layer4"].cuda())
    assert O.per_channel_rel_err(got.cpu(), y).max().item() < tol


def test_default_tolerance_is_the_default_modes():
    assert DEF_TOL == STEP_TOL[DEFAULT_PRECISION] and DEF_TOL < 1e-3


def test_full_step_per_channel(eng, toy, ref):
    g, params, x = toy
    y = eng.step(x.cuda())
    assert torch.isfinite(y).all()
    e = O.per_channel_rel_err(y.cpu(), ref[1])
    assert e.max().item() < STEP_TOL[eng.precision], e


def test_step_matches_golden_fixture(eng, toy):
    g, params, x = toy
    gold = np.load(__file__.rsplit("/", 1)[0] + "/golden/pangu_toy_49x192.npz")
    y = eng.step(x.cuda()).cpu()
    for c in range(69):
        want = torch.from_numpy(gold["step1_sub"][c])
        assert ((y[c, ::6, ::16] - want).abs().max() / torch.from_numpy(gold["step1_channel_absmax"])[c]).item() < STEP_TOL[eng.precision]


def test_rollout_4_steps_in_place(eng, toy):
    """config[1]: 24-h rollout = 4 autoregressive 6-h steps, state never leaves HBM (in-place step)."""
    g, params, x = toy
    xs = x.cuda().clone()
    xr = x
    for _ in range(4):
        eng.step(xs, xs)
        xr = O.forward(params, xr)
    e = O.per_channel_rel_err(xs.cpu(), xr)
    assert e.max().item() < 1e-3


def test_deterministic(eng, toy):
    g, params, x = toy
    a = eng.step(x.cuda())
    b = eng.step(x.cuda())
    assert torch.equal(a, b)


def test_longitude_shift_equivariance_toy(eng, toy):
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    y = eng.step(x.cuda())
    p2 = dict(params)
    p2["const_masks"] = torch.roll(params["const_masks"], 96, dims=-1)
    e2 = PanguEngine(g, eng.precision, "cuda:0")
    e2.load_params(p2)
    y2 = e2.step(torch.roll(x, 96, dims=-1).cuda())
    assert O.per_channel_rel_err(torch.roll(y2, -96, dims=-1).cpu(), y.cpu()).max().item() < 3 * STEP_TOL[eng.precision]


@pytest.mark.parametrize("grid", [(61, 192), (73, 288), (25, 96)])
def test_other_geometries_step_vs_oracle(grid):
    """Padding / window-type / down-sample parity cases the 49x192 toy and the full grid do not hit: an even coarse
    latitude count (61 -> 16 rows: no DownSample pad row), three longitude window columns at the coarse level (288),
    and the smallest grid the window shapes allow (25x96: one longitude window at the coarse level)."""
    from skyrim_amd.pangu.engine import PanguEngine
    g = PanguGeometry(*grid)
    params = init_synthetic(g, 5)
    x = synthetic_state(g, 5)
    e = PanguEngine(g, device="cuda:0")
    e.load_params(params)
    y = e.step(x.cuda())
    err = O.per_channel_rel_err(y.cpu(), O.forward(params, x))
    assert torch.isfinite(y).all() and err.max().item() < DEF_TOL, err


def test_step_before_prepare_is_an_error(toy):
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    e = PanguEngine(g, device="cuda:0")
    with pytest.raises(RuntimeError, match="not prepared"):
        e.step(x.cuda())
    e.load_params(params)
    with pytest.raises(ValueError):
        e.step(x.cuda()[:, :, :96].contiguous())


def test_profile_hooks_cover_the_step(eng, toy):
    g, params, x = toy
    eng.profile(True)
    eng.step(x.cuda())
    stats = eng.profile_read()
    eng.profile(False)
    by = {s["name"]: s for s in stats}
    # 3-term modes: projection + MLP as one kernel per block (fused_block.hip); the others: proj, fc1, fc2 as separate launches
    mlp = "proj_mlp_r1"
    assert by[mlp]["launches"] == 12 and by["attn_r0"]["launches"] == 4 and by["attn_r1"]["launches"] == 12 and by["embed"]["launches"] == 1
    # round 6: QKV runs inside the attention launch wherever the head's weights fit LDS beside the bias copies -- every layer of the default plan
    # (one weight plane at C = 384), the C = 192 layers of the hi / lo plans
    want_qkv = (0, 0) if eng.precision == DEFAULT_PRECISION else (0, 12)
    assert (by["qkv_r0"]["launches"], by["qkv_r1"]["launches"]) == want_qkv, (eng.precision, by["qkv_r0"], by["qkv_r1"])
    assert by["fc2_r1"]["launches"] == by["proj_r1"]["launches"] == (0 if mlp == "proj_mlp_r1" else 12)
    assert all(s["total_ms"] > 0 for s in stats if s["launches"])


# --------------------------------------------------------------------------------------------- #
#  BASELINE.json full size: 721 x 1440
# --------------------------------------------------------------------------------------------- #
@pytest.fixture(scope="module")
def full():
    from skyrim_amd.pangu.engine import PanguEngine
    g = PanguGeometry(721, 1440)
    params = init_synthetic(g, 0)
    x = synthetic_state(g, 0)
    e = PanguEngine(g, device="cuda:0")       # the default precision mode (bench.py's `value`)
    e.load_params(params)
    return g, params, x, e


@pytest.fixture(scope="module")
def full_ref(full):
    """BASELINE configs[1]: the oracle's 24-h rollout (4 steps) at 721x1440 -- the committed golden vectors of it (tests/golden/full_pangu.npz:
    lattice samples, whole-field maxima, cell means; tests/_golden_full.py), or, with SKYRIM_TEST_LIVE_ORACLE=1, the live host job (~1 min
    of 128 threads per step, every grid point)."""
    import _oracle_jobs
    if os.environ.get("SKYRIM_TEST_LIVE_ORACLE") == "1":
        return _oracle_jobs.PanguRollout()      # [k] = step k of O.rollout(params, x, 4); the job started in pytest_configure
    from _golden_full import FullSizeGolden
    return FullSizeGolden("pangu")


def full_err(ref, k, y):
    """max over channels of SURVEY 8(d)'s per-channel error of engine state ``y`` against step k of the oracle's rollout.  Golden vectors: the
    lattice points' figure, with the cell means of the WHOLE field held to the same tolerance by the caller (second value)."""
    if hasattr(ref, "errors"):
        e = ref.errors(k, y)
        return float(e["rel"].max()), float(e["cell"].max())
    return O.per_channel_rel_err(y.cpu(), ref[k]).max().item(), 0.0


@pytest.mark.timeout(1500)
def test_full_size_step_vs_oracle(full, full_ref):
    """The headline configuration against the CPU oracle, per channel (the north star's 1e-3 bar)."""
    g, params, x, e = full
    y = e.step(x.cuda())
    err, cell = full_err(full_ref, 0, y)
    assert torch.isfinite(y).all()
    print(f"full-size step: max per-channel rel err {err:.3e}, cell means {cell:.3e} ({DEFAULT_PRECISION}; plan {e.term_plan_in_effect:#05x}, guard {e.guard_report})")
    assert e.term_plan_in_effect == 0x66F and e.guard_report[0][1] < 5e-4, e.guard_report      # the load-time guard kept the default plan at full size
    assert err < 1e-3 and cell < 1e-3, (err, cell)
    assert err < DEF_TOL and cell < DEF_TOL, (err, cell)       # what the default mode is asserted to deliver


@pytest.mark.timeout(1500)
def test_full_size_24h_rollout_vs_oracle(full, full_ref):
    """BASELINE configs[1] as named: Pangu 24-h rollout = 4 autoregressive 6-h steps at 721x1440, each engine state fed back in
    place, against the oracle's own rollout -- every step inside the bar, with the 2x margin the default mode is chosen for."""
    g, params, x, e = full
    state = x.cuda().clone()
    errs = []
    for k in range(4):
        e.step(state, out=state)                # in place, as bench.py times it
        errs.append(full_err(full_ref, k, state))
    assert torch.isfinite(state).all()
    print("full-size 24-h rollout: max per-channel rel err per step " + " ".join(f"{a:.3e}" for a, _ in errs) + "; cell means " +
          " ".join(f"{c:.3e}" for _, c in errs) + f" ({DEFAULT_PRECISION})")
    assert max(max(a, c) for a, c in errs) < DEF_TOL, errs


@pytest.mark.timeout(1500)
def test_full_size_term_plans_vs_oracle(full, full_ref):
    """The per-layer term plans at 721x1440 over the 24-h rollout (one oracle run serves all): every plan inside the 1e-3 bar at every
    step; printed so that the choice of the default plan rests on full-size numbers."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x, _ = full
    import os
    # (plan, calibration, rounding, asserted bound at EVERY step): the two-term plan of round 4's default 3x inside the bar; three terms everywhere;
    # (the default plan 0x66F itself is test_full_size_24h_rollout_vs_oracle).  SKYRIM_TEST_ALL_PLANS=1: the rest of DESIGN.md 3's table
    plans = [(0x6F, "synthetic", "compensated", 3e-4), (0x00, "off", "nearest", 3e-4)]
    if os.environ.get("SKYRIM_TEST_ALL_PLANS"):             # the rest of the table of DESIGN.md 3 (13 s of calibration per compensated plan)
        plans += [(0x6F, "synthetic", "nearest", 1e-3), (0x66F, "synthetic", "compensated", 1e-3), (0x6F, "off", "nearest", 1e-3), (0xFF, "synthetic", "nearest", 1e-3), (0xFF, "off", "nearest", 1e-3), (0x0F, "synthetic", "nearest", 1e-3),
                  (0x66, "synthetic", "nearest", 1e-3), (0x66, "off", "nearest", 1e-3), (0xFF, "synthetic", "compensated", 1e-3), (0x66F, "synthetic", "nearest", 1e-3)]
    for plan, cal, rounding, bound in plans:
        e = PanguEngine(g, "f16x3q", "cuda:0", term_plan=plan)
        e.load_params(params, calibration=cal, rounding=rounding, guard=False)          # the plan AS GIVEN is what is measured
        state = x.cuda().clone()
        errs = []
        for k in range(4):
            e.step(state, out=state)
            errs.append(max(full_err(full_ref, k, state)))
        print(f"full-size term plan {plan:#04x} calibration {cal} rounding {rounding}: max per-channel rel err per step " + " ".join(f"{v:.3e}" for v in errs))
        assert max(errs) < bound, (hex(plan), cal, rounding, errs)
        del e
        torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_full_size_longitude_shift_equivariance_and_fp16_mode(full):
    """Size-independent property at full size (no oracle needed) + agreement of the two precision modes."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x, e = full
    y = e.step(x.cuda())
    p2 = dict(params)
    p2["const_masks"] = torch.roll(params["const_masks"], 480, dims=-1)
    e2 = PanguEngine(g, device="cuda:0")
    e2.load_params(p2)
    y2 = e2.step(torch.roll(x, 480, dims=-1).cuda())
    assert O.per_channel_rel_err(torch.roll(y2, -480, dims=-1).cpu(), y.cpu()).max().item() < 2 * DEF_TOL
    del e2


# --------------------------------------------------------------------------------------------- #
#  through the reference's own API surface (skyrim.core mirror)
# --------------------------------------------------------------------------------------------- #
def test_five_day_rollout_of_the_default_mode(toy):
    """20 autoregressive 6-h steps of the DEFAULT mode on the toy grid against the oracle (the 4-step full-size rollouts say nothing
    about states that have drifted away from the calibration state).  Two figures per step, both asserted at every step with the default
    mode's own tolerance:
    (a) STEP parity along the oracle's trajectory -- fed the oracle's state k, the engine's step k + 1 against the oracle's;
    (b) the FREE-RUNNING difference -- each side feeds its own output back.
    Measured over 40 steps (10 days; 107 s of oracle time, hence 20 here): (a) at most 1.35e-4, at step 40; (b) 1.1 - 1.2e-4 throughout -- the
    synthetic network damps a perturbation (the oracle's own response to a 1e-6 relative perturbation of the state: 7e-6 after one step, 1.4e-6
    from step 5 on), so the free-running difference is each step's fresh rounding, not an accumulation."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    e = PanguEngine(g, device="cuda:0")
    e.load_params(params)
    free, ref = x.cuda().clone(), x
    fed_err, free_err = [], []
    for k in range(20):
        fed = e.step(ref.cuda()).cpu()
        ref = O.forward(params, ref)
        fed_err.append(O.per_channel_rel_err(fed, ref).max().item())
        e.step(free, free)
        free_err.append(O.per_channel_rel_err(free.cpu(), ref).max().item())
    assert torch.isfinite(free).all()
    print("pangu default mode, 20-step rollout (toy grid): step parity along the oracle trajectory max %.3e, free-running max %.3e" % (max(fed_err), max(free_err)))
    assert max(fed_err) < DEF_TOL, fed_err
    assert max(free_err) < DEF_TOL, free_err


def test_pangu_model_rollout_through_reference_api(toy, tmp_path):
    """GlobalModel.rollout -> predict_one_step -> run_basic_inference -> TimeLoop -> skpangu_step, with the
    per-step netCDF files of save_forecast; values against the oracle's rollout from the same IC."""
    import datetime
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.labeled import open_dataarray
    g, params, x = toy
    t0 = datetime.datetime(2024, 5, 13, 18, 0)
    m = PanguModel(ic_source="gfs", geom=g, params=params)
    assert m.in_channel_names == m.out_channel_names and len(m.out_channel_names) == 69
    pred, paths = m.rollout(t0, n_steps=2, save=True, save_config={"output_dir": str(tmp_path), "file_type": "netcdf"})
    assert pred.dims == ("time", "channel", "lat", "lon") and pred.shape == (2, 69, g.n_lat, g.n_lon)
    assert [p.rsplit("/", 1)[1] for p in paths] == ["pangu__synthetic__20240513_18:00__20240514_00:00.nc",
                                                     "pangu__file__20240514_00:00__20240514_06:00.nc"]
    ic = torch.from_numpy(m.data_source[t0])
    want = O.rollout(params, ic, 2)
    assert O.per_channel_rel_err(torch.from_numpy(pred.values[0].copy()), want[0]).max().item() < 1e-3
    assert O.per_channel_rel_err(torch.from_numpy(pred.values[1]), want[1]).max().item() < 1e-3
    back = open_dataarray(paths[0])
    assert O.per_channel_rel_err(torch.from_numpy(back.values[1].copy()), want[0]).max().item() < 1e-3
    fc = m.forecast(t0, n_steps=2, channels=["t2m", "z500"])
    assert fc.shape == (3, 2, g.n_lat, g.n_lon)
    assert np.allclose(fc.values[2, 0], pred.values[1, 68], rtol=1e-4, atol=1e-3)


def test_rollout_keeps_the_state_in_hbm_and_saves_the_same_files(toy, tmp_path):
    """``rollout`` feeds ``pred`` straight back -- the state must not be uploaded again after step 0
    (core/models/utils.py: ResidentState), the per-step files are written by a worker thread while the next step runs, and both must be
    invisible: same values as the upload path, files byte-identical to a synchronous ``save_forecast`` of the same predictions."""
    import datetime
    import filecmp
    from skyrim_amd.common import save_forecast
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.core.models.utils import perturb_initial_conditions
    g, params, x = toy
    t0 = datetime.datetime(2024, 5, 13, 18, 0)
    m = PanguModel(ic_source="gfs", geom=g, params=params)
    cfg = {"output_dir": str(tmp_path / "async"), "file_type": "netcdf"}
    pred, paths = m.rollout(t0, n_steps=3, save=True, save_config=cfg)
    assert m.model.io_counters == {"state_uploads": 1, "resident_hits": 2}        # one H2D (the initial condition), then HBM-resident
    assert not pred.values.flags.writeable
    # the upload path: every prediction handed back as a COPY (a different array: it must be uploaded), saved synchronously
    m2 = PanguModel(ic_source="gfs", geom=g, params=params)
    p2, t, src = None, t0, m2.source_label
    for k in range(3):
        p2 = m2.predict_one_step(t, initial_condition=None if p2 is None else p2.copy())
        sync = save_forecast(p2, "pangu", t, t + m2.time_step, src, config={"output_dir": str(tmp_path / "sync"), "file_type": "netcdf", "forecast_id": cfg["forecast_id"]})
        assert Path(sync).name == Path(paths[k]).name and filecmp.cmp(sync, paths[k], shallow=False), k
        t, src = t + m2.time_step, "file"
    assert m2.model.io_counters == {"state_uploads": 3, "resident_hits": 0}
    assert np.array_equal(p2.values, pred.values)
    # an edited prediction is a copy (the delivered array is read-only), and a copy is uploaded
    with pytest.raises(ValueError):
        pred.values[1, 0, 0, 0] = 0.0
    edited = perturb_initial_conditions(pred, "t2m", 48.0, 11.5, 250.0)
    assert edited.values.flags.writeable and edited.values[1, 68].min() <= 250.0
    nxt = m.predict_one_step(t, initial_condition=edited)
    # the delivered array's last copy may still be in flight: shape and coordinates do not wait, the first read of the numbers does
    assert nxt._ready is not None and nxt.shape == pred.shape and nxt.time.values[0] == np.datetime64(t) and nxt._ready is not None
    assert m.model.io_counters["state_uploads"] == 2 and np.array_equal(nxt.values[0], edited.values[1]) and nxt._ready is None
    fed = m.predict_one_step(t + m.time_step, initial_condition=nxt)              # fed back unread or read: the same resident state
    assert m.model.io_counters["resident_hits"] == 3 and np.array_equal(fed.values[0], nxt.values[1])


def test_forecast_interleaves_6h_and_24h_networks(toy):
    """Multi-step generator with a 24-h parameter set: step 4 comes from the 24-h network applied to the initial
    state, steps 1-3 and 5 from the 6-h network (earth2mip's Pangu schedule, what GlobalModel.forecast drives)."""
    import datetime
    from skyrim_amd.pangu.timeloop import PanguTimeLoop
    g, params, x = toy
    p24 = init_synthetic(g, 24)
    loop = PanguTimeLoop(params, g, params24=p24)
    t0 = datetime.datetime(2024, 1, 1)
    outs = []
    for k, (t, y, _) in enumerate(loop(t0, x[None, None].cuda())):
        outs.append((t, y[0].cpu()))
        if k == 5:
            break
    assert [o[0] for o in outs] == [t0 + datetime.timedelta(hours=6 * k) for k in range(6)]
    assert torch.equal(outs[0][1], x)
    want = [x]
    for k in range(1, 6):
        want.append(O.forward(p24, x) if k == 4 else O.forward(params, want[k - 1]))
    for k in range(1, 6):
        assert O.per_channel_rel_err(outs[k][1], want[k]).max().item() < 1e-3, k


def test_member_parallel_ensemble_single_gpu(toy):
    """config[4] semantics on one GPU: perturbed members, mean / spread, gathered member states."""
    from skyrim_amd.pangu.engine import PanguEngine
    from skyrim_amd.pangu.ensemble import MemberParallelEnsemble, perturbed_member
    g, params, x = toy
    eng = PanguEngine(g, device="cuda:0")
    eng.load_params(params)
    ens = MemberParallelEnsemble(eng.step, 3, params["norm.std"], perturb_scale=1e-2)
    out = ens.run(x.cuda(), 2, gather=True)
    assert out["members"].shape == (3, 69, g.n_lat, g.n_lon)
    ref = torch.stack([O.rollout(params, perturbed_member(x.cuda(), params["norm.std"], m, 1e-2).cpu(), 2)[-1] for m in range(3)])
    assert O.per_channel_rel_err(out["mean"].cpu(), ref.mean(0)).max().item() < 1e-3
    assert torch.allclose(out["spread"].cpu(), ref.std(0, unbiased=False), rtol=0.05, atol=1e-3 * ref.abs().max().item())


@pytest.mark.timeout(600)
def test_skyrim_facade_default_grid():
    """config[0] plumbing on the real grid: Skyrim('pangu').predict(6 h) -> GlobalPrediction (2, 69, 721, 1440)."""
    from skyrim_amd.core import Skyrim
    s = Skyrim("pangu", ic_source="gfs")
    pred, paths = s.predict("20240513", "1800", lead_time=6, save=False)
    assert pred.prediction.shape == (2, 69, 721, 1440) and paths == []
    assert np.isfinite(pred.prediction.values).all()
    assert pred.prediction.time.values[1] == np.datetime64("2024-05-14T00:00")
    assert abs(pred.point(48.0, 11.5, "t2m", n_step=1)) < 1e4


def test_step_through_the_custom_op_boundary(toy):
    """torch.ops.skyrim_hip.pangu_* called directly (the engine's own call path): same result as PanguEngine.step, honours the
    current stream, refuses CPU tensors and wrong dtypes."""
    from skyrim_amd import ops  # noqa: F401
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    eng = PanguEngine(g, device="cuda:0")
    eng.load_params(params)
    xd = x.cuda()
    want = eng.step(xd)
    out = torch.empty_like(xd)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.ops.skyrim_hip.pangu_step(eng._ctx.value, xd, out)
    side.synchronize()
    assert torch.equal(out, want)
    x1 = torch.empty(eng.tokens(1), dtype=torch.float32, device="cuda:0")
    torch.ops.skyrim_hip.pangu_patch_embed(eng._ctx.value, xd, x1)
    assert torch.equal(x1, eng.patch_embed(xd))
    with pytest.raises(NotImplementedError):
        torch.ops.skyrim_hip.pangu_step(eng._ctx.value, x, torch.empty_like(x))
    with pytest.raises(ValueError):
        torch.ops.skyrim_hip.pangu_step(eng._ctx.value, xd.double(), out)


@pytest.mark.parametrize("conv", [dict(roll_sign=+1), dict(pad="back"), dict(roll_sign=+1, pad="back", mask_value=-1000.0),
                                  dict(surface="last"), dict(qkv_order="h3d"), dict(bias_index="kq"),
                                  dict(surface="last", qkv_order="h3d", bias_index="kq", roll_sign=+1)])
def test_switchable_conventions_match_the_oracle(toy, conv):
    """roll_sign / pad / mask_value / surface / qkv_order / bias_index (include/skyrim_pangu.h skpangu_config <-> oracle Conventions): the
    engine follows the oracle under each setting, and does NOT match the oracle of the other setting -- the test that flips the
    convention.  The last three are applied in skpangu_prepare only (window tables, a row permutation, the bias gather)."""
    from skyrim_amd.pangu.engine import PanguEngine
    g0, params, x = toy
    g = PanguGeometry(g0.n_lat, g0.n_lon, conv.get("pad", "centre"))
    eng = PanguEngine(g, device="cuda:0", roll_sign=conv.get("roll_sign", -1), mask_value=conv.get("mask_value", -100.0),
                      surface=conv.get("surface", "first"), qkv_order=conv.get("qkv_order", "3hd"), bias_index=conv.get("bias_index", "qk"))
    eng.load_params(params)
    y = eng.step(x.cuda()).cpu()
    want = O.forward(params, x, conv=O.Conventions(**conv))
    assert O.per_channel_rel_err(y, want).max().item() < DEF_TOL
    assert O.per_channel_rel_err(y, O.forward(params, x)).max().item() > 1e-3
    y2 = eng.step(eng.step(x.cuda())).cpu()                     # rolled + unrolled blocks compose over steps
    assert O.per_channel_rel_err(y2, O.rollout(params, x, 2, conv=O.Conventions(**conv))[1]).max().item() < DEF_TOL


def test_fused_mlp_kernel_matches_the_two_gemm_path_and_the_oracle(toy, ref):
    """csrc/fused_mlp.hip (fc1 -> GELU -> fc2 -> LayerNorm -> residual in one kernel, token tile and hidden in registers) against
    the two tiled GEMMs of round 1 and against the oracle: one block at each resolution (C = 192: 64 tokens per wave, C = 384:
    32) and the whole step; the toy grid's token counts (4992, 1344) are not multiples of the 256 / 128-token tiles."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    taps, y_ref = ref
    outs = {}
    for mlp in ("fused", "split"):
        for prec in ("f16x3q", "bf16x3"):
            eng = PanguEngine(g, prec, "cuda:0", mlp=mlp)
            eng.load_params(params)
            assert eng.mlp == mlp
            outs[mlp, prec] = eng.step(x.cuda()).cpu()
            assert O.per_channel_rel_err(outs[mlp, prec], y_ref).max().item() < 3e-4, (mlp, prec)
            if prec == "f16x3q":
                x1 = taps["embed"].float().cuda()
                b = eng.block(1, 0, x1).cpu()
                assert rel(b, taps["layer1.block0"]) < STAGE_TOL[prec], mlp
                x2 = taps["down"].float().cuda()
                want2 = O.earth_block(O._block_params(params, 2, 0), taps["down"], O.Geometry(g.n_lat, g.n_lon).res(2), O.HEADS[1], False)
                assert rel(eng.block(2, 0, x2).cpu(), want2) < STAGE_TOL[prec], mlp
    assert O.per_channel_rel_err(outs["fused", "f16x3q"], outs["split", "f16x3q"]).max().item() < 2e-4


@pytest.mark.parametrize("plan", [0x00, 0x05, 0x0A, 0x0F, 0xF0, 0xFF])
def test_per_layer_term_plan(toy, ref, plan):
    """skpangu_config.term_plan: bit l = layer l + 1 runs proj / fc1 / fc2 with two MFMA terms (csrc/fused_block2.hip), bit 4 + l its QKV
    with one.  Every plan stays inside the bar; a layer that keeps its lo planes is bit-identical to the three-term engine's block, a layer
    that drops them is not; the error grows with the number of rounded weight sets."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    taps, y_ref = ref
    eng = PanguEngine(g, "f16x3q", "cuda:0", term_plan=plan)
    eng.load_params(params)
    assert eng.term_plan == plan
    err = O.per_channel_rel_err(eng.step(x.cuda()).cpu(), y_ref).max().item()
    assert err < (3e-4 if plan == 0 else 1e-3), (hex(plan), err)
    base = PanguEngine(g, "f16x3q", "cuda:0")
    base.load_params(params)
    for layer, tap in ((1, "embed"), (2, "down"), (3, "down"), (4, "up")):
        xin = taps[tap].float().cuda().contiguous()
        same = torch.equal(eng.block(layer, 0, xin), base.block(layer, 0, xin))
        touched = bool((plan >> (layer - 1)) & 1) or bool((plan >> (3 + layer)) & 1)
        assert same == (not touched), (hex(plan), layer)


def test_calibration_folds_the_dropped_weight_term_into_the_biases(toy, ref):
    """skpangu_calibrate (include/skyrim_pangu.h): the one-plane Linears' biases gain (W - fp16(W)) x mean(operand), the means taken
    from ONE three-term step on a state that is not the forecast's.  Against the uncalibrated plan the step error falls (the CPU
    statement of the same thing: tests/test_term_plan_calibration.py); calibrating twice is calibrating once; the master biases come
    back with calibration "off"; engines without a plan do nothing."""
    from skyrim_amd.pangu.engine import PanguEngine, calibration_state
    g, params, x = toy
    _, y_ref = ref
    errs, outs = {}, {}
    for cal in ("off", "synthetic"):
        eng = PanguEngine(g, "f16x3q", "cuda:0", term_plan=0xFF)
        eng.load_params(params, calibration=cal, guard=False)           # the plan as given: the load-time guard would leave an uncalibrated 0xFF
        assert eng.calibrated_on == (None if cal == "off" else "synthetic")
        outs[cal] = eng.step(x.cuda()).cpu()
        errs[cal] = O.per_channel_rel_err(outs[cal], y_ref).max().item()
    print(f"plan 0xFF one step: uncalibrated {errs['off']:.2e}, calibrated {errs['synthetic']:.2e}")
    assert errs["synthetic"] < 0.85 * errs["off"], errs         # measured 5.0e-4 -> 3.8e-4 (the CPU emulation of the same: 4.9e-4 -> 2.8e-4)
    cs = calibration_state(g, params["norm.mean"], params["norm.std"])
    assert not torch.equal(cs, x)
    eng.calibrate(cs)                                               # again, same state: the same biases
    assert torch.equal(eng.step(x.cuda()).cpu(), outs["synthetic"])
    eng.calibrate(x)                                                # in-sample: different biases, still inside the bar
    y_in = eng.step(x.cuda()).cpu()
    assert not torch.equal(y_in, outs["synthetic"])
    assert O.per_channel_rel_err(y_in, y_ref).max().item() < 0.85 * errs["off"]
    eng.calibrate(None)                                             # no state: the master biases, i.e. the uncalibrated plan
    assert eng.calibrated_on is None and torch.equal(eng.step(x.cuda()).cpu(), outs["off"])
    e3 = PanguEngine(g, "f16x3q", "cuda:0")
    e3.load_params(params)
    y3 = e3.step(x.cuda())
    e3.calibrate(cs)
    assert e3.calibrated_on is None and torch.equal(e3.step(x.cuda()), y3)
    with pytest.raises(ValueError):
        PanguEngine(g, "f16x3q", "cuda:0", term_plan=0xFF).load_params(params, calibration="era5")


def test_calibration_taps_are_the_oracles_operands(toy, monkeypatch):
    """pangu/calibration.py engine_taps: what it reads back from the tiled three-term engine, block by block, against the operands the
    oracle's Linears see on the same state (stream, mid-block stream, hidden activation: token order on both sides)."""
    from skyrim_amd.pangu.engine import PanguEngine
    from skyrim_amd.pangu.calibration import engine_taps
    g, params, x = toy
    names = {id(v): k for k, v in params.items()}
    seen = {}
    plain = O._linear

    def linear(xx, w, b=None, *a, **kw):
        key = names.get(id(w), "")
        if key.endswith(("attn.qkv.weight", "mlp.fc1.weight", "mlp.fc2.weight")):
            seen[key[:-len(".weight")]] = xx
        return plain(xx, w, b, *a, **kw)

    monkeypatch.setattr(O, "_linear", linear)
    taps = {}
    O.forward(params, x, taps=taps)
    monkeypatch.setattr(O, "_linear", plain)
    tap = PanguEngine(g, "f16x3q", "cuda:0", mlp="split")
    tap.load_params(params)
    before = {(1, 0): "embed", (2, 0): "down", (3, 0): "layer2.block5", (4, 0): "up"}
    worst = {}
    n = 0
    for layer, i, ops in engine_taps(tap, params, x):
        pre = f"layer{layer}.block{i}."
        xin = taps[before.get((layer, i), f"layer{layer}.block{i - 1}")]
        e = {"attn.qkv": rel(ops["attn.qkv"], xin), "mlp.fc1": rel(ops["mlp.fc1"], seen[pre + "mlp.fc1"]), "mlp.fc2": rel(ops["mlp.fc2"], seen[pre + "mlp.fc2"])}
        print(pre, " ".join(f"{k} {v:.1e}" for k, v in e.items()), tuple(ops["attn.proj"].shape))
        assert ops["attn.proj"].shape == xin.shape and torch.isfinite(ops["attn.proj"]).all()
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0.0), v)
        n += 1
    assert n == 16 and max(worst.values()) < 3e-3, worst


def test_compensated_rounding_of_the_one_plane_weights(toy, ref):
    """pangu/calibration.py through PanguEngine.load_params(rounding="compensated"): the operands come from the engine's own buffers (one
    step of the tiled three-term engine on the built-in calibration state), the weights of the short Linears are rounded with error
    feedback against the operand covariances, the biases take the mean of the rest.  Same kernels, same step time; against nearest
    rounding + bias fold the step error falls towards the three-term engine's (CPU statement: tests/test_term_plan_calibration.py)."""
    from skyrim_amd.pangu.engine import PanguEngine
    from skyrim_amd.pangu.calibration import engine_taps
    g, params, x = toy
    taps, y_ref = ref
    errs = {}
    for rounding in ("nearest", "compensated"):
        eng = PanguEngine(g, "f16x3q", "cuda:0", term_plan=0xFF)
        eng.load_params(params, rounding=rounding, guard=False)
        assert eng.rounding == rounding and eng.calibrated_on == "synthetic"
        state = x.cuda().clone()
        e = []
        for _ in range(2):
            state = eng.step(state)
            e.append(state.cpu())
        errs[rounding] = O.per_channel_rel_err(e[0], y_ref).max().item()
    e3 = PanguEngine(g, "f16x3q", "cuda:0")
    e3.load_params(params)
    errs["three terms"] = O.per_channel_rel_err(e3.step(x.cuda()).cpu(), y_ref).max().item()
    print("plan 0xFF one step: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["compensated"] < 0.6 * errs["nearest"] and errs["compensated"] < 3 * errs["three terms"], errs
    eng.calibrate(None)                                             # back to nearest rounding, master biases
    off = PanguEngine(g, "f16x3q", "cuda:0", term_plan=0xFF)
    off.load_params(params, calibration="off", guard=False)
    assert torch.equal(eng.step(x.cuda()), off.step(x.cuda()))
    # the operands the statistics are taken from are the oracle's (attention output, mid-block stream, hidden activation of block 0)
    tap = PanguEngine(g, "f16x3q", "cuda:0", mlp="split")
    tap.load_params(params)
    layer, i, ops = next(engine_taps(tap, params, x))
    assert (layer, i) == (1, 0) and set(ops) == {"attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"}
    bp = O._block_params(params, 1, 0)
    x1 = taps["embed"]
    assert rel(ops["attn.qkv"], x1) < 1e-3
    want_mid_to_out = O.earth_block(bp, x1, O.Geometry(g.n_lat, g.n_lon).res(1), O.HEADS[0], False)
    mid = ops["mlp.fc1"].cpu()
    hid = torch.nn.functional.gelu(torch.nn.functional.linear(mid, bp["mlp.fc1.weight"], bp["mlp.fc1.bias"]))
    assert rel(ops["mlp.fc2"], hid) < 2e-3
    out = mid + torch.nn.functional.layer_norm(torch.nn.functional.linear(hid, bp["mlp.fc2.weight"], bp["mlp.fc2.bias"]), (192,), bp["norm2.weight"], bp["norm2.bias"], 1e-5)
    assert rel(out, want_mid_to_out) < 1e-3                          # mid-block stream and hidden are the block's own
    assert ops["attn.proj"].shape == x1.shape and torch.isfinite(ops["attn.proj"]).all()


def test_one_term_block_gemms_in_the_coarse_layers(toy, ref):
    """Term-plan bits 8-11 (include/skyrim_pangu.h): proj / fc1 / fc2 of a layer with ONE MFMA term -- the activation operands too as their
    fp16 hi plane (csrc/fused_block2.hip, ONE).  "f16x1m" = f16x2m with layers 2 / 3 (12 of 16 blocks) in that form.  With the weights
    fitted to the rounded operands (rounding="compensated") the step stays near the three-term error; the oracle emulation of exactly
    this plan: 1.3e-4 against 6.5e-5."""
    from skyrim_amd.pangu.engine import PanguEngine, TERM_PLANS
    g, params, x = toy
    taps, y_ref = ref
    assert TERM_PLANS["f16x1m"] == 0x66F
    errs, outs = {}, {}
    for rounding in ("nearest", "compensated"):
        eng = PanguEngine(g, "f16x1m", "cuda:0")
        eng.load_params(params, rounding=rounding, guard=False)
        assert eng.term_plan == 0x66F
        outs[rounding] = eng.step(x.cuda()).cpu()
        errs[rounding] = O.per_channel_rel_err(outs[rounding], y_ref).max().item()
    two = PanguEngine(g, "f16x2m", "cuda:0")
    two.load_params(params, rounding="compensated")
    y2 = two.step(x.cuda()).cpu()
    errs["two-term, compensated"] = O.per_channel_rel_err(y2, y_ref).max().item()
    print("f16x1m one step: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["nearest"] < 1e-3 and errs["compensated"] < 5e-4, errs
    assert not torch.equal(outs["compensated"], y2)
    # a one-term layer's block differs from the two-term engine's, an untouched layer's does not
    x2, x1 = taps["down"].float().cuda().contiguous(), taps["embed"].float().cuda().contiguous()
    assert not torch.equal(eng.block(2, 0, x2), two.block(2, 0, x2))
    with pytest.raises(RuntimeError):
        PanguEngine(g, "f16x3q", "cuda:0", term_plan=0x100)          # one-term without two-term: rejected by skpangu_create


def test_time_loop_calibrates_on_the_first_initial_condition(toy, ref):
    """PanguTimeLoop(calibration="first") -- OPT-IN: the biases are fitted on the first state the loop is called with, once; later calls
    reuse them.  The default is "synthetic" whatever the weights' source (deterministic: same initial condition, same bits, in every
    process); a state file can be named instead (``calibration=<path>`` / SKYRIM_PANGU_CALIBRATION)."""
    import datetime
    from skyrim_amd.pangu.timeloop import PanguTimeLoop
    g, params, x = toy
    _, y_ref = ref
    loop = PanguTimeLoop(params, g, calibration="first")
    assert loop.engine.calibrated_on is None
    t0 = datetime.datetime(2024, 1, 1)
    it = loop(t0, x[None, None].cuda())
    next(it)
    _, y, _ = next(it)
    it.close()
    assert loop.engine.calibrated_on == "first"
    assert O.per_channel_rel_err(y[0].cpu(), y_ref).max().item() < DEF_TOL
    it = loop(t0, synthetic_state(g, 9)[None, None].cuda())         # a second forecast: same biases
    next(it)
    it.close()
    it = loop(t0, x[None, None].cuda())
    next(it)
    _, y2, _ = next(it)
    it.close()
    assert torch.equal(y2, y)
    off = PanguTimeLoop(params, g, calibration="off", guard=False)
    assert off.engine.calibrated_on is None and PanguTimeLoop(params, g).engine.calibrated_on == "synthetic"


def test_calibration_is_deterministic_and_can_name_a_state_file(toy, tmp_path, monkeypatch):
    """Two fresh loops that saw DIFFERENT first initial conditions give the same bits for the same initial condition (default
    calibration: the synthetic state); a named state file calibrates on that state, the same in every process."""
    import datetime
    from skyrim_amd.pangu.timeloop import PanguTimeLoop
    g, params, x = toy
    t0 = datetime.datetime(2024, 1, 1)

    def forecast(loop, state):
        it = loop(t0, state[None, None].cuda())
        next(it)
        _, y, _ = next(it)
        it.close()
        return y

    a, b = PanguTimeLoop(params, g), PanguTimeLoop(params, g)
    forecast(a, synthetic_state(g, 5))                                   # a has seen another state first
    assert torch.equal(forecast(a, x), forecast(b, x))
    path = tmp_path / "analysis.pt"
    torch.save(synthetic_state(g, 7), path)
    monkeypatch.setenv("SKYRIM_PANGU_CALIBRATION", str(path))
    c, d = PanguTimeLoop(params, g), PanguTimeLoop(params, g)
    assert c.engine.calibrated_on == "state"
    forecast(c, synthetic_state(g, 5))
    assert torch.equal(forecast(c, x), forecast(d, x))
    with pytest.raises(ValueError):
        PanguTimeLoop(params, g, calibration=str(tmp_path / "missing.pt"))


def _outlier_params(params, seed=3, frac=0.01, scale=30.0):
    """Trained-weight-like stress: 1 % of the rows of every Linear weight (2-d parameters but the bias tables) scaled x30 -- heavy-tailed rows the
    1/sqrt(fan_in) random init never has (their W - fp16(W) residue and their activations are what the one-plane plans lean on)."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in params.items():
        if v.dim() == 2 and k.endswith(".weight") and "bias_table" not in k and "norm" not in k:
            v = v.clone()
            rows = torch.randperm(v.shape[0], generator=gen)[:max(1, int(frac * v.shape[0]))]
            v[rows] *= scale
        out[k] = v
    return out


def test_outlier_weights_and_states_stay_inside_the_bar_or_trip_the_guard(toy):
    """The default mode under outliers.  (1) 1 % of the rows of every Linear weight x5: the default-mode assertion holds (measured 9e-5; nearest
    rounding 2.8e-4).  (2) the same x30: NO mode survives that on every Linear class at once -- three terms everywhere and bf16x3 sit at ~1e-2
    like the default, while each class alone stays below 3e-4 (tools/pangu_outlier_scan.py) -- so what is asserted is that the default's term plan
    is not the weak link: within 2x of the three-term engine on the same tensors.  (3) two channels of the state pushed out by 1e4 sigma: the bare
    engine stays finite but leaves the bar (22 significant bits per operand element: 1e4 x 2^-22 against O(1) signals), so the time loop REFUSES
    such a state -- FloatingPointError from its range check (PanguTimeLoop.RANGE_LIMIT), like the FiniteGuard does for non-finite output."""
    import datetime
    from skyrim_amd.pangu.engine import PanguEngine
    from skyrim_amd.pangu.timeloop import PanguTimeLoop
    g, params, x = toy

    def run(p, state, guard=None, **kw):
        e = PanguEngine(g, device="cuda:0", **kw)
        e.load_params(p, guard=guard)
        return e.step(state.cuda()).cpu()

    mild = _outlier_params(params, scale=5.0)
    err = O.per_channel_rel_err(run(mild, x), O.forward(mild, x)).max().item()
    print(f"outlier stress (1 % of the weight rows x5): max per-channel rel err {err:.3e} ({DEFAULT_PRECISION})")
    assert err < DEF_TOL, err
    heavy = _outlier_params(params, scale=30.0)
    ref = O.forward(heavy, x)
    e_def = O.per_channel_rel_err(run(heavy, x, guard=False), ref).max().item()
    e_3t = O.per_channel_rel_err(run(heavy, x, precision="f16x3q", term_plan=0), ref).max().item()
    print(f"outlier stress (1 % of the weight rows x30): default plan as given {e_def:.3e}, three terms everywhere {e_3t:.3e}")
    assert e_def < 2.0 * e_3t + 1e-3, (e_def, e_3t)
    mean, std = params["norm.mean"], params["norm.std"]
    wild = x.clone()
    for c in (3, 40):
        wild[c] = mean[c] + 1e4 * (x[c] - mean[c])
    y = run(mild, wild)                                          # the bare engine (C ABI) computes on: finite, but outside the bar
    if torch.isfinite(y).all():
        err = O.per_channel_rel_err(y, O.forward(mild, wild)).max().item()
        print(f"outlier stress (x5 rows + two channels at 1e4 sigma), bare engine: max per-channel rel err {err:.3e}")
    loop = PanguTimeLoop(mild, g)
    with pytest.raises(FloatingPointError, match="sigma"):       # the reference-API path refuses the forecast
        it = loop(datetime.datetime(2024, 1, 1), wild[None, None].cuda())
        next(it)
    it = loop(datetime.datetime(2024, 1, 1), x[None, None].cuda())    # and an ordinary state goes through
    next(it)
    _, y1, _ = next(it)
    it.close()
    assert torch.isfinite(y1).all()


def test_load_time_guard_never_applies_the_default_plan_blind(toy, ref):
    """PanguEngine._guard: after the plan is prepared, ONE step of it against ONE step of the three-term engine on the
    calibration state, per channel in sigma units; at GUARD_TOL (5e-4, half the bar) or more the engine falls back 0x66F -> 0x6F -> 0x00, refits,
    says so and reports the plan in effect.  (i) the synthetic weights keep the default plan; (ii) a plan loaded without calibration / with
    nearest rounding does not survive (that combination used to run blind); (iii) QKV rows x30: the one-term block GEMMs go, the
    result meets the bar against the oracle; (iv) 1 % of every Linear's rows x30: even two three-term evaluations disagree by 1e-2 -- the
    weights amplify rounding in every mode, the oracle's fp32 included -- and the load is REFUSED (FloatingPointError), or, run anyway
    (SKYRIM_PANGU_GUARD=warn), warned about."""
    import warnings
    from skyrim_amd.pangu.engine import GUARD_TOL, PanguEngine
    from tools.pangu_outlier_scan import outliers
    g, params, x = toy
    _, y_ref = ref
    e = PanguEngine(g, device="cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("error")                              # (i) no fall-back, no warning
        e.load_params(params)
    assert e.term_plan_in_effect == e.term_plan_requested == 0x66F
    (plan, err), = e.guard_report
    assert plan == 0x66F and 0 < err < GUARD_TOL, e.guard_report
    print(f"guard, synthetic weights: plan {plan:#05x} at {err:.2e} sigma of the three-term engine (tolerance {GUARD_TOL:g}), {e.guard_seconds * 1e3:.0f} ms")
    for kw in (dict(calibration="off"), dict(rounding="nearest")):  # (ii)
        with pytest.warns(RuntimeWarning, match="term plan 0x66f"):
            e.load_params(params, **kw)
        assert e.term_plan_in_effect != 0x66F and e.guard_report[-1][1] < GUARD_TOL and e.term_plan_requested == 0x66F
        assert O.per_channel_rel_err(e.step(x.cuda()).cpu(), y_ref).max().item() < DEF_TOL
    e.load_params(params)                                           # a later load starts from the plan asked for again
    assert e.term_plan_in_effect == 0x66F
    qkv = outliers(params, scale=30.0, only="qkv")                  # (iii)
    with pytest.warns(RuntimeWarning, match="running plan 0x06f"):
        e.load_params(qkv)
    assert e.term_plan_in_effect == 0x6F and [p_ for p_, _ in e.guard_report] == [0x66F, 0x6F]
    err = O.per_channel_rel_err(e.step(x.cuda()).cpu(), O.forward(qkv, x)).max().item()
    print(f"guard, QKV rows x30: {[(hex(a), float(f'{b:.2e}')) for a, b in e.guard_report]} -> vs oracle {err:.2e}")
    assert err < 1e-3, err
    heavy = outliers(params, scale=30.0)                            # (iv)
    with pytest.raises(FloatingPointError, match="amplify"):
        e.load_params(heavy)
    e.release()


@pytest.mark.parametrize("precision", [DEFAULT_PRECISION, "f16x2m", "f16x3q"])
def test_fused_qkv_attention_is_bit_identical_to_the_two_launches(toy, monkeypatch, precision):
    """Round 6: QKV + window attention as ONE kernel (csrc/attention.hip: qkv_attention_kernel; q / k / v stay in registers) against the
    QKV launch + attention launch it replaces (SKP_SPLIT_ATTN=1 at engine construction): the same products in the same order, so the same
    BITS -- on the toy grid, whose windows carry padding rows and the shifted-window mask, in the default plan (one weight plane in the coarse
    layers, hi / lo in layers 1 / 4), the two-term plan, and three terms (where only the C = 192 layers fuse: hi / lo planes at C = 384 do not fit
    LDS beside the bias copies).  The per-stage profile shows which form ran."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    outs, launches = {}, {}
    for split in (True, False):
        if split:
            monkeypatch.setenv("SKP_SPLIT_ATTN", "1")
        else:
            monkeypatch.delenv("SKP_SPLIT_ATTN", raising=False)
        e = PanguEngine(g, precision, "cuda:0")
        e.load_params(params, guard=False)
        e.profile(True)
        outs[split] = e.step(x.cuda())
        launches[split] = {s_["name"]: s_["launches"] for s_ in e.profile_read()}
        e.profile(False)
        y2 = e.step(outs[split])                       # and a second step on its own output
        outs[split] = (outs[split].cpu(), y2.cpu())
        e.release()
    assert launches[True]["qkv_r0"] == 4 and launches[True]["qkv_r1"] == 12
    assert launches[False]["qkv_r0"] == 0 and launches[False]["qkv_r1"] == (12 if precision == "f16x3q" else 0)
    assert launches[False]["attn_r0"] == 4 and launches[False]["attn_r1"] == 12
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])


def test_step_as_a_captured_hip_graph(toy):
    """One in-place step captured as a HIP graph (72 launches -> one replay): same bits as the eager step, replayable."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    eng = PanguEngine(g, device="cuda:0")
    eng.load_params(params)
    want1 = eng.step(x.cuda())
    want2 = eng.step(want1)
    xs = x.cuda().clone()
    graph = eng.capture(xs)
    assert torch.equal(xs, x.cuda())
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(xs, want1)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(xs, want2)
