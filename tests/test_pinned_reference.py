"""tools/pin_reference.py: the script that turns "parity unpinned" into a pin where the reference itself can run, and the tests that pick its
fixtures up.

In this build environment neither earth2mip / onnxruntime / torch-harmonics / jax nor any weight file exists, so the
pick-up tests SKIP with that reason and the oracles stay unpinned.  The stub backend exercises everything else: the fixture format, the
sampling, the free-running comparison, the refusal to treat a self-oracle fixture as a pin."""
import importlib.util
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("pin_reference", ROOT / "tools" / "pin_reference.py")
PIN = importlib.util.module_from_spec(spec)
sys.modules["pin_reference"] = PIN
spec.loader.exec_module(PIN)

GOLDEN = ROOT / "tests" / "golden"
ENV = {"pangu": "SKYRIM_PANGU_WEIGHTS", "sfno": "SKYRIM_SFNO_WEIGHTS", "graphcast": "SKYRIM_GRAPHCAST_WEIGHTS"}


def test_stub_backend_writes_a_fixture_that_pins_nothing(tmp_path):
    out = PIN.main(["pangu", "--backend", "stub", "--grid", "25x96", "--steps", "2", "--seed", "5", "--out", str(tmp_path)])
    f = dict(np.load(out))
    assert out.name == "pangu_ref_25x96.npz" and str(f["model"]) == "pangu" and not bool(f["pinned"]) and "self-oracle" in str(f["backend"])
    assert f["sub"].shape == (2, 69, 5, 6) and f["absmax"].shape == (2, 69) and int(f["seed"]) == 5 and f["stride"].tolist() == [6, 16]
    # the comparison the pick-up tests make, on the oracle itself: free-running over both steps, zero difference
    from oracle import pangu_oracle as O
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    p = init_synthetic(PanguGeometry(25, 96), 5)
    _, state_fn, _ = PIN.model_spec("pangu", 25, 96)
    errs = PIN.check_against(f, lambda h, k: O.forward(p, torch.from_numpy(h[-1])).numpy(), state_fn(5))
    assert errs == [0.0, 0.0]
    # a perturbed candidate is caught at the bar
    errs = PIN.check_against(f, lambda h, k: O.forward(p, torch.from_numpy(h[-1])).numpy() * (1 + 2e-3), state_fn(5))
    assert min(errs) > 1e-3


def test_real_backends_say_what_is_missing():
    """Here the reference's stack is absent: the script must say so, not fall back to the stub."""
    for model, backend in (("pangu", "onnxruntime"), ("sfno", "earth2mip"), ("graphcast", "earth2mip")):
        if importlib.util.find_spec(backend) is not None:
            pytest.skip(f"{backend} is importable here: the real pin can be made")
        with pytest.raises(SystemExit, match="cannot pin"):
            PIN.main([model, "--weights", "/nonexistent", "--backend", backend])
    with pytest.raises(SystemExit, match="--weights is required"):
        PIN.main(["pangu"])


def _pinned_fixture(model):
    """(fixture dict, weights path) when a REAL fixture and the weight file it was made from are both present; else skip with the reason."""
    files = sorted(GOLDEN.glob(f"{model}_ref_*.npz"))
    if not files:
        pytest.skip(f"no tests/golden/{model}_ref_*.npz: the reference cannot run in this environment (tools/pin_reference.py) -- parity unpinned")
    f = dict(np.load(files[-1]))
    if not bool(f["pinned"]):
        pytest.skip("fixture written by the stub backend: pins nothing")
    w = os.environ.get(ENV[model])
    if not w or not Path(w).exists():
        pytest.skip(f"{files[-1].name} exists but {ENV[model]} does not name the weight file it was made from")
    if PIN.sha256_of(w) != str(f["weights_sha256"]):
        pytest.skip(f"{ENV[model]} is not the weight file of {files[-1].name} (sha256 differs)")
    return f, w


def test_pangu_oracle_against_the_pinned_reference():
    """CPU: oracle/pangu_oracle.py on the ingested reference weights against the reference's own outputs (1e-3 per channel)."""
    f, w = _pinned_fixture("pangu")
    from oracle import pangu_oracle as O
    from skyrim_amd.pangu.spec import PanguGeometry
    from skyrim_amd.pangu.timeloop import _load_weights
    n_lat, n_lon = int(f["n_lat"]), int(f["n_lon"])
    p = _load_weights(w, PanguGeometry(n_lat, n_lon))
    _, state_fn, _ = PIN.model_spec("pangu", n_lat, n_lon)
    errs = PIN.check_against(f, lambda h, k: O.forward(p, torch.from_numpy(h[-1])).numpy(), state_fn(int(f["seed"])))
    assert max(errs) < 1e-3, errs


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["pangu", "sfno", "graphcast"])
def test_engine_against_the_pinned_reference(model):
    """GPU: the HIP engine, through the reference-shaped TimeLoop and the weight-ingestion path, against the reference's own outputs."""
    f, w = _pinned_fixture(model)
    import datetime
    n_lat, n_lon = int(f["n_lat"]), int(f["n_lon"])
    if model == "pangu":
        from skyrim_amd.pangu.spec import PanguGeometry
        from skyrim_amd.pangu.timeloop import PanguTimeLoop
        loop = PanguTimeLoop(geom=PanguGeometry(n_lat, n_lon))          # params = None: resolved from SKYRIM_PANGU_WEIGHTS
    elif model == "sfno":
        from skyrim_amd.sfno.spec import SfnoConfig
        from skyrim_amd.sfno.timeloop import SfnoTimeLoop
        loop = SfnoTimeLoop(cfg=SfnoConfig(n_lat=n_lat, n_lon=n_lon))
    else:
        from skyrim_amd.graphcast.spec import GraphcastConfig
        from skyrim_amd.graphcast.timeloop import GraphcastTimeLoop
        loop = GraphcastTimeLoop(cfg=GraphcastConfig(n_lat=n_lat, n_lon=n_lon))
    t0 = datetime.datetime(2024, 1, 1)

    def step(hist, k):
        it = iter(loop(t0 + k * loop.time_step, torch.from_numpy(hist[-loop.n_history_levels:]).to(loop.device)[None]))
        next(it)
        return next(it)[1][0].float().cpu().numpy()
    _, state_fn, _ = PIN.model_spec(model, n_lat, n_lon)
    errs = PIN.check_against(f, step, state_fn(int(f["seed"])))
    assert max(errs) < 1e-3, errs
