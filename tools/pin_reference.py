#!/usr/bin/env python
"""Pin the oracles (and through them the kernels) to the REFERENCE's own arithmetic -- on a box where it can run.

The arithmetic of the reference path lives in packages and weight files that are absent from this build environment
(/root/reference/requirements.txt:1-4: earth2mip -> onnxruntime / torch-harmonics / jax + deepmind-graphcast; weights fetched by
``earth2mip.registry.get_model("e2mip://...")``, /root/reference/skyrim/core/models/pangu.py:46, fourcastnet_v2.py:37, graphcast.py:52-54).
Every parity claim of this repository is therefore "unpinned": engine == oracle, oracle == published algorithm as restated here.  This
script closes the gap wherever the real thing IS importable:

    python tools/pin_reference.py pangu     --weights pangu_weather_6.onnx          [--backend onnxruntime]
    python tools/pin_reference.py sfno      --weights <fcnv2_sm package directory>  [--backend earth2mip]
    python tools/pin_reference.py graphcast --weights <graphcast package directory> [--backend earth2mip]

It runs the REAL model on the repository's seeded synthetic state (``spec.synthetic_state``, the state every test uses) for ``--steps``
6-h steps and writes ``tests/golden/<model>_ref_<lat>x<lon>.npz``: a strided sample of every output channel, the per-channel maxima, the
seed, the backend's name / version and the SHA-256 of the weight file.  ``tests/test_pinned_reference.py`` picks such a fixture up when it
exists AND the same weight file is reachable (``SKYRIM_<MODEL>_WEIGHTS``): the oracle (CPU) and the HIP engine (GPU) are then run on
the same weights and state and held to the fixture at the 1e-3 per-channel bar -- that is the pin.  Fixtures are DATA (inputs are
regenerated from the seed, outputs are a sample); nothing of the reference's source is copied.

``--backend stub`` writes a fixture from the repository's OWN oracle on synthetic weights.  It pins nothing (``pinned = False`` in the
file) and exists so that the file format, the sampling and the pick-up logic are exercised by the CPU tests.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SUB = (6, 16)                 # latitude / longitude stride of the stored sample (the stride of tests/golden/pangu_toy_49x192.npz)


class BackendUnavailable(RuntimeError):
    pass


def _need(module: str):
    try:
        return importlib.import_module(module)
    except Exception as e:      # ImportError, or a broken install
        raise BackendUnavailable(f"backend needs `{module}`, which is not importable here ({type(e).__name__}: {e})") from e


def sha256_of(path) -> str:
    p = Path(path)
    h = hashlib.sha256()
    files = [p] if p.is_file() else sorted(q for q in p.rglob("*") if q.is_file())
    for f in files:
        with open(f, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 24), b""):
                h.update(chunk)
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------------------------------------------- #
#  model descriptions: grid, seeded state, channel count
# ---------------------------------------------------------------------------------------------------------------------------------- #
def model_spec(model: str, n_lat: int, n_lon: int):
    """-> (n_channels, state(seed) -> ndarray (n_history, C, lat, lon) float32, n_history)."""
    if model == "pangu":
        from skyrim_amd.pangu.spec import PanguGeometry, synthetic_state
        g = PanguGeometry(n_lat, n_lon)
        return 69, (lambda seed: synthetic_state(g, seed).numpy()[None]), 1
    if model == "sfno":
        from skyrim_amd.sfno.spec import SfnoConfig, synthetic_state
        cfg = SfnoConfig(n_lat=n_lat, n_lon=n_lon)
        return cfg.in_chans, (lambda seed: synthetic_state(cfg, seed).numpy()[None]), 1
    if model == "graphcast":
        from skyrim_amd.graphcast.spec import GraphcastConfig, synthetic_states
        cfg = GraphcastConfig(n_lat=n_lat, n_lon=n_lon)
        return cfg.n_vars, (lambda seed: np.stack([t.numpy() for t in synthetic_states(cfg, seed)])), 2
    raise SystemExit(f"unknown model {model!r}")


# ---------------------------------------------------------------------------------------------------------------------------------- #
#  backends: each returns (step_fn(history ndarray (n_hist, C, lat, lon), k) -> next state (C, lat, lon), description, pinned)
# ---------------------------------------------------------------------------------------------------------------------------------- #
def backend_onnxruntime(model, weights, n_lat, n_lon, args):
    """Pangu's own graph (pangu_weather_6.onnx) under onnxruntime: inputs ``input`` (5, 13, lat, lon) [z, q, t, u, v x 1000..50 hPa] and
    ``input_surface`` (4, lat, lon) [msl, u10m, v10m, t2m] in physical units -- the split of the 69-channel state the reference documents
    (/root/reference/skyrim/core/models/pangu.py:6-13, 32-36)."""
    if model != "pangu":
        raise BackendUnavailable("the onnxruntime backend runs Pangu's ONNX graph only")
    ort = _need("onnxruntime")
    sess = ort.InferenceSession(str(weights), providers=args.providers.split(","))
    names = [i.name for i in sess.get_inputs()]

    def step(hist, k):
        x = hist[-1]
        feed = {names[0]: x[:65].reshape(5, 13, n_lat, n_lon).astype(np.float32), names[1]: x[65:].astype(np.float32)}
        up, sf = sess.run(None, feed)
        return np.concatenate([np.asarray(up).reshape(65, n_lat, n_lon), np.asarray(sf).reshape(4, n_lat, n_lon)]).astype(np.float32)
    return step, f"onnxruntime {ort.__version__} ({','.join(sess.get_providers())})", True


def backend_earth2mip(model, weights, n_lat, n_lon, args):
    """The reference's own call: an earth2mip TimeLoop built from a local package directory, driven exactly as
    ``run_basic_inference`` drives it (/root/reference/skyrim/core/models/utils.py:29-40); GraphCast through ``stepper`` like
    /root/reference/skyrim/core/models/graphcast.py:102-118 is left to its TimeLoop ``__call__`` here (same states)."""
    e2 = _need("earth2mip")
    torch = _need("torch")
    import datetime
    registry = _need("earth2mip.registry")
    net = _need({"pangu": "earth2mip.networks.pangu", "sfno": "earth2mip.networks.fcnv2_sm", "graphcast": "earth2mip.networks.graphcast"}[model])
    package = registry.get_model(str(weights))
    loop = net.load_time_loop_operational(package) if model == "graphcast" else net.load(package)
    t0 = datetime.datetime(2024, 1, 1)

    def step(hist, k):
        x = torch.from_numpy(hist[-loop.n_history_levels:]).to(loop.device)[None]
        it = iter(loop(t0 + k * loop.time_step, x))
        next(it)                                            # the echo of the initial state
        _, out, _ = next(it)
        return out[0].float().cpu().numpy()
    return step, f"earth2mip {getattr(e2, '__version__', '?')}", True


def backend_stub(model, weights, n_lat, n_lon, args):
    """The repository's OWN oracle on seeded synthetic weights: NOT a pin -- exercises the fixture format and the pick-up logic."""
    import torch
    seed = args.seed
    if model == "pangu":
        from oracle import pangu_oracle as O
        from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
        p = init_synthetic(PanguGeometry(n_lat, n_lon), seed)
        fn = lambda h, k: O.forward(p, torch.from_numpy(h[-1])).numpy()                       # noqa: E731
    elif model == "sfno":
        from oracle import sfno_oracle as S
        from skyrim_amd.sfno.spec import SfnoConfig, init_synthetic
        cfg = SfnoConfig(n_lat=n_lat, n_lon=n_lon, **args.stub_cfg)
        p = init_synthetic(cfg, seed)
        fn = lambda h, k: S.forward(p, torch.from_numpy(h[-1]), cfg).numpy()                  # noqa: E731
    else:
        raise BackendUnavailable("stub backend: pangu and sfno only")
    return fn, "stub: this repository's oracle on synthetic weights (self-oracle, pins nothing)", False


BACKENDS = {"onnxruntime": backend_onnxruntime, "earth2mip": backend_earth2mip, "stub": backend_stub}
DEFAULT_BACKEND = {"pangu": "onnxruntime", "sfno": "earth2mip", "graphcast": "earth2mip"}


# ---------------------------------------------------------------------------------------------------------------------------------- #
def sample(y: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(y[:, ::SUB[0], ::SUB[1]])


def fixture_path(model: str, n_lat: int, n_lon: int, out_dir=None) -> Path:
    return Path(out_dir or ROOT / "tests" / "golden") / f"{model}_ref_{n_lat}x{n_lon}.npz"


def make_fixture(model, step_fn, state0, n_steps):
    hist, subs, maxs = state0.astype(np.float32), [], []
    for k in range(n_steps):
        y = np.asarray(step_fn(hist, k), dtype=np.float32)
        if y.shape != hist.shape[1:]:
            raise RuntimeError(f"backend returned {y.shape}, expected {hist.shape[1:]}")
        subs.append(sample(y))
        maxs.append(np.abs(y).reshape(y.shape[0], -1).max(1))
        hist = np.concatenate([hist[1:], y[None]]) if hist.shape[0] > 1 else y[None]
    return np.stack(subs), np.stack(maxs)


def check_against(fixture: dict, step_fn, state0) -> list[float]:
    """Per step: max over channels of max|candidate - reference| / max|reference| on the stored sample.  ``step_fn`` as for a backend.
    The candidate is rolled out on ITS OWN states (free-running), like the fixture was."""
    hist, errs = state0.astype(np.float32), []
    for k in range(int(fixture["steps"])):
        y = np.asarray(step_fn(hist, k), dtype=np.float32)
        d = np.abs(sample(y).astype(np.float64) - fixture["sub"][k]).reshape(y.shape[0], -1).max(1)
        errs.append(float((d / fixture["absmax"][k]).max()))
        hist = np.concatenate([hist[1:], y[None]]) if hist.shape[0] > 1 else y[None]
    return errs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model", choices=["pangu", "sfno", "graphcast"])
    ap.add_argument("--weights", help="the reference's weight file / package directory (not needed for --backend stub)")
    ap.add_argument("--backend", choices=sorted(BACKENDS))
    ap.add_argument("--grid", default="721x1440")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", help="directory of the fixture (default tests/golden)")
    ap.add_argument("--providers", default="CPUExecutionProvider", help="onnxruntime execution providers, comma separated")
    args = ap.parse_args(argv)
    args.stub_cfg = getattr(args, "stub_cfg", {})
    n_lat, n_lon = (int(v) for v in args.grid.split("x"))
    backend = args.backend or DEFAULT_BACKEND[args.model]
    if backend != "stub" and not args.weights:
        raise SystemExit("--weights is required for a real backend")
    try:
        step_fn, desc, pinned = BACKENDS[backend](args.model, args.weights, n_lat, n_lon, args)
    except BackendUnavailable as e:
        raise SystemExit(f"cannot pin {args.model} here: {e}.  Run this script where the reference's dependencies and weights are installed.")
    C, state_fn, n_hist = model_spec(args.model, n_lat, n_lon)
    state0 = state_fn(args.seed)
    sub, absmax = make_fixture(args.model, step_fn, state0, args.steps)
    out = fixture_path(args.model, n_lat, n_lon, args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(out, model=args.model, backend=desc, pinned=pinned, n_lat=n_lat, n_lon=n_lon, seed=args.seed, steps=args.steps,
                        stride=np.array(SUB), sub=sub, absmax=absmax, weights_sha256=sha256_of(args.weights) if args.weights else "")
    print(f"{out}: {args.model} {n_lat}x{n_lon}, {args.steps} step(s), backend {desc}, pinned={pinned}, {out.stat().st_size / 1e6:.1f} MB")
    return out


if __name__ == "__main__":
    main()
