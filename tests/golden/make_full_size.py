"""Generates tests/golden/full_<model>.npz: the CPU oracles' autoregressive rollouts at BASELINE.json's full sizes (721x1440), reduced to
lattice samples + whole-field maxima + cell means (tests/_golden_full.py says what and why).  One oracle step is one to two minutes on 128
host threads, so this runs on a many-core host (the GPU box's CPU: `gpurun -- python tests/golden/make_full_size.py all`), never in a test.

SELF-ORACLE, REFERENCE PARITY UNPINNED: the vectors pin the oracles at full size and stand in for their live runs in
`pytest -m gpu`; inputs and parameters are regenerated from seeds by the tests, outputs are stored here.

    python tests/golden/make_full_size.py pangu|sfno|graphcast|all [steps=4]
"""
import hashlib
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(ROOT / "tests"))
os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")
from _golden_full import lattice, reduce_field  # noqa: E402


def oracle_hash(*names):
    h = hashlib.sha256()
    for n in names:
        h.update((ROOT / "oracle" / n).read_bytes())
    return h.hexdigest()[:16]


def pangu(steps):
    from oracle import pangu_oracle as O
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
    g = PanguGeometry(721, 1440)
    params, x = init_synthetic(g, 0), synthetic_state(g, 0)
    for _ in range(steps):
        y = O.forward(params, x)
        yield y, x
        x = y
    return


def sfno(steps):
    from oracle import sfno_oracle as O
    from skyrim_amd.sfno.spec import SfnoConfig, init_synthetic, synthetic_state
    cfg = SfnoConfig()
    params, x = init_synthetic(cfg, 0), synthetic_state(cfg, 0)
    tr = O.Transforms(cfg)
    for _ in range(steps):
        y = O.forward(params, x, cfg, tr=tr)
        yield y, x
        x = y


def graphcast(steps):
    from oracle import graphcast_graph as OG
    from oracle import graphcast_oracle as O
    from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states
    cfg = GraphcastConfig()
    p, og = init_synthetic(cfg, 0), OG.build(cfg.n_lat, cfg.n_lon, cfg.splits)
    a, b = synthetic_states(cfg, 0)
    for k in range(steps):
        y = O.forward(p, og, a, b, forcings(cfg, 1000.0 + 6.0 * k))
        yield y, b
        a, b = b, y


MODELS = {"pangu": (pangu, ("pangu_oracle.py",)), "sfno": (sfno, ("sfno_oracle.py",)), "graphcast": (graphcast, ("graphcast_oracle.py", "graphcast_graph.py"))}


def make(model, steps):
    fn, srcs = MODELS[model]
    parts, t0 = [], time.time()
    with torch.no_grad():
        for k, (y, prev) in enumerate(fn(steps)):
            assert torch.isfinite(y).all()
            parts.append(reduce_field(y, prev))
            print(f"{model}: oracle step {k + 1}/{steps} at {time.time() - t0:.0f} s", flush=True)
            shape = tuple(y.shape)
    ii, jj = lattice(shape[-2], shape[-1])
    meta = (f"{model} oracle rollout, {steps} steps, state {shape}, seeds 0; oracle sources {oracle_hash(*srcs)}; torch {torch.__version__}; "
            f"{torch.get_num_threads()} threads; {time.strftime('%Y-%m-%dT%H:%MZ', time.gmtime())}")
    out = {k: np.stack([p[k] for p in parts]) for k in ("samples", "absmax", "incmax", "cells")}
    np.savez_compressed(Path(__file__).parent / f"full_{model}.npz", ii=ii, jj=jj, meta=np.array(meta), **out)
    print(meta, {k: v.shape for k, v in out.items()}, flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for m in (MODELS if which == "all" else [which]):
        make(m, steps)
