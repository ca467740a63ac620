"""GPU diagnostic for the SFNO path: engine vs CPU oracle on a small configuration (development aid)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import sfno_oracle as O  # noqa: E402
from skyrim_amd.sfno.engine import SfnoEngine  # noqa: E402
from skyrim_amd.sfno.spec import SfnoConfig, init_synthetic, synthetic_state  # noqa: E402

CONFIGS = {
    "tiny": SfnoConfig(n_lat=33, n_lon=64, in_chans=5, out_chans=5, embed_dim=16, num_layers=3, scale_factor=2),
    "small": SfnoConfig(n_lat=97, n_lon=192, in_chans=11, out_chans=11, embed_dim=48, num_layers=4, scale_factor=3),
}

for name in sys.argv[1:] or ["tiny", "small"]:
    cfg = CONFIGS[name]
    p, x = init_synthetic(cfg, 0), synthetic_state(cfg, 0)
    t0 = time.time()
    ref = O.forward(p, x, cfg)
    e2 = SfnoEngine(cfg, terms=2)
    e2.load_params(p)
    print(f"{name}: terms=2 per-channel rel err max {O.per_channel_rel_err(e2.step(x.to(e2.device)).cpu(), ref).max().item():.3e}", flush=True)
    eng = SfnoEngine(cfg)
    eng.load_params(p)
    y = eng.step(x.to(eng.device))
    torch.cuda.synchronize()
    err = O.per_channel_rel_err(y.cpu(), ref)
    print(f"{name}: lmax {cfg.lmax} mmax {cfg.mmax}  oracle {time.time() - t0:.1f}s  per-channel rel err max {err.max().item():.3e} "
          f"median {err.median().item():.3e}  finite {bool(torch.isfinite(y).all())}", flush=True)
    xs = x.to(eng.device).clone()
    xr = x
    for _ in range(3):
        eng.step(xs, xs)
        xr = O.forward(p, xr, cfg)
    print(f"   3-step rollout (in place) err {O.per_channel_rel_err(xs.cpu(), xr).max().item():.3e}", flush=True)
