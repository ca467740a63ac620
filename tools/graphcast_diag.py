"""GPU diagnostic for the GraphCast path: engine vs CPU oracle on a small configuration (development aid)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import graphcast_oracle as O  # noqa: E402
from skyrim_amd.graphcast.engine import GraphcastEngine  # noqa: E402
from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states  # noqa: E402

CONFIGS = {
    "tiny": GraphcastConfig(n_lat=33, n_lon=64, splits=2, latent=32, steps=3),
    "small": GraphcastConfig(n_lat=61, n_lon=120, splits=3, latent=64, steps=4),
}
for name in sys.argv[1:] or ["tiny", "small"]:
    cfg = CONFIGS[name]
    eng = GraphcastEngine(cfg)
    g = eng.graph
    p = init_synthetic(cfg, 0)
    x0, x1 = synthetic_states(cfg, 0)
    f = forcings(cfg, 1000.0)
    t0 = time.time()
    ref = O.forward(p, g, x0, x1, f, cfg)
    eng.load_params(p)
    y = eng.step(x0.to(eng.device), x1.to(eng.device), f.to(eng.device))
    torch.cuda.synchronize()
    print(f"{name}: mesh {g.n_mesh} nodes, edges g2m {len(g.g2m_edges)} mesh {len(g.mesh_edges)} m2g {len(g.m2g_edges)}; oracle {time.time() - t0:.1f}s  "
          f"per-channel rel err max {O.per_channel_rel_err(y.cpu(), ref).max().item():.3e}  vs increment {O.increment_rel_err(y.cpu(), ref, x1).max().item():.3e}  "
          f"finite {bool(torch.isfinite(y).all())}", flush=True)
