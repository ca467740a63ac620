"""Full-size GraphCast step on the GPU: time per step, finiteness, determinism, per-stage times (no oracle at this size)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skyrim_amd.graphcast.engine import GraphcastEngine  # noqa: E402
from skyrim_amd.graphcast.spec import GraphcastConfig, flops_per_step, forcings, init_synthetic, synthetic_states  # noqa: E402

cfg = GraphcastConfig()
t0 = time.time()
eng = GraphcastEngine(cfg)
g = eng.graph
print(f"graph {time.time() - t0:.1f}s: mesh {g.n_mesh}, edges g2m {len(g.g2m_edges)} mesh {len(g.mesh_edges)} m2g {len(g.m2g_edges)}", flush=True)
t0 = time.time()
p = init_synthetic(cfg, 0)
x0, x1 = synthetic_states(cfg, 0)
f = forcings(cfg, 1000.0)
eng.load_params(p)
print(f"init + prepare {time.time() - t0:.1f}s  mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
a, b, fd = x0.to(eng.device), x1.to(eng.device), f.to(eng.device)
y = eng.step(a, b, fd)
y2 = eng.step(a, b, fd)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(y).all()), "deterministic", bool(torch.equal(y, y2)), flush=True)
n = 3
t0 = time.perf_counter()
for _ in range(n):
    nxt = eng.step(a, b, fd)
    a, b = b, nxt
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = flops_per_step(cfg, g.n_grid, g.n_mesh, len(g.mesh_edges), len(g.g2m_edges), len(g.m2g_edges))
print(f"{1e3 * dt:.1f} ms/step  {1 / dt:.2f} steps/s  {fl / dt / 1e12:.1f} TFLOP/s ({fl / 1e12:.1f} TF/step), finite {bool(torch.isfinite(b).all())}", flush=True)
eng.profiling = True
eng.step(a, b, fd)
st = eng.profile_read()
tot = sum(d["total_ms"] for d in st)
for d in sorted(st, key=lambda d: -d["total_ms"]):
    print(f"  {d['name']:16s} {d['launches']:3d} launches {d['total_ms']:8.2f} ms ({100 * d['total_ms'] / tot:4.1f} %)  {d['flops'] / max(d['total_ms'], 1e-9) / 1e9:7.1f} TFLOP/s", flush=True)
