"""Full-size SFNO step on the GPU: time per step, finiteness, determinism (no oracle at this size)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skyrim_amd.sfno.engine import SfnoEngine  # noqa: E402
from skyrim_amd.sfno.spec import SfnoConfig, flops_per_step, init_synthetic, synthetic_state  # noqa: E402

cfg = SfnoConfig()
t0 = time.time()
p, x = init_synthetic(cfg, 0), synthetic_state(cfg, 0)
print(f"init {time.time() - t0:.1f}s", flush=True)
eng = SfnoEngine(cfg, terms=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
t0 = time.time()
eng.load_params(p)
print(f"prepare {time.time() - t0:.1f}s  mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
xd = x.to(eng.device)
y = eng.step(xd)
torch.cuda.synchronize()
y2 = eng.step(xd)
print("finite", bool(torch.isfinite(y).all()), "deterministic", bool(torch.equal(y, y2)), "max|y|", float(y.abs().max()), flush=True)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    eng.step(xd, xd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{1e3 * dt:.1f} ms/step  {1 / dt:.2f} steps/s  {flops_per_step(cfg) / dt / 1e12:.1f} TFLOP/s algorithmic ({flops_per_step(cfg) / 1e12:.2f} TF/step), "
      f"{eng.launches_per_step()} launches, finite {bool(torch.isfinite(xd).all())}", flush=True)
eng.profiling = True
for _ in range(3):
    eng.step(xd, xd)
st = eng.profile_read()
tot = sum(d["total_ms"] for d in st)
for d in sorted(st, key=lambda d: -d["total_ms"]):
    ms = d["total_ms"] / 3
    print(f"  {d['name']:22s} {d['launches'] // 3:3d} launches  {ms:7.3f} ms/step ({100 * d['total_ms'] / tot:4.1f} %)  {d['flops'] / 3 / ms / 1e9:8.1f} TFLOP/s dense  {d['bytes'] / 3 / ms / 1e6:7.1f} GB/s", flush=True)
