"""Time the per-stage kernels of a library variant (SKYRIM_PANGU_LIB) on the full grid; prints ms/launch per stage."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3q"
g = PanguGeometry(721, 1440)
eng = PanguEngine(g, prec)
eng.load_params(init_synthetic(g, 0))
x = synthetic_state(g, 0).to(eng.device)
for _ in range(2):
    eng.step(x)
eng.profile(True)
for _ in range(4):
    eng.step(x)
torch.cuda.synchronize()
st = eng.profile_read()
print(" ".join(f"{s['name']}={s['total_ms'] / max(s['launches'], 1):.4f}" for s in st if s["launches"]))
