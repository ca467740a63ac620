import sys, time, torch
sys.path.insert(0, '.')
import os
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")
from oracle import pangu_oracle as O
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
g = PanguGeometry(721, 1440)
p, x = init_synthetic(g, 0), synthetic_state(g, 0)
for n in (128, 64, 32, 16):
    torch.set_num_threads(n)
    t0 = time.time()
    with torch.no_grad():
        y = O.forward(p, x)
    print(f"threads {n}: {time.time() - t0:.1f} s", flush=True)
