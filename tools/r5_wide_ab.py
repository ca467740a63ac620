"""Round 5 A/B on the GPU box: the wide row-tile block kernel (csrc/fused_block_wide.hip) against the two-workgroup form, same process, same
weights; SKP_BLK_WIDE / SKP_WIDE_PIPE are read per launch.  Prints ms/step and the largest per-channel difference from the first config."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from skyrim_amd.pangu.engine import PanguEngine
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state

g = PanguGeometry(721, 1440)
p = init_synthetic(g, 0)
x0 = synthetic_state(g, 0)
configs = [("0", "0"), ("2", "0"), ("3", "0"), ("2", "1"), ("3", "1"), ("0", "0")]
steps = int(os.environ.get("AB_STEPS", "10"))
for mode in sys.argv[1:] or ["f16x2m", "f16x1m"]:
    eng = PanguEngine(g, mode)
    eng.load_params(p, calibration="off", rounding="nearest")
    x = x0.to(eng.device)
    ref = None
    for wide, pipe in configs:
        os.environ["SKP_BLK_WIDE"], os.environ["SKP_WIDE_PIPE"] = wide, pipe
        y = eng.step(x)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        d = ((y - ref).abs().amax(dim=(1, 2)) / ref.abs().amax(dim=(1, 2))).max().item()
        for _ in range(3):
            eng.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step(x)
        torch.cuda.synchronize()
        print(f"{mode} wide={wide} pipe={pipe}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step, max rel diff vs first {d:.2e}, finite {bool(torch.isfinite(y).all())}", flush=True)
    del eng
    torch.cuda.empty_cache()
