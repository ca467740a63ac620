"""Round 5, GPU box: the one-term coarse layers (f16x1m) at 721x1440 over the 24-h rollout against the oracle, for several choices of what the
compensated rounding is fitted on (the calibration state + N of its successive forecasts)."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import _oracle_jobs as J
from oracle import pangu_oracle as O
from skyrim_amd.pangu.engine import PanguEngine
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state

t0 = time.time()
J.start(["pangu_full_rollout4"])
g = PanguGeometry(721, 1440)
params, x = init_synthetic(g, 0), synthetic_state(g, 0)
outs = {}
for mode, nf in (("f16x1m", 1), ("f16x1m", 2), ("f16x1m", 3), ("f16x2m", 1)):
    e = PanguEngine(g, mode, "cuda:0")
    e.calibration_forecasts = nf
    t1 = time.time()
    e.load_params(params)
    state = x.cuda().clone()
    ys = []
    for k in range(4):
        e.step(state, out=state)
        ys.append(state.cpu().clone())
    outs[(mode, nf)] = ys
    print(f"{mode} forecasts={nf}: load {time.time() - t1:.1f} s", flush=True)
    del e
    torch.cuda.empty_cache()
ref = J.PanguRollout()
std = params["norm.std"]
for key, ys in outs.items():
    errs = [O.per_channel_rel_err(ys[k], ref[k]).max().item() for k in range(4)]
    sig = [O.per_channel_sigma_err(ys[k], ref[k], std).max().item() for k in range(4)]
    print(key, "rel " + " ".join(f"{v:.3e}" for v in errs), "| sigma " + " ".join(f"{v:.2e}" for v in sig), flush=True)
print(f"total {time.time() - t0:.0f} s")
J.stop()
