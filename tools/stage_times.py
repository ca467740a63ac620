"""GPU box: the default engine at 721x1440 as loaded (calibration + guard) -- ms per step, per-stage times from the engine's own HIP events,
what the load-time guard measured, and a checksum of the output (two builds that should agree bit for bit print the same ``mean_abs``).
Environment switches of the kernels under test are read once per process: run it once per setting.

    python tools/stage_times.py
"""
import json
import os
import sys
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402


def main():
    g = PanguGeometry(721, 1440)
    params = init_synthetic(g, 0)
    x = synthetic_state(g, 0).cuda()
    eng = PanguEngine(g, device="cuda:0")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.load_params(params)
    for _ in range(3):
        eng.step(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.step(x)
    b.record()
    torch.cuda.synchronize()
    eng.profile(True)
    y = x.clone()
    for _ in range(5):
        y = eng.step(x)
    torch.cuda.synchronize()
    prof = {r["name"]: round(r["total_ms"] / 5, 4) for r in eng.profile_read()}
    eng.profile(False)
    y_sum = float(y.double().abs().mean())
    print(json.dumps({"ms_per_step": round(a.elapsed_time(b) / 10, 3), "plan": hex(eng.term_plan_in_effect),
                      "guard": [(hex(p), round(e, 6)) for p, e in eng.guard_report], "warnings": [str(i.message)[:80] for i in w],
                      "mean_abs": y_sum, "stage_ms_per_step": prof}, default=str))


if __name__ == "__main__":
    main()
