"""GPU box: a candidate term plan against the default at 721x1440 -- load-time guard figure, 24-h rollout against the oracle's golden vectors
(tests/golden/full_pangu.npz: lattice + cell means), step time.

    python tools/plan_probe.py 0x66F 0x6FF ...
"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(ROOT / "tests"))
from _golden_full import FullSizeGolden  # noqa: E402
from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402


def main():
    plans = [int(a, 16) for a in sys.argv[1:]] or [0x66F, 0x6FF]
    g = PanguGeometry(721, 1440)
    params, x = init_synthetic(g, 0), synthetic_state(g, 0)
    gold = FullSizeGolden("pangu")
    for plan in plans:
        e = PanguEngine(g, "f16x3q", "cuda:0", term_plan=plan)
        t0 = time.perf_counter()
        e.load_params(params)
        load = time.perf_counter() - t0
        state, errs = x.cuda().clone(), []
        for k in range(gold.steps):
            e.step(state, out=state)
            r = gold.errors(k, state)
            errs.append((float(r["rel"].max()), float(r["cell"].max())))
        xs = x.cuda().clone()
        for _ in range(3):
            e.step(xs, out=xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.step(xs, out=xs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print(f"plan {plan:#05x}: guard {[(hex(a), float(f'{b:.3e}')) for a, b in (e.guard_report or [])]} -> {e.term_plan_in_effect:#05x}; rollout rel " +
              " ".join(f"{a:.3e}" for a, _ in errs) + "; cells " + " ".join(f"{c:.2e}" for _, c in errs) + f"; {ms:.2f} ms/step; load {load:.1f} s", flush=True)
        e.release()


if __name__ == "__main__":
    main()
