"""GPU box: what the load-time precision guard (PanguEngine._guard) measures -- sigma-unit error of one step of each plan against the three-term
engine -- on the synthetic weights, on weights with outlier rows, and what the same plans do against the oracle (toy grid).

    python tools/guard_probe.py [--full]
"""
import sys
import time
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import pangu_oracle as O  # noqa: E402
from skyrim_amd.pangu import engine as E  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402
from tools.pangu_outlier_scan import outliers  # noqa: E402


def main():
    full = "--full" in sys.argv
    g = PanguGeometry(721, 1440) if full else PanguGeometry(49, 192)
    params, x = init_synthetic(g, 0), synthetic_state(g, 0)
    cases = [("synthetic", params)] if full else [("synthetic", params), ("rows x5", outliers(params, scale=5.0)), ("rows x30", outliers(params, scale=30.0)),
                                                   ("qkv rows x30", outliers(params, scale=30.0, only="qkv")), ("mlp rows x30", outliers(params, scale=30.0, only="mlp"))]
    for name, p in cases:
        ref = None if full else O.forward(p, x)
        for kw in (dict(), dict(calibration="off"), dict(rounding="nearest")):
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                t0 = time.perf_counter()
                e = E.PanguEngine(g, device="cuda:0")
                e.load_params(p, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            y = e.step(x.cuda()).cpu()
            err = float("nan") if ref is None else O.per_channel_rel_err(y, ref).max().item()
            print(f"{name:14s} {str(kw):28s}: guard {[(hex(a), float(f'{b:.2e}')) for a, b in (e.guard_report or [])]} -> plan {e.term_plan_in_effect:#05x}; "
                  f"vs oracle {err:.3e}; load {dt:.1f} s, guard {getattr(e, 'guard_seconds', 0):.2f} s; warnings {len(w)}", flush=True)
            e.release()
            del e


if __name__ == "__main__":
    main()
