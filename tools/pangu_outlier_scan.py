"""GPU box: the outlier stress of tests/test_pangu_gpu.py (1 % of the Linear weight rows x scale) across term plans and roundings, toy grid.

    python tools/pangu_outlier_scan.py [scale ...]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import pangu_oracle as O  # noqa: E402
from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402


def outliers(params, seed=3, frac=0.01, scale=30.0, only=None, skip=()):
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in params.items():
        if v.dim() == 2 and k.endswith(".weight") and "bias_table" not in k and "norm" not in k and (only is None or only in k) and not any(t in k for t in skip):
            v = v.clone()
            rows = torch.randperm(v.shape[0], generator=gen)[:max(1, int(frac * v.shape[0]))]
            v[rows] *= scale
        out[k] = v
    return out


def main():
    scales = [float(a) for a in sys.argv[1:] if not a.startswith("--")] or [1.0, 5.0, 30.0]
    g = PanguGeometry(49, 192)
    params, x = init_synthetic(g, 0), synthetic_state(g, 0)
    plans = [(0x6F, "nearest"), (0x6F, "compensated"), (0x66, "nearest"), (0x66, "compensated"), (0x0F, "nearest"), (0x0F, "compensated"),
             (0x06, "nearest"), (0x06, "compensated"), (0x00, "nearest")]
    if "--where" in sys.argv:                       # which Linear class carries the x30 failure (three-term plan and the default plan)
        scales = []
        for name, kw in (("qkv only", dict(only="qkv")), ("all but qkv", dict(skip=("qkv",))), ("proj only", dict(only="attn.proj")), ("mlp only", dict(only="mlp"))):
            p = outliers(params, scale=30.0, **kw)
            ref = O.forward(p, x)
            for plan, rounding in ((0x00, "nearest"), (0x6F, "compensated")):
                e = PanguEngine(g, "f16x3q", "cuda:0", term_plan=plan)
                e.load_params(p, calibration="synthetic" if plan else "off", rounding=rounding)
                y = e.step(x.cuda()).cpu()
                print(f"rows x30 {name:12s} plan {plan:#04x} {rounding:11s}: max per-channel rel err {O.per_channel_rel_err(y, ref).max().item():.3e}", flush=True)
                del e
        for prec in ("f16x3", "bf16x3"):
            p = outliers(params, scale=30.0)
            e = PanguEngine(g, prec, "cuda:0")
            e.load_params(p)
            y = e.step(x.cuda()).cpu()
            print(f"rows x30 all          precision {prec}: max per-channel rel err {O.per_channel_rel_err(y, O.forward(p, x)).max().item():.3e}", flush=True)
            del e
    for sc in scales:
        p = outliers(params, scale=sc) if sc != 1.0 else params
        ref = O.forward(p, x)
        for plan, rounding in plans:
            e = PanguEngine(g, "f16x3q", "cuda:0", term_plan=plan)
            e.load_params(p, calibration="synthetic" if plan else "off", rounding=rounding)
            y = e.step(x.cuda()).cpu()
            err = O.per_channel_rel_err(y, ref).max().item() if torch.isfinite(y).all() else float("nan")
            print(f"rows x{sc:<4g} plan {plan:#04x} {rounding:11s}: max per-channel rel err {err:.3e}", flush=True)
            del e


if __name__ == "__main__":
    main()
