"""GPU box: the fused QKV + attention kernel (csrc/attention.hip: qkv_attention_kernel) against the two-launch form (SKP_SPLIT_ATTN=1) --
same network, same weights: difference of one step on the toy grid (and against the oracle), per-stage times at 721x1440.

    python tools/qa_check.py [--full-only]
"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402


def engine(g, params, split, **kw):
    if split:
        os.environ["SKP_SPLIT_ATTN"] = "1"
    else:
        os.environ.pop("SKP_SPLIT_ATTN", None)
    e = PanguEngine(g, device="cuda:0", **kw)
    e.load_params(params, guard=False)
    os.environ.pop("SKP_SPLIT_ATTN", None)
    return e


def main():
    if "--full-only" not in sys.argv:
        from oracle import pangu_oracle as O
        g = PanguGeometry(49, 192)
        params, x = init_synthetic(g, 0), synthetic_state(g, 0)
        ref = O.forward(params, x)
        for kw in (dict(), dict(precision="f16x2m"), dict(precision="f16x3q")):
            ys = {}
            for split in (True, False):
                e = engine(g, params, split, **kw)
                ys[split] = e.step(x.cuda()).cpu()
                e.release()
            d = O.per_channel_rel_err(ys[False], ys[True]).max().item()
            print(f"toy {kw}: fused vs split {d:.3e}; vs oracle fused {O.per_channel_rel_err(ys[False], ref).max().item():.3e} split {O.per_channel_rel_err(ys[True], ref).max().item():.3e}", flush=True)
    g = PanguGeometry(721, 1440)
    params, x = init_synthetic(g, 0), synthetic_state(g, 0)
    outs = {}
    for split in (True, False):
        e = engine(g, params, split)
        xs = x.cuda()
        y = e.step(xs)
        outs[split] = y.cpu()
        torch.cuda.synchronize()
        e.profile(True)
        for _ in range(5):
            e.step(xs, out=y)
        st = e.profile_read()
        e.profile(False)
        tot = sum(s["total_ms"] for s in st) / 5
        print(("split" if split else "fused") + f": {tot:.2f} ms/step; " + ", ".join(f"{s['name']} {s['total_ms'] / max(1, s['launches']):.3f}x{s['launches'] // 5}" for s in st if s["launches"]), flush=True)
        e.release()
    num = (outs[False].double() - outs[True].double()).abs().flatten(1).max(1).values
    den = outs[True].double().abs().flatten(1).max(1).values
    print(f"full size: fused vs split, max per-channel rel diff {(num / den).max().item():.3e}, finite {bool(torch.isfinite(outs[False]).all())}")


if __name__ == "__main__":
    main()
