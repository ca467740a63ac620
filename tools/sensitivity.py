"""Per-GEMM operand-rounding sensitivity on the CPU oracle (toy grid): round ONLY the A operand and/or ONLY the weights
of one GEMM class to fp16 (11-bit significand) and report the per-channel max relative error of one step against the
unrounded oracle.  Decides which GEMMs may drop MFMA terms."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import pangu_oracle as O  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402

CLASSES = ["attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2", "down.linear", "up.linear1", "up.linear2"]


def main():
    nlat, nlon = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (49, 192)
    g = PanguGeometry(nlat, nlon)
    p = init_synthetic(g, 0)
    x = synthetic_state(g, 0)
    names = {id(v): k for k, v in p.items()}
    ref = O.forward(p, x)
    orig = O._linear
    policy = {}

    def rd(t, dt):
        return t.to(dt).to(t.dtype)

    def patched(xx, w, b=None, emu=None):
        key = names.get(id(w), "")
        for cls, (ra, rw, layers) in policy.items():
            if cls in key and (layers is None or any(key.startswith(l) for l in layers)):
                if ra is not None:
                    xx = rd(xx, ra)
                if rw is not None:
                    w = rd(w, rw)
        return F.linear(xx, w, b)

    O._linear = patched
    try:
        print(f"grid {nlat}x{nlon}; per-channel max rel err of one step, rounding one operand class to fp16 / bf16")
        for cls in CLASSES:
            row = []
            for ra, rw in ((torch.float16, None), (None, torch.float16), (torch.float16, torch.float16)):
                policy.clear()
                policy[cls] = (ra, rw, None)
                e = O.per_channel_rel_err(O.forward(p, x), ref).max().item()
                row.append(e)
            print(f"  {cls:12s}  A->f16 {row[0]:.2e}   W->f16 {row[1]:.2e}   both {row[2]:.2e}", flush=True)
        for combo in (["attn.qkv"], ["attn.qkv", "mlp.fc1"], ["attn.qkv", "attn.proj"], CLASSES):
            policy.clear()
            for cls in combo:
                policy[cls] = (torch.float16, torch.float16, None)
            e = O.per_channel_rel_err(O.forward(p, x), ref).max().item()
            print(f"  both->f16 for {combo}: {e:.2e}", flush=True)
        for combo in (CLASSES,):
            policy.clear()
            for cls in combo:
                policy[cls] = (None, torch.float16, None)
            e = O.per_channel_rel_err(O.forward(p, x), ref).max().item()
            print(f"  W->f16 for all linears: {e:.2e}", flush=True)
    finally:
        O._linear = orig


if __name__ == "__main__":
    main()
