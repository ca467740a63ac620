"""Two-term block kernel (csrc/fused_block2.hip): parity on the toy grid and ms per launch at 721x1440 for the schedule variants
(SKP_BLK2_VARIANT is read once per process: this script re-executes itself per variant).  GPU box only."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(prec):
    import time
    import torch
    from oracle import pangu_oracle as O
    from skyrim_amd.pangu.engine import PanguEngine
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
    out = {"precision": prec, "variant": os.environ.get("SKP_BLK2_VARIANT", "-") + ":" + os.environ.get("SKP_DUO_STAGGER", "")}
    if os.environ.get("SKEW_TOY", "1") == "1" and not (10 <= int(os.environ.get("SKP_BLK2_VARIANT", "0")) % 100 < 20):
        g = PanguGeometry(49, 192)
        p, x = init_synthetic(g, 0), synthetic_state(g, 0)
        e = PanguEngine(g, prec)
        e.load_params(p)
        ref = O.rollout(p, x, 4)
        xs = x.cuda().clone()
        errs = []
        for k in range(4):
            e.step(xs, xs)
            errs.append(O.per_channel_rel_err(xs.cpu(), ref[k]).max().item())
        out["toy_err_4_steps"] = errs
        del e
    if os.environ.get("SKEW_FULL", "1") == "1":
        g = PanguGeometry(721, 1440)
        p, x = init_synthetic(g, 0), synthetic_state(g, 0)
        e = PanguEngine(g, prec)
        e.load_params(p)
        xs = x.cuda()
        for _ in range(2):
            e.step(xs, xs)
        e.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            e.step(xs, xs)
        torch.cuda.synchronize()
        out["ms_per_step"] = 1e3 * (time.perf_counter() - t0) / 6
        out["stages_ms_per_launch"] = {s["name"]: round(s["total_ms"] / s["launches"], 4) for s in e.profile_read() if s["launches"]}
        out["finite"] = bool(torch.isfinite(xs).all())
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    runs = [(os.environ.get("SKEW_PREC", "f16x2"), v) for v in sys.argv[1:] or ("0", "1", "2", "3", "4")]      # "5:40000" = variant 5 with SKP_DUO_STAGGER=40000
    for prec, v in runs:
        env = dict(os.environ)
        if v is not None:
            env["SKP_BLK2_VARIANT"] = v.split(":")[0]
            if ":" in v:
                env["SKP_DUO_STAGGER"] = v.split(":")[1]
        r = subprocess.run([sys.executable, __file__, "child", prec], env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(lines[-1] if lines else f"FAILED {prec} {v}: rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}", flush=True)
