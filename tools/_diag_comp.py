import sys, torch
sys.path.insert(0, '.')
from oracle import pangu_oracle as O
from skyrim_amd.pangu.engine import PanguEngine, calibration_state
from skyrim_amd.pangu.calibration import calibrated_params, engine_taps, KINDS
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
g = PanguGeometry(49, 192); p = init_synthetic(g, 0); x = synthetic_state(g, 0)
ref = O.forward(p, x)
def err(y, r=ref): return O.per_channel_rel_err(y.cpu(), r.cpu() if hasattr(r, 'cpu') else r).max().item()
def run(prec, params, **kw):
    e = PanguEngine(g, prec, "cuda:0"); e.load_params(params, **kw); return e.step(x.cuda()).cpu()
wn = dict(p)
for k in p:
    if k.endswith(".weight") and any(s in k for s in KINDS): wn[k] = p[k].half().float()
a = run("f16x3q", wn); b = run("f16x2q", wn, calibration="off"); b0 = run("f16x2q", p, calibration="off")
print("3-term on nearest weights vs oracle", err(a), " 2-term on nearest weights", err(b), " 2-term on W", err(b0), " a vs b", err(b, a), flush=True)
tap = PanguEngine(g, "f16x3q", "cuda:0", mlp="split"); tap.load_params(p)
cs = calibration_state(g, p["norm.mean"], p["norm.std"])
for plan in (0x0F, 0xF0, 0xFF):
    for rounding in ("nearest", "compensated"):
        q = calibrated_params(p, plan, engine_taps(tap, p, cs), rounding=rounding)
        c = run("f16x3q", q); d = PanguEngine(g, "f16x3q", "cuda:0", term_plan=plan); d.load_params(q, calibration="off"); d = d.step(x.cuda()).cpu()
        oq = O.forward(q, x)
        print(hex(plan), rounding, "oracle(q) vs oracle(p)", err(oq), " 3-term engine on q", err(c), " plan engine on q", err(d), " plan vs 3-term", err(d, c), flush=True)
