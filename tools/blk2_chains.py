"""GPU box: one Pangu library (SKYRIM_PANGU_LIB, default the in-tree one) -- toy-grid parity of the default mode against the oracle (unless
--no-parity) and the sustained step time at 721x1440.  Used to A/B build-time variants of csrc/fused_block2.hip, e.g. SKP_BLK2_CHAINS4:
    bash tools/blk2_chains.sh"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")
import torch  # noqa: E402

from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402

tag = os.path.basename(os.environ.get("SKYRIM_PANGU_LIB", "in-tree"))
if "--no-parity" not in sys.argv:
    from oracle import pangu_oracle as O
    g = PanguGeometry(49, 192)
    p, x = init_synthetic(g, 0), synthetic_state(g, 0)
    e = PanguEngine(g, device="cuda:0")
    e.load_params(p)
    print(tag, "toy parity %.3e" % O.per_channel_rel_err(e.step(x.cuda()).cpu(), O.forward(p, x)).max().item(), flush=True)
    del e
g = PanguGeometry(721, 1440)
e = PanguEngine(g, device="cuda:0")
e.load_params(init_synthetic(g, 0), calibration="off")
x = synthetic_state(g, 0).cuda()
for _ in range(4):
    e.step(x, x)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(12):
    e.step(x, x)
torch.cuda.synchronize()
print(tag, "%.3f ms/step, finite %s" % ((time.perf_counter() - t) / 12 * 1e3, bool(torch.isfinite(x).all())), flush=True)
