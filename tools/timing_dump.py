import sys, torch
sys.path.insert(0, '.')
from skyrim_amd.pangu.engine import PanguEngine
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
g = PanguGeometry(721, 1440)
eng = PanguEngine(g, "bf16x3"); eng.load_params(init_synthetic(g, 0))
x = synthetic_state(g, 0).cuda()
eng.step(x); eng.step(x)
raw = eng.debug_buffer("stagger", torch.int32).view(128, 4096)
t = raw[:, :256].view(128, 64, 4).float().cpu()
t2 = raw[:, 2048:2048 + 512].view(128, 64, 8).float().cpu()
names = ["qkv", "proj", "fc1", "fc2"]
# launch order inside a step: blocks -> (qkv, proj, fc1, fc2) dma launches; embed has none; up has 2; recover 2
seq = []
def blk(r): return [f"{n}_r{r}" for n in names]
seq += blk(0) * 2 + blk(1) * 12 + ["up1", "up2"] + blk(0) * 2 + ["rec_s", "rec_u"]
import collections
agg = collections.defaultdict(list)
for i, n in enumerate(seq):
    agg[n].append(t[i])
for n, v in agg.items():
    m = torch.stack(v).mean(0)            # [64 blocks][4]
    m = m.mean(0)
    print(f"{n:8s} cycles/block(sum over its tiles): main {m[0]:10.0f} setup+issue {m[1]:8.0f} epilogue {m[2]:9.0f} drain {m[3]:8.0f}")

print("LN epilogue phases (cycles per block, summed over tiles): rows+issue | pass1 | pass2 | rstd | group0 | group1 | groups2-3")
agg2 = collections.defaultdict(list)
for i, n in enumerate(seq):
    agg2[n].append(t2[i])
for n in ("proj_r0", "fc2_r0", "proj_r1", "fc2_r1", "up1"):
    m = torch.stack(agg2[n]).mean(0).mean(0)
    print(f"{n:8s}", " ".join(f"{v:9.0f}" for v in m[:6]))
