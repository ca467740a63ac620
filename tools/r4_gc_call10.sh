#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/node_dbg3.log
import torch, numpy as np
from skyrim_amd import ops
from skyrim_amd.graphcast import fused as fz
L=512
def f16(x): return x.to(torch.float16).to(torch.float64)
dev = torch.device("cuda:0")
for n_src, rows in ((1, 4096), (2, 4096)):
    gen = torch.Generator().manual_seed(20 + n_src)
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    srcs = [3.0 * r(rows, L).float() for _ in range(n_src)]
    w1, w2 = (r(L, L * n_src) / (L * n_src) ** 0.5).float(), (r(L, L) / L ** 0.5).float()
    b1, b2, gamma, beta = (0.1 * r(L)).float(), (0.1 * r(L)).float(), (1 + 0.1 * r(L)).float(), (0.1 * r(L)).float()
    x = torch.cat(srcs, dim=1).double()
    y = torch.nn.functional.layer_norm(torch.nn.functional.silu(x @ w1.double().T + b1.double()) @ w2.double().T + b2.double(), (L,), gamma.double(), beta.double(), 1e-5)
    want = srcs[0].double() + y
    sd = [s.to(dev) for s in srcs]
    w1f, w2f = fz.prep_w1_fragments(w1.to(dev)), fz.prep_w2_fragments(w2.to(dev))
    tab = [t.to(dev) for t in (b1, b2, gamma, beta)]
    out = torch.zeros(rows, L, device=dev)
    ops.hip.gc_node_mlp(sd, [0] * n_src, [L] * n_src, w1f, w2f, *tab, sd[0], 0, L, out, 0, L, rows)
    torch.cuda.synchronize()
    err = (out.cpu().double() - want).abs()
    per_row = err.amax(1)
    bad = (per_row > 3e-5).nonzero().flatten().tolist()
    print("node", n_src, "rows", rows, "bad rows", len(bad), [(b, b // 64, (b % 64) // 16, b % 16) for b in bad[:24]])
    for b in bad[:4]:
        e = err[b]
        print("    row", b, "max", e.max().item(), "cols>1e-5:", (e > 1e-5).sum().item(), "argmax col", e.argmax().item(), "want/out at argmax", want[b, e.argmax()].item(), out[b, e.argmax()].item())
PY
