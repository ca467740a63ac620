#!/bin/bash
set -u
cd "$(dirname "$0")/.."
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, numpy as np
from skyrim_amd import ops
from skyrim_amd.graphcast import fused as fz
L=512
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(21)
r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
rows = 4096
x = (3.0 * r(rows, L)).float()
eye = torch.eye(L)
zero, one = torch.zeros(L), torch.ones(L)
def run(w1, w2, b1):
    w1f, w2f = fz.prep_w1_fragments(w1.to(dev)), fz.prep_w2_fragments(w2.to(dev))
    out = torch.zeros(rows, L, device=dev)
    ops.hip.gc_node_mlp([x.to(dev)], [0], [L], w1f, w2f, b1.to(dev), zero.to(dev), one.to(dev), zero.to(dev), None, 0, L, out, 0, L, rows)
    torch.cuda.synchronize()
    return out.cpu().double()
def ref(w1, w2, b1):
    h = torch.nn.functional.silu(x.double() @ w1.double().T + b1.double())
    return torch.nn.functional.layer_norm(h @ w2.double().T, (L,), None, None, 1e-5), h
for name, w1, w2 in (("identity/identity", eye, eye), ("random/identity", (r(L, L) / L ** 0.5).float(), eye), ("identity/random", eye, (r(L, L) / L ** 0.5).float())):
    got = run(w1, w2, zero)
    want, h = ref(w1, w2, zero)
    err = (got - want).abs()
    pr = err.amax(1)
    bad = (pr > 3e-5).nonzero().flatten().tolist()
    print(name, "max err", err.max().item(), "bad rows", len(bad), bad[:8])
    for b in bad[:3]:
        e = err[b]
        print("   row", b, "cols>1e-5", (e > 1e-5).sum().item(), "argmax", e.argmax().item(), "h range", h[b].min().item(), h[b].max().item(), "x max", x[b].abs().max().item())
PY
