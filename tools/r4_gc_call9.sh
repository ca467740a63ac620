#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/node_dbg2.log
import torch, numpy as np
from skyrim_amd import ops
from skyrim_amd.graphcast import fused as fz
L=512
for n_src, rows in ((2, 333), (2, 128), (1, 333)):
    gen = torch.Generator().manual_seed(20 + n_src)
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    dev = torch.device("cuda:0")
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
    print(n_src, rows, "max err", err.max().item(), "rel", (err.max() / want.abs().max()).item(), "rows>1e-4:", (per_row > 1e-4).nonzero().flatten().tolist()[:20], "median row err", per_row.median().item())
PY
timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_gc.json 2> $O/bench_gc.err
python -c "
import json,sys
d=json.loads(open('$O/bench_gc.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'finite', d['config']['finite'])
for k,v in d['roofline']['stages'].items(): print('   ', k, v)
" || tail -c 600 $O/bench_gc.err
timeout 1500 python -m pytest tests/test_graphcast_gpu.py tests/test_graphcast_fused_gpu.py -m gpu -q -k "not ten_day and not node_mlp" 2>&1 | tail -6
