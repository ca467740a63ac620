#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
timeout 600 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q 2>&1 | tail -8
timeout 300 python tools/gc_edge_probe.py 2640 4096 2>&1 | grep -v amdgpu.ids | tee $O/probe7.log
timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_gc.json 2> $O/bench_gc.err
python -c "
import json,sys
d=json.loads(open('$O/bench_gc.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'finite', d['config']['finite'])
for k,v in d['roofline']['stages'].items(): print('   ', k, v)
" || tail -c 600 $O/bench_gc.err
