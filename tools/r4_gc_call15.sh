#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
timeout 900 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_graphcast_gpu.py -m gpu -q -k "not ten_day and not two_process" 2>&1 | tail -4
timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_gc.json 2> $O/bench_gc.err
python -c "
import json,sys
d=json.loads(open('$O/bench_gc.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'finite', d['config']['finite'])
for k,v in d['roofline']['stages'].items(): print('   ', k, v)
" || tail -c 600 $O/bench_gc.err
timeout 1200 python -m pytest tests/test_pangu_gpu.py -m gpu -q -s -k "full_size_step_vs_oracle or precision_modes" 2>&1 | grep -i "err\|passed\|failed\|mode" | tail -20
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-models --no-alt-modes > $O/bench_pangu.json 2> $O/bench_pangu.err
python -c "
import json
d=json.loads(open('$O/bench_pangu.json').read().strip().splitlines()[-1])
print('pangu ms/step', d['ms_per_step'], 'parity', d.get('parity'))
"
