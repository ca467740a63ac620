#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_pangu_gpu.py -m gpu -q -s -k "full_step_per_channel or outlier or calibration_is_deterministic or other_geometries or switchable or golden" 2>&1 | grep -i "outlier\|passed\|failed\|Error\|assert " | tee $O/pangu_quick.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-models --no-alt-modes > $O/bench_pangu.json 2> $O/bench_pangu.err
python -c "
import json
d=json.loads(open('$O/bench_pangu.json').read().strip().splitlines()[-1])
print('pangu ms/step', d['ms_per_step'], 'parity', d.get('parity'))
for k,v in d['roofline']['stages'].items():
    if 'attn' in k or 'qkv' in k or 'proj' in k: print('   ', k, v)
" || tail -c 600 $O/bench_pangu.err
