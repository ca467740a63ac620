#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4
mkdir -p $O
SKYRIM_TEST_ALL_PLANS=1 timeout 1500 python -m pytest tests/test_pangu_gpu.py -m gpu -q -s -k "full_size_term_plans or outlier or calibration_is_deterministic or calibrates_on_the_first" 2>&1 | grep -i "full-size\|outlier\|passed\|failed\|Error\|assert" | tee $O/pangu_plans.log
timeout 900 python -m pytest tests/test_graphcast_fused_gpu.py tests/test_graphcast_gpu.py -m gpu -q -k "fused or full_size or sharded or latent_512" 2>&1 | tail -5 | tee $O/gc_tests.log
timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_gc.json 2> $O/bench_gc.err
python -c "
import json
d=json.loads(open('$O/bench_gc.json').read().strip().splitlines()[-1])
print('gc ms/step', d['ms_per_step']); r=d['roofline']; print({k:r[k] for k in ('bound','achieved','peak','unit','frac','kernel','hbm','mfma')})
for k,v in r['stages'].items(): print('   ', k, v)
" || tail -c 800 $O/bench_gc.err
