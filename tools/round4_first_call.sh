#!/bin/bash
# GPU box, first call of the next round: what round 3 ran out of GPU minutes for.
#   1. full-size 24-h rollout errors of the compensated plans and of the one-term plan (one oracle rollout serves all; ~6 min)
#   2. the Pangu GPU tests with the compensated rounding as the process default (what promoting it to DEFAULT_ROUNDING would run; ~4 min)
#   3. sustained step times of f16x2m / f16x1m (20 steps each)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SKYRIM_TEST_ALL_PLANS=1 python -m pytest tests/test_pangu_gpu.py -m gpu -q -s -k full_size_term_plans 2>&1 | grep -i "full-size\|passed\|failed\|Error" > gpurun_out/r4_plans.log
SKYRIM_PANGU_ROUNDING=compensated python -m pytest tests/test_pangu_gpu.py tests/test_ingest_gpu.py -m gpu -q -k "not full_size" 2>&1 | tail -5 > gpurun_out/r4_compensated_suite.log
for m in f16x2m f16x1m; do
  python bench.py --precision $m --steps 20 --warmup 3 --no-cpu-baseline --no-models --no-alt-modes --no-parity > gpurun_out/r4_bench_$m.json 2> gpurun_out/r4_bench_$m.err
done
cat gpurun_out/r4_plans.log gpurun_out/r4_compensated_suite.log; head -c 300 gpurun_out/r4_bench_f16x1m.json
