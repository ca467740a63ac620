#!/bin/bash
# GPU box, round 4: the fused GraphCast kernels -- unit tests, the GraphCast GPU tests, a bench line and a kernel trace.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4gc
mkdir -p $O
timeout 900 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q 2>&1 | tail -40 > $O/unit.log
echo "unit rc=$?"; tail -15 $O/unit.log
timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_gc.json 2> $O/bench_gc.err
echo "bench rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/bench_gc.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'finite', d['config']['finite'])
for k,v in d['roofline']['stages'].items(): print('   ', k, v)
" || tail -c 600 $O/bench_gc.json
tail -5 $O/bench_gc.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python bench.py --model graphcast --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/stats.log 2>&1
cp $(ls $O/stats/*/p_kernel_stats.csv $O/stats/p_kernel_stats.csv 2>/dev/null | head -1) $O/gc_kernel_stats.csv 2>/dev/null
cut -c1-150 $O/gc_kernel_stats.csv | head -14
rm -rf $O/stats
timeout 1500 python -m pytest tests/test_graphcast_gpu.py -m gpu -q -k "not ten_day" 2>&1 | tail -30 > $O/gc_tests.log
echo "gc tests rc=$?"; tail -12 $O/gc_tests.log
