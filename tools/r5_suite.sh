#!/bin/bash
# the driver's round-end sequence on one box: GPU suite (with durations), smoke, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=45 ) > gpurun_out/suite_r5.log 2>&1
tail -60 gpurun_out/suite_r5.log | grep -v 'Librccl\|RCCL version\|HIP version\|ROCm version\|Hostname'
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_r5.log 2> gpurun_out/bench_r5.err
tail -1 gpurun_out/bench_r5.log | cut -c1-3200; tail -5 gpurun_out/bench_r5.err
