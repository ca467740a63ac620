#!/bin/bash
# the driver's round-end sequence on one box: GPU suite (with durations), smoke
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 -s ) > gpurun_out/suite_r5.log 2>&1
grep -n "full-size\|passed\|failed\|Error\|gain-1\|real" gpurun_out/suite_r5.log | cut -c1-250 | tail -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
