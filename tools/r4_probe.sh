#!/bin/bash
# GPU box: unit tests of the fused GraphCast kernels + the per-phase clocks of tools/gc_edge_probe.py (round-4 development loop)
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q 2>&1 | tail -3
timeout 300 python tools/gc_edge_probe.py 2640 4096 2>&1 | grep -v "amdgpu.ids\|^nodes"
