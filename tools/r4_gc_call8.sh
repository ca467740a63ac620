#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
timeout 600 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q 2>&1 | tail -4
for v in default; do
  lib=skyrim_amd/lib/variants/libgc_$v.so; [ $v = default ] && lib=skyrim_amd/lib/libskyrim_graphcast.so
  SKYRIM_GRAPHCAST_LIB=$lib timeout 300 python tools/gc_edge_probe.py 2640 4096 2>&1 | grep -v amdgpu.ids
done | tee $O/probe8.log
