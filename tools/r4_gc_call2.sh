#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
timeout 600 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q 2>&1 | tail -15
for v in default unrolled rd5 rd2; do
  lib=skyrim_amd/lib/variants/libgc_$v.so; [ $v = default ] && lib=skyrim_amd/lib/libskyrim_graphcast.so
  SKYRIM_GRAPHCAST_LIB=$lib timeout 300 python tools/gc_edge_probe.py 2>&1 | tail -6
done | tee $O/probe2.log
