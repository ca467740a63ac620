#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
for v in p1 p1rd5 p1rd2 p5; do
  SKYRIM_GRAPHCAST_LIB=skyrim_amd/lib/variants/libgc_$v.so timeout 300 python tools/gc_edge_probe.py 2640 4096 2>&1 | grep -v amdgpu.ids | head -4
done | tee $O/probe4.log
