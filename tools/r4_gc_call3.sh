#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4gc
mkdir -p $O
for v in default nodma nomfma neither; do
  lib=skyrim_amd/lib/variants/libgc_$v.so; [ $v = default ] && lib=skyrim_amd/lib/libskyrim_graphcast.so
  SKYRIM_GRAPHCAST_LIB=$lib timeout 300 python tools/gc_edge_probe.py 2>&1 | grep -v amdgpu.ids | head -4
done | tee $O/probe3.log
echo "--- 64 nodes (gathers hit L2)"
GCP_NODES=64 timeout 300 python tools/gc_edge_probe.py 2>&1 | grep -v amdgpu.ids | head -4 | tee -a $O/probe3.log
