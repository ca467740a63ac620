#!/bin/bash
# Build-time variants of the fused GraphCast kernels for tools/gc_edge_probe.py (here, before gpurun: hipcc cross-compiles).
#   bash tools/build_gc_variants.sh "name:flags" ...      e.g.  "nodma:-DSKP_PROBES=1" "nomfma:-DSKP_PROBES=2"
set -eu
cd "$(dirname "$0")/../skyrim_amd/csrc"
mkdir -p ../lib/variants ../lib/obj
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c graphcast_fused.hip -o ../lib/obj/gcf_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libgc_$name.so ../lib/obj/graphcast_ops.o ../lib/obj/gcf_$name.o ../lib/obj/aux.o
  echo "built libgc_$name.so ($flags)"
done
