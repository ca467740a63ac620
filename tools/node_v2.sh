#!/bin/bash
# GPU box: the node kernel's second / third form (-DSKGC_NODE_V2: first Linear K-outer on 128-row tiles; -DSKGC_NODE_V3: hidden units in two halves,
# every weight stage against both row groups -- csrc/graphcast_fused.hip) against the in-tree build.
#   bash tools/node_v2.sh [v2] [v3]          (default: both)
# Per variant: builds the library if it is not there, the node unit tests and the small-graph engine tests with it, the kernel timings of
# tools/gc_edge_probe.py at 40962 and 1 M rows and the full-size step; at the end the same timings with the default library.
cd "$(dirname "$0")/.."
C=skyrim_amd/csrc; O=skyrim_amd/lib/obj; mkdir -p skyrim_amd/lib/variants
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],3), {k: v["ms_per_step"] for k, v in r["stages"].items()})'
timings() {
  for rows in 40962 1038240; do python tools/gc_edge_probe.py 8 8 $rows 2>&1 | grep "node mlp" | sed "s/^/$1 /"; done
  python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "$show" "graphcast/$1"
}
for v in ${*:-v2 v3}; do
  V=$PWD/skyrim_amd/lib/variants/libskyrim_graphcast_node$v.so
  D=$(echo $v | tr a-z A-Z)
  if [ ! -f $V ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSKGC_NODE_$D -c $C/graphcast_fused.hip -o /tmp/gcf_$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $V $O/graphcast_ops.o /tmp/gcf_$v.o $O/aux.o || exit 1
  fi
  (
    export SKYRIM_GRAPHCAST_LIB=$V SKGC_NODE_$D=1
    timeout 600 python -m pytest tests/test_graphcast_fused_gpu.py tests/test_graphcast_gpu.py -m gpu -q -x -k "not full_size and not ten_day" 2>&1 | tail -3
    timings $v
  )
done
timings default
