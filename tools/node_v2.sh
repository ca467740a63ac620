#!/bin/bash
# GPU box: the node kernel's second form (-DSKGC_NODE_V2, csrc/graphcast_fused.hip: first Linear K-outer on 128-row tiles) against the in-tree
# build.  Builds the variant library if it is not there; then, with the variant: the node unit tests and the engine tests on small graphs, the
# kernel timings of tools/gc_edge_probe.py (40962 and 1 M rows) and the full-size step; then the same timings with the default library.
cd "$(dirname "$0")/.."
V=skyrim_amd/lib/variants/libskyrim_graphcast_nodev2.so
if [ ! -f $V ]; then
  C=skyrim_amd/csrc; O=skyrim_amd/lib/obj; mkdir -p skyrim_amd/lib/variants
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSKGC_NODE_V2 -c $C/graphcast_fused.hip -o /tmp/gcf_v2.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o $V $O/graphcast_ops.o /tmp/gcf_v2.o $O/aux.o || exit 1
fi
export SKYRIM_GRAPHCAST_LIB=$PWD/$V SKGC_NODE_V2=1
timeout 600 python -m pytest tests/test_graphcast_fused_gpu.py tests/test_graphcast_gpu.py -m gpu -q -x -k "not full_size and not ten_day" 2>&1 | tail -3
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],3), {k: v["ms_per_step"] for k, v in r["stages"].items()})'
for v in variant default; do
  [ $v = default ] && unset SKYRIM_GRAPHCAST_LIB SKGC_NODE_V2
  for rows in 40962 1038240; do python tools/gc_edge_probe.py 8 8 $rows 2>&1 | grep "node mlp" | sed "s/^/$v /"; done
  python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "$show" "graphcast/$v"
done
