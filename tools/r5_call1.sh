#!/bin/bash
# Round 5, first GPU call: (1) the wide row-tile Pangu block kernel -- toy-grid parity against the oracle with several tiles per workgroup, then
# the full-size A/B; (2) GraphCast's node kernel forms (tools/node_v2.sh without its long test runs).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
K='(f16x2m and (earth_specific_block or full_step_per_channel or rollout_4)) or one_term_block_gemms or per_layer_term_plan'
for pipe in 0 1; do
  echo "== toy parity wide=3 grid=3 pipe=$pipe"
  SKP_BLK_WIDE=3 SKP_WIDE_GRID=3 SKP_WIDE_PIPE=$pipe timeout 420 python -m pytest tests/test_pangu_gpu.py -q -x -m gpu -k "$K" 2>&1 | tail -4
done
echo "== full-size A/B"
timeout 600 python tools/r5_wide_ab.py f16x2m f16x1m 2>&1 | grep -v amdgpu.ids
echo "== graphcast node forms"
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],3), {k: v["ms_per_step"] for k, v in r["stages"].items()})'
timings() {
  for rows in 40962 1038240; do timeout 200 python tools/gc_edge_probe.py 8 8 $rows 2>&1 | grep "node mlp" | sed "s/^/$1 /"; done
  timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "$show" "graphcast/$1"
}
for v in v2 v3; do
  V=$PWD/skyrim_amd/lib/variants/libskyrim_graphcast_node$v.so
  D=$(echo $v | tr a-z A-Z)
  (
    export SKYRIM_GRAPHCAST_LIB=$V SKGC_NODE_$D=1
    timeout 300 python -m pytest tests/test_graphcast_fused_gpu.py -m gpu -q -x -k "node or hi_lo or fused_engine" 2>&1 | tail -3
    timings $v
  )
done
timings default
