#!/bin/bash
cd "$(dirname "$0")/.."
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],3), {k: v["ms_per_step"] for k, v in r["stages"].items()})'
timeout 600 python -m pytest tests/test_graphcast_gpu.py tests/test_graphcast_fused_gpu.py tests/test_rccl_gpu.py -q -x -m gpu -k "not full_size and not ten_day" 2>&1 | tail -3
for cfg in "spatial:" "spatial+xcd:SKGC_XCD_TILE_ORDER=1" "level:SKGC_MESH_ORDER=level" "level+xcd:SKGC_MESH_ORDER=level SKGC_XCD_TILE_ORDER=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "$show" "graphcast/$name"
done
