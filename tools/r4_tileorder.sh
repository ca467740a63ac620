#!/bin/bash
# GPU box: A/B of the fused edge kernel's XCD-aware tile order (graphcast_fused.hip; SKGC_XCD_TILE_ORDER=1; launch order is the default), then its tests.
cd "$(dirname "$0")/.."
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],3), {k: v["ms_per_step"] for k, v in r["stages"].items()})'
for v in xcd plain xcd plain; do
  env=""; [ $v = xcd ] && env="SKGC_XCD_TILE_ORDER=1"
  env $env python bench.py --model graphcast --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "$show" "graphcast/$v"
done
timeout 900 python -m pytest tests/test_graphcast_gpu.py tests/test_graphcast_fused_gpu.py -m gpu -q -x -k "not full_size and not ten_day" 2>&1 | tail -3
