#!/bin/bash
# GPU box: A/B of the strided GEMM's XCD-aware tile order (sfno_ops.hip; SKSFNO_NO_XCD_REMAP=1 is the plain order) on the two models that
# run it, then the small-grid parity tests of both with the new order.
cd "$(dirname "$0")/.."
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],3), {k: (v["ms_per_step"] if isinstance(v, dict) else v) for k, v in (r.get("stages") or d.get("stages_ms_per_step") or r.get("stages_ms_per_step") or {}).items()})'
for m in sfno graphcast; do
  for v in remap plain; do
    env=""; [ $v = plain ] && env="SKSFNO_NO_XCD_REMAP=1"
    env $env python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "$show" "$m/$v"
  done
done
timeout 900 python -m pytest tests/test_sfno_gpu.py tests/test_graphcast_gpu.py tests/test_graphcast_fused_gpu.py -m gpu -q -x -k "not full_size and not ten_day" 2>&1 | tail -3
