#!/bin/bash
# GPU box: correctness + stage timing of every SKP_MLP_VARIANT of csrc/fused_mlp.hip (experiments)
for v in ${VARIANTS:-0 1 2 3}; do
  SKP_MLP_VARIANT=$v timeout 300 python -m pytest tests/test_pangu_gpu.py -m gpu -x -q -k "fused_mlp" 2>&1 | tail -1
  SKP_MLP_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-alt-modes > gpurun_out/bv_$v.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bv_$v.log").read().strip().splitlines()[-1])
    st=d["roofline"]["stages"]
    print("variant $v: step %.2f ms  mlp_r0 %.4f  mlp_r1 %.4f" % (d["ms_per_step"], st["mlp_r0"]["ms_per_launch"], st["mlp_r1"]["ms_per_launch"]))
except Exception as e:
    print("variant $v failed", e)
PY
done
