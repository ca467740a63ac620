#!/bin/bash
# GPU box: FETCH_SIZE (L2 -> fabric reads) of the fused edge kernels per launch, XCD-aware tile order against launch order.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp SKYRIM_PANGU_CALIBRATION=off
for v in xcd plain; do
  env=""; [ $v = xcd ] && env="SKGC_XCD_TILE_ORDER=1"
  out=gpurun_out/fetch_$v; rm -rf $out
  env $env timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o p -- python bench.py --model graphcast --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $out.log 2>&1
  python - $out $v <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
tot, cnt = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].split("(")[0][:60]
    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in tot:
    if "edge_update" in k or "node_mlp" in k or "gemm_strided" in k:
        print(sys.argv[2], k, cnt[k], "launches", round(tot[k] / cnt[k] * 64 / 1e9, 4), "GB/launch (FETCH_SIZE x 64 B)")
PY
  rm -rf $out
done
