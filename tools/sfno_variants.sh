#!/bin/bash
# A/B runs of `bench.py --model sfno` under environment switches:  source tools/sfno_variants.sh; run base; run chan SKSFNO_F_SYN=channel
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout 200 python bench.py --model sfno --no-cpu-baseline --no-parity > gpurun_out/sfno_$name.json 2> gpurun_out/sfno_$name.err || tail -5 gpurun_out/sfno_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/sfno_{n}.json"))
    st = d["roofline"]["stages"]
    print(f"{n:>14s} {d['ms_per_step']:7.3f} ms | " + " ".join(f"{k}={v['ms_per_step']:.2f}" for k, v in st.items()))
except Exception as e:
    print(n, "failed", e)
PY
}
