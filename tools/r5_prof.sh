#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_x1m
SKYRIM_PANGU_CALIBRATION=off SKYRIM_PANGU_ROUNDING=nearest timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x1m -o x1m -- python tools/mode_times.py f16x1m > gpurun_out/prof_x1m/run.log 2>&1
find gpurun_out/prof_x1m -name '*kernel_stats.csv' | head -3
f=$(find gpurun_out/prof_x1m -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
