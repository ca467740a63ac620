#!/bin/bash
# GPU box: the measurements kept under profiles/ for a round -- default bench lines of the three models, rocprofv3 --kernel-trace --stats
# of the same commands, and the counter passes (tools/pmc_collect.sh).  Everything lands in gpurun_out/final/ (gpurun merges that
# directory back); copy <round>_* from there into profiles/ afterwards.
#   bash tools/final_profiles.sh [models...]        (default: pangu sfno graphcast)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${PROFILE_ROUND:-r06}
O=gpurun_out/final
mkdir -p $O profiles
MODELS=${*:-pangu sfno graphcast}
for m in $MODELS; do
  extra=""; [ $m != pangu ] && extra="--model $m"
  lib=skyrim_amd/lib/libskyrim_$m.so
  stamp="$(sha256sum $lib | cut -c1-16) $(basename $lib), $(date -u +%Y-%m-%dT%H:%MZ), src $(python -c 'import bench; print(bench.src_sha16())')"
  rm -rf $O/stats_$m
  # (calibration off: its one-time launches stay out of the trace; guard off: it would judge -- and replace -- the uncalibrated plan)
  SKYRIM_BENCH_FULL_LINE=1 SKYRIM_PANGU_CALIBRATION=off SKYRIM_PANGU_GUARD=off timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -o p -- python bench.py $extra --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-alt-modes --no-models > $O/stats_$m.log 2>&1
  echo "stats $m rc=$?"
  cp $(ls $O/stats_$m/*/p_kernel_stats.csv $O/stats_$m/p_kernel_stats.csv 2>/dev/null | head -1) $O/${R}_${m}_kernel_stats.csv 2>/dev/null
  bash tools/pmc_collect.sh $m $extra
  python tools/pmc_summary.py gpurun_out/pmc_$m $O/${R}_${m}_pmc.json --steps 3 --bench-json $O/stats_$m.log --stamp "$stamp" > $O/pmc_$m.log 2>&1 || tail -3 $O/pmc_$m.log
  cp $O/${R}_${m}_pmc.json profiles/ 2>/dev/null          # the bench lines below read the counter summaries (roofline.traffic)
  rm -rf gpurun_out/pmc_${m}_*                              # raw counter CSVs: tens of MB, summarised above
done
for m in $MODELS; do
  case $m in
    pangu) python bench.py > $O/${R}_bench_pangu_line.json 2> $O/bench_pangu.err      # stdout: the compact line the driver records
           cp bench_detail.json $O/${R}_bench_pangu.json                             # the full record of the same run
           python bench.py --graph --no-cpu-baseline --no-parity --no-alt-modes --no-models > $O/${R}_bench_pangu_graph.json 2> $O/bench_pangu_graph.err ;;
    sfno) python bench.py --model sfno > $O/${R}_bench_sfno.json 2> $O/bench_sfno.err ;;
    graphcast) python bench.py --model graphcast --steps 5 > $O/${R}_bench_graphcast.json 2> $O/bench_graphcast.err ;;
  esac
  echo "bench $m rc=$?"
done
ls $O
