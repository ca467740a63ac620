#!/bin/bash
# GPU box: the measurements kept under profiles/ for a round -- default bench lines of the three models, rocprofv3 --kernel-trace --stats
# of the same commands, and the counter passes (tools/pmc_collect.sh).  Everything lands in gpurun_out/final/.
#   bash tools/final_profiles.sh [models...]        (default: pangu sfno graphcast)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
MODELS=${*:-pangu sfno graphcast}
for m in $MODELS; do
  extra=""; [ $m != pangu ] && extra="--model $m"
  rm -rf $O/stats_$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -o p -- python bench.py $extra --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-alt-modes > $O/stats_$m.log 2>&1
  echo "stats $m rc=$?"
  bash tools/pmc_collect.sh $m $extra
  python tools/pmc_summary.py gpurun_out/pmc_$m profiles/r02_${m}_pmc.json --steps 3 > $O/pmc_$m.log 2>&1 || tail -3 $O/pmc_$m.log
  cp profiles/r02_${m}_pmc.json $O/ 2>/dev/null
done
# the bench lines last: they read the counter summaries written above (roofline.traffic)
for m in $MODELS; do
  case $m in
    pangu) python bench.py > $O/bench_pangu.json 2> $O/bench_pangu.err
           python bench.py --graph --no-cpu-baseline --no-parity --no-alt-modes > $O/bench_pangu_graph.json 2> $O/bench_pangu_graph.err ;;
    sfno) python bench.py --model sfno > $O/bench_sfno.json 2> $O/bench_sfno.err ;;
    graphcast) python bench.py --model graphcast --steps 5 > $O/bench_graphcast.json 2> $O/bench_graphcast.err ;;
  esac
  echo "bench $m rc=$?"
done
ls $O
