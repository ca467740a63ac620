#!/bin/bash
# GPU box: the measurements kept under profiles/ for a round -- default bench lines of the three models, rocprofv3 --kernel-trace --stats
# of the same commands, and the counter passes (tools/pmc_collect.sh).  Everything lands in gpurun_out/final/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_pangu.json 2> $O/bench_pangu.err
python bench.py --graph --no-cpu-baseline --no-parity --no-alt-modes > $O/bench_pangu_graph.json 2> $O/bench_pangu_graph.err
python bench.py --model sfno > $O/bench_sfno.json 2> $O/bench_sfno.err
python bench.py --model graphcast --steps 5 > $O/bench_graphcast.json 2> $O/bench_graphcast.err
for m in pangu sfno graphcast; do
  extra=""; [ $m != pangu ] && extra="--model $m"
  rm -rf $O/stats_$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -o p -- python bench.py $extra --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-alt-modes > $O/stats_$m.log 2>&1
  echo "stats $m rc=$?"
done
bash tools/pmc_collect.sh pangu
bash tools/pmc_collect.sh sfno --model sfno
bash tools/pmc_collect.sh graphcast --model graphcast
ls $O
