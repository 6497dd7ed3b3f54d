#!/bin/bash
# usage: kt_quick.sh <tag> <workload...>: kernel-trace stats of short bench runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for w in "$@"; do
  out=gpurun_out/kt_${tag}_$w; mkdir -p $out
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-secondary > $out/bench.json 2> $out/err.txt || { echo "failed $w"; tail -3 $out/err.txt; exit 1; }
  cp $out/kt/*/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/kt
done
