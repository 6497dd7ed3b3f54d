#!/bin/bash
# usage: scratch/kt.sh <tag> <bench args...>: rocprofv3 kernel trace of one bench run, per-kernel totals
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_$tag; mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > $out/bench.json 2> $out/err.txt || { tail -3 $out/err.txt; exit 1; }
cp $out/kt/*/*kernel_stats.csv $out/kernel_stats.csv
python3 - $out <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1] + '/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:9]:
    print('%-70s calls %5s total %9.2f ms avg %9.1f us %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
j = json.load(open(sys.argv[1] + '/bench.json')); print(j['ms_per_step'], j['value'])
PY
