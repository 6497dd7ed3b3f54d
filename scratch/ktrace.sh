#!/bin/bash
# usage: scratch/ktrace.sh <tag> <env assignments...> : rocprofv3 kernel trace of a short bench; prints per-kernel totals per render
tag=$1; shift
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kt_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/kt_$tag.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/kt_$tag.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep ms_per_step gpurun_out/kt_$tag.log | cut -c1-200
python3 - $tag <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/kt_%s/*/*kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    n = r['Name']; short = n.split('(anonymous namespace)::')[-1].split('(')[0][-40:] if '::' in n else n[:40]
    if 'k_trace_pool' in n: short = 'k_trace_pool<' + n.split('k_trace_pool<')[1].split('>')[0] + '>'
    print('%-45s calls %5s  total/render %8.3f ms  avg %9.1f us  max %9.1f us' % (short, r['Calls'], float(r['TotalDurationNs']) / 4e6, float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
