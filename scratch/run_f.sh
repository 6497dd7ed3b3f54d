#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
for args in "--workload c5" "--workload table --no-cpu" "--workload table --partition tiles --no-cpu" "--workload c4 --no-cpu --partition tiles" "--workload cbox --no-cpu" "--workload cbox --partition tiles --no-cpu"; do
tag=$(echo $args | tr ' -' '__')
timeout -k 10 400 python -u bench.py $args --steps 3 --warmup 1 > gpurun_out/r2f/bench_$tag.json 2> gpurun_out/r2f/bench_$tag.err || { tail -5 gpurun_out/r2f/bench_$tag.err; exit 1; }
python3 -c "
import json,sys
j=json.load(open('gpurun_out/r2f/bench_$tag.json'))
print('$args', j['ms_per_step'], j['value'], j.get('phases_ms_per_step'), j.get('cpu_baseline',{}).get('value'))
"
done
