#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2g/pytest.log 2>&1; rc=$?
tail -8 gpurun_out/r2g/pytest.log
[ $rc -eq 0 ] || exit 1
for args in "--workload c4" "--workload table --bvh host" "--workload table --bvh device" "--workload cbox --bvh host" "--workload cbox --bvh device" "--workload dragon --bvh host" "--workload subsurface"; do
tag=$(echo $args | tr ' -' '__')
timeout -k 10 300 python -u bench.py $args --steps 3 --warmup 1 --no-cpu > gpurun_out/r2g/bench_$tag.json 2> gpurun_out/r2g/bench_$tag.err || exit 1
python3 -c "
import json,sys
j=json.load(open('gpurun_out/r2g/bench_$tag.json'))
r=j['roofline']
print('$args', j['ms_per_step'], j['value'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'boxes/ray', r['boxes_tested_per_ray'], 'prims/ray', r['prims_tested_per_ray'], j['config']['bvh'])
"
done
