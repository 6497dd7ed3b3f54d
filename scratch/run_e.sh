#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2e/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r2e/pytest.log
[ $rc -eq 0 ] || exit 1
for w in table cbox dragon subsurface; do
for b in host device; do
timeout -k 10 300 python -u bench.py --workload $w --bvh $b --steps 3 --warmup 1 --no-cpu > gpurun_out/r2e/bench_${w}_${b}.json 2> gpurun_out/r2e/bench_${w}_${b}.err || exit 1
python3 -c "
import json,sys
j=json.load(open('gpurun_out/r2e/bench_${w}_${b}.json'))
r=j['roofline']
print('$w $b', j['ms_per_step'], j['value'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'boxes/ray', r['boxes_tested_per_ray'], 'prims/ray', r['prims_tested_per_ray'])
"
done
done
