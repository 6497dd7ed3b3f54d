#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2d/pytest.log 2>&1; rc=$?
tail -25 gpurun_out/r2d/pytest.log
[ $rc -eq 0 ] || echo "PYTEST FAILED rc=$rc"
for w in c4 table cbox dragon; do
for b in device host; do
if [ $w = c4 ] && [ $b = host ]; then continue; fi
for wide in 1 0; do
NORI_HIP_WIDE=$wide timeout -k 10 300 python -u bench.py --workload $w --bvh $b --steps 2 --warmup 1 --no-cpu > gpurun_out/r2d/bench_${w}_${b}_$wide.json 2> gpurun_out/r2d/bench_${w}_${b}_$wide.err || exit 1
python3 -c "
import json,sys
j=json.load(open('gpurun_out/r2d/bench_${w}_${b}_$wide.json'))
r=j['roofline']
print('$w $b wide=$wide', j['ms_per_step'], j['value'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'boxes/ray', r['boxes_tested_per_ray'], 'prims/ray', r['prims_tested_per_ray'])
"
done
done
done
