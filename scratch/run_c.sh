#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -u -m pytest tests/test_gpu_c4.py "tests/test_gpu_parity.py" -k "c4 or device_built" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2c/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r2c/pytest.log
[ $rc -eq 0 ] || exit $rc
for w in c4 table cbox dragon; do
for m in plan plain; do
NORI_HIP_BVH=device NORI_HIP_LBVH_MORTON=$m timeout -k 10 300 python -u bench.py --workload $w --bvh device --steps 2 --warmup 1 --no-cpu > gpurun_out/r2c/bench_${w}_$m.json 2> gpurun_out/r2c/bench_${w}_$m.err || exit 1
python3 -c "
import json,sys
j=json.load(open('gpurun_out/r2c/bench_${w}_$m.json'))
print('$w $m', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['boxes_tested_per_ray'], j['roofline']['prims_tested_per_ray'], j['config']['bvh'])
"
done
done
