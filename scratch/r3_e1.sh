#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in 1 0; do
  NORI_HIP_LEAF2=$v timeout -k 10 200 python bench.py --workload c4 --no-cpu --no-secondary --steps 4 > gpurun_out/r3_e1_$v.json 2> gpurun_out/r3_e1_$v.err || { tail -5 gpurun_out/r3_e1_$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_e1_$v.json")); r=d["roofline"]; print("LEAF2=$v", d["ms_per_step"], d["value"], r["kernel_ms"], r["boxes_tested_per_ray"], r["prims_tested_per_ray"], d["rays_per_sample"])
PY
done
timeout -k 10 300 python -u -m pytest tests/test_gpu_c4.py -x -q --timeout 600 --timeout-method thread -k "not full_size_hits" 2>&1 | tail -3
