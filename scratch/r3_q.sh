#!/bin/bash
set -o pipefail
for v in default; do
  timeout -k 10 200 python bench.py --workload c4 --no-cpu --no-secondary --steps 4 > gpurun_out/r3_q_$v.json 2> gpurun_out/r3_q_$v.err || { tail -5 gpurun_out/r3_q_$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_q_$v.json")); r=d["roofline"]; print("QUANT=$v", d["ms_per_step"], d["value"], r["kernel_ms"], r["boxes_tested_per_ray"], r["prims_tested_per_ray"], r["node_visits_per_ray"], d["rays_per_sample"])
PY
done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread 2>&1 | tail -5
