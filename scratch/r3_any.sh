#!/bin/bash
set -o pipefail
for w in table cbox dragon; do
for a in default reference; do
  if [ $a = default ]; then unset NORI_HIP_ANYORDER; else export NORI_HIP_ANYORDER=$a; fi
  timeout -k 10 200 python bench.py --workload $w --no-cpu --steps 5 > gpurun_out/r3_any_${w}_$a.json 2> gpurun_out/r3_any.err || { tail -5 gpurun_out/r3_any.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_any_${w}_$a.json")); r=d["roofline"]; print("$w anyorder=$a", d["ms_per_step"], d["value"], r["kernel_ms"], r["boxes_tested_per_ray"])
PY
done; done
