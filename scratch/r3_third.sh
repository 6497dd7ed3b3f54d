#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for p in mega wavefront; do
  NORI_HIP_PIPELINE=$p timeout -k 10 200 python bench.py --workload proj --spp 64 --no-cpu --steps 3 > gpurun_out/r3_proj_$p.json 2> gpurun_out/r3_proj_$p.err || { tail -5 gpurun_out/r3_proj_$p.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_proj_$p.json")); r=d["roofline"]; print("proj 64spp $p", d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms"], d["mrays_per_s"])
PY
done
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread -k "not full_size_hits" 2>&1 | tail -3
