#!/bin/bash
# occupancy variants of the trace stage: scratch/var<N>/libnori_hip.so built with -DND_WIDE_WAVES=N
set -o pipefail
run() {
  timeout -k 10 200 python bench.py --workload c4 --no-cpu --no-secondary --steps 4 > gpurun_out/r3_var_$1.json 2> gpurun_out/r3_var_$1.err || { tail -5 gpurun_out/r3_var_$1.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_var_$1.json")); r=d["roofline"]; print("$1", d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms"], r["frac"], r["boxes_tested_per_ray"], r["prims_tested_per_ray"], r["node_visits_per_ray"])
PY
}
run w5
cp cuda-path-tracer_amd/libnori_hip.so /tmp/libnori_hip_w5.so
for v in 9 7; do cp scratch/var$v/libnori_hip.so cuda-path-tracer_amd/libnori_hip.so && run w$v; done
cp /tmp/libnori_hip_w5.so cuda-path-tracer_amd/libnori_hip.so
