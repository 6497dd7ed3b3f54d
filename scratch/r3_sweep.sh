#!/bin/bash
# tunables of the trace stage on C4 with the current kernel; one line per setting
set -o pipefail
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload c4 --no-cpu --no-secondary --steps 3 > gpurun_out/r3_sw_$tag.json 2> gpurun_out/r3_sw_$tag.err || { tail -5 gpurun_out/r3_sw_$tag.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_sw_$tag.json")); r=d["roofline"]; print("$tag", d["ms_per_step"], d["value"], r["kernel_ms"], r["launches_per_step"])
PY
}
run base A=1
run base2 A=1
if [ -f scratch/var4/libnori_hip.so ]; then
  cp cuda-path-tracer_amd/libnori_hip.so /tmp/libnori_hip_w5.so
  cp scratch/var4/libnori_hip.so cuda-path-tracer_amd/libnori_hip.so && run w4deep A=1
  cp /tmp/libnori_hip_w5.so cuda-path-tracer_amd/libnori_hip.so
fi
