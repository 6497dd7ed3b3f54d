#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_twin_$tag.json 2> gpurun_out/s2_twin_$tag.err || { tail -5 gpurun_out/s2_twin_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_twin_$tag.json"))
r=d["roofline"]
print("$tag ms/step", d["ms_per_step"], "trace ms", r["kernel_ms"], "launches", r["launches_per_step"], flush=True)
PY
}
run c4_p1 c4 NORI_HIP_PIPELINES=1 || exit 1
for sp in 50 56 62 69 75 81; do run c4_split$sp c4 NORI_HIP_TWIN_SPLIT=$sp NORI_HIP_FIN_SHARE=1 || exit 1; done
for sp in 50 62 75; do run c4_split${sp}_share2 c4 NORI_HIP_TWIN_SPLIT=$sp NORI_HIP_FIN_SHARE=2 || exit 1; done
