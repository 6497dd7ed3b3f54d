#!/bin/bash
# two pipelines on scenes that live in the caches, with room left for the other pipeline's shade stage beside the trace stage
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_tc_$tag.json 2> gpurun_out/s2_tc_$tag.err || { tail -5 gpurun_out/s2_tc_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_tc_$tag.json"))
print("$tag ms/step", d["ms_per_step"], flush=True)
PY
}
for wl in table c5; do
  run ${wl}_p1 $wl NORI_HIP_PIPELINES=1 || exit 1
  run ${wl}_p2 $wl NORI_HIP_PIPELINES=2 || exit 1
  for b in 3 4; do
    run ${wl}_p2_bpc$b $wl NORI_HIP_PIPELINES=2 NORI_HIP_TRACE_BPC=$b || exit 1
    run ${wl}_p2_bpc${b}_sw5 $wl NORI_HIP_PIPELINES=2 NORI_HIP_TRACE_BPC=$b NORI_HIP_SHADE_WAVES=5 || exit 1
  done
  run ${wl}_p1_bpc4 $wl NORI_HIP_PIPELINES=1 NORI_HIP_TRACE_BPC=4 || exit 1
done
