#!/bin/bash
# C4, two pipelines: room for the other pipeline's shade stage beside the trace stage?
set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_cb_$tag.json 2> gpurun_out/s2_cb_$tag.err || { tail -5 gpurun_out/s2_cb_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_cb_$tag.json"))
print("$tag ms/step", d["ms_per_step"], "trace ms", d["roofline"]["kernel_ms"], flush=True)
PY
}
run default || exit 1
for b in 3 4; do run bpc$b NORI_HIP_TRACE_BPC=$b || exit 1; run bpc${b}_sw5 NORI_HIP_TRACE_BPC=$b NORI_HIP_SHADE_WAVES=5 || exit 1; done
run sw5 NORI_HIP_SHADE_WAVES=5 || exit 1
run sw3 NORI_HIP_SHADE_WAVES=3 || exit 1
run default_again || exit 1
