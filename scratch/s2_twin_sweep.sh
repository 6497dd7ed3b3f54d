#!/bin/bash
# C4 with two pipelines: pool size per pipeline, hand-over to the tail kernel, dynamic share of the trace stage's rounds
set -o pipefail
mkdir -p gpurun_out
run() { # tag env...
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_tsw_$tag.json 2> gpurun_out/s2_tsw_$tag.err || { tail -5 gpurun_out/s2_tsw_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_tsw_$tag.json"))
r=d["roofline"]
print("$tag ms/step", d["ms_per_step"], "trace ms", r["kernel_ms"], "launches", r["launches_per_step"], flush=True)
PY
}
run default || exit 1
for p in 2621440 3145728 3670016 5242880 6291456; do run pool_$p NORI_HIP_POOL=$p || exit 1; done
for f in 786432 1048576 1572864 2097152; do run finish_$f NORI_HIP_FINISH=$f || exit 1; done
for d in 15 35 50; do run dyn_$d NORI_HIP_DYN=$d || exit 1; done
for x in 32 128; do run xcd_$x NORI_HIP_XCD=$x || exit 1; done
run default_again || exit 1
