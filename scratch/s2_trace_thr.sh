#!/bin/bash
# trace-stage step thresholds re-swept on the scenes that live in the caches (the kernel is VALU-issue-bound there since the four-box nodes)
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_thr_$tag.json 2> gpurun_out/s2_thr_$tag.err || { tail -5 gpurun_out/s2_thr_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_thr_$tag.json"))
print("$tag ms/step", d["ms_per_step"], "trace ms", d["roofline"]["kernel_ms"], flush=True)
PY
}
for wl in table cbox; do
  for t in 1,1,12 8,1,12 16,1,12 1,8,12 1,16,12 1,24,12 1,32,12 8,8,12 16,16,12 1,1,6 1,1,20 1,16,20; do run ${wl}_thr_$t $wl NORI_HIP_THR=$t || exit 1; done
  for b in 0 24 48; do run ${wl}_box2_$b $wl NORI_HIP_BOX2=$b || exit 1; done
  run ${wl}_spec0 $wl NORI_HIP_SPEC=0 || exit 1
  run ${wl}_spec0_p16 $wl NORI_HIP_SPEC=0 NORI_HIP_THR=1,16,12 || exit 1
done
