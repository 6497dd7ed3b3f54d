#!/bin/bash
# two pipelines per render call: one pipeline / two with their own tail kernels / two with one tail launch over both lists
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_twin_$tag.json 2> gpurun_out/s2_twin_$tag.err || { tail -5 gpurun_out/s2_twin_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_twin_$tag.json"))
r=d["roofline"]
print("$tag ms/step", d["ms_per_step"], "Msamples/s", d["value"], "trace ms", r["kernel_ms"], "launches", r["launches_per_step"], "rays/sample", d["rays_per_sample"], flush=True)
PY
}
run c4_p1 c4 NORI_HIP_PIPELINES=1 || exit 1
run c4_p2_separate c4 NORI_HIP_TWIN_TAIL=separate || exit 1
run c4_p2_merged c4 || exit 1
run c4_p1_again c4 NORI_HIP_PIPELINES=1 || exit 1
run c4_p2_merged_again c4 || exit 1
for wl in table c5 cbox; do
  run ${wl}_p1 $wl NORI_HIP_PIPELINES=1 || exit 1
  run ${wl}_p2 $wl NORI_HIP_PIPELINES=2 || exit 1
done
timeout -k 10 800 python -u -m pytest tests/test_gpu_c4.py tests/test_gpu_group.py -m gpu -x -q --timeout 400 --timeout-method thread 2>&1 | tail -5
