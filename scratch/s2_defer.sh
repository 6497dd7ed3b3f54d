#!/bin/bash
# shade stage with the film updates of ended paths in pass B (NORI_HIP_SHADE_DEFER=1) against the default, per workload
set -o pipefail
mkdir -p gpurun_out
for wl in cbox table dragon c5 c4; do
  for g in 0 1; do
    NORI_HIP_SHADE_DEFER=$g timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_defer_${wl}_$g.json 2> gpurun_out/s2_defer_${wl}_$g.err || { tail -5 gpurun_out/s2_defer_${wl}_$g.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/s2_defer_${wl}_$g.json"))
print("${wl} defer=$g ms/step", d["ms_per_step"], "Msamples/s", d["value"], flush=True)
PY
  done
done
NORI_HIP_SHADE_DEFER=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
