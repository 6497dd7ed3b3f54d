#!/bin/bash
# tail kernel in two rounds (NORI_HIP_FINISH_ROUNDS=<min_alive>,<cap2>) against one round, per workload; then the parity tests with two rounds
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_rounds_$tag.json 2> gpurun_out/s2_rounds_$tag.err || { tail -5 gpurun_out/s2_rounds_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_rounds_$tag.json"))
print("$tag ms/step", d["ms_per_step"], flush=True)
PY
}
for wl in c4 cbox dragon table subsurface; do
  for r in 0 16,16 24,16 32,16 48,16 24,8 32,8 32,32 48,32; do run ${wl}_$r $wl NORI_HIP_FINISH_ROUNDS=$r || exit 1; done
done
NORI_HIP_FINISH_ROUNDS=32,16 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
