#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u scratch/parity_fractions.py || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread -k "not full_size_hits" > gpurun_out/r3_gputest2.log 2>&1 || { tail -40 gpurun_out/r3_gputest2.log; exit 1; }
tail -3 gpurun_out/r3_gputest2.log
for w in c4 cbox table dragon; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu --no-secondary --steps 4 > gpurun_out/r3_b2_$w.json 2> gpurun_out/r3_b2_$w.err || { tail -5 gpurun_out/r3_b2_$w.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/r3_b2_$w.json")); print("$w", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"])
PY
done
