#!/bin/bash
# sensitivity of the shade stage to occupancy and run length (table and c5), kernel stats per setting
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/s2_knob_$tag.json 2> gpurun_out/s2_knob_$tag.err || { tail -5 gpurun_out/s2_knob_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_knob_$tag.json"))
print("$tag ms/step", d["ms_per_step"], flush=True)
PY
}
for w in 3 4 5 6 8; do run table_w$w table NORI_HIP_SHADE_WAVES=$w || exit 1; done
for b in 4 8 16 24 32; do run table_b$b table NORI_HIP_SHADE_BPC=$b || exit 1; done
for w in 3 5 6; do run cbox_w$w cbox NORI_HIP_SHADE_WAVES=$w || exit 1; done
for b in 4 8 16 24; do run cbox_b$b cbox NORI_HIP_SHADE_BPC=$b || exit 1; done
