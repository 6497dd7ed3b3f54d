#!/bin/bash
# the automatic choice (two pipelines from NORI_HIP_TWIN_MIN samples per call on scenes without images or subsurface shapes) against one pipeline
set -o pipefail
mkdir -p gpurun_out
run() { # tag workload args... -- env...
  tag=$1; wl=$2; spp=$3; shift; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --spp $spp --no-cpu --no-secondary --steps 6 --warmup 2 > gpurun_out/s2_ta_$tag.json 2> gpurun_out/s2_ta_$tag.err || { tail -5 gpurun_out/s2_ta_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_ta_$tag.json"))
print("$tag ms/step", d["ms_per_step"], "pipelines", d["roofline"].get("pipelines"), flush=True)
PY
}
for spp in 4 8 16 32 64; do
  run cbox_${spp}_p1 cbox $spp NORI_HIP_PIPELINES=1 || exit 1
  run cbox_${spp}_p2 cbox $spp NORI_HIP_PIPELINES=2 || exit 1
done
for spp in 4 16; do
  run dragon_${spp}_p1 dragon $spp NORI_HIP_PIPELINES=1 || exit 1
  run dragon_${spp}_p2 dragon $spp NORI_HIP_PIPELINES=2 || exit 1
done
for wl in cbox table dragon c5 subsurface c4; do
  run ${wl}_auto $wl 0 || exit 1
done
