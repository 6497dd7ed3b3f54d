#!/bin/bash
# pool size per pipeline with two pipelines on scenes that live in the caches
set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-secondary --steps 5 --warmup 2 > gpurun_out/s2_tp_$tag.json 2> gpurun_out/s2_tp_$tag.err || { tail -5 gpurun_out/s2_tp_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2_tp_$tag.json"))
print("$tag ms/step", d["ms_per_step"], "launches", d["roofline"]["launches_per_step"], flush=True)
PY
}
run cbox_default cbox || exit 1
for p in 1572864 2097152 4194304 6291456; do run cbox_pool$p cbox NORI_HIP_POOL=$p || exit 1; done
run table_default table || exit 1
for p in 2097152 4194304 6291456; do run table_pool$p table NORI_HIP_POOL=$p || exit 1; done
run dragon_default dragon || exit 1
for p in 2097152 4194304 8388608; do run dragon_pool$p dragon NORI_HIP_POOL=$p || exit 1; done
for f in 262144 524288 1048576; do run cbox_finish$f cbox NORI_HIP_FINISH=$f || exit 1; done
