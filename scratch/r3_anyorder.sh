#!/bin/bash
for w in cbox table dragon c5; do
  for o in near reference; do
    NORI_HIP_ANYORDER=$o timeout -k 10 300 python bench.py --workload $w --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '$o', d['ms_per_step'], d['value'], r['kernel_ms'], r['boxes_tested_per_ray'], r['prims_tested_per_ray'])" || exit 1
  done
done
