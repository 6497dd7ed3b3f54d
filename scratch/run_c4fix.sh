#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_c4.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/c4fix_tests.log 2>&1 || { tail -30 gpurun_out/c4fix_tests.log; exit 1; }
tail -2 gpurun_out/c4fix_tests.log
for cfg in "NORI_HIP_CULL=0 NORI_HIP_ANYORDER=near" "NORI_HIP_CULL=0 NORI_HIP_ANYORDER=far" "NORI_HIP_CULL=1 NORI_HIP_ANYORDER=near" "NORI_HIP_CULL=1 NORI_HIP_ANYORDER=far" "NORI_HIP_CULL=1 NORI_HIP_XCD=0" "NORI_HIP_CULL=1 NORI_HIP_LBVH_ALIGN=0" "NORI_HIP_CULL=1 NORI_HIP_SPEC=1" "NORI_HIP_CULL=1 NORI_HIP_BOX2=0"; do
  echo "CFG $cfg" >> gpurun_out/c4fix_bench.txt
  env $cfg timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/c4fix_bench.txt 2>&1 || exit 1
done
