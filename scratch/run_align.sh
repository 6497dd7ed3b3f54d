#!/bin/bash
mkdir -p gpurun_out
NORI_HIP_LBVH_ALIGN=1 NORI_HIP_LBVH_LEAF=2 timeout -k 10 400 python -u -m pytest tests/test_gpu_c4.py tests/test_gpu_parity.py -m gpu -x -q -k "c4 or device_built" --timeout 120 --timeout-method thread > gpurun_out/align_tests.log 2>&1 || { tail -30 gpurun_out/align_tests.log; exit 1; }
tail -3 gpurun_out/align_tests.log
for a in 0 1; do
  echo "ALIGN=$a" >> gpurun_out/align_bench.txt
  NORI_HIP_LBVH_ALIGN=$a timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/align_bench.txt 2>&1 || exit 1
done
