#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/w8b_bench.txt
NORI_HIP_WIDE=8 timeout -k 10 400 python -u -m pytest tests/test_gpu_c4.py tests/test_gpu_parity.py -m gpu -x -q -k "c4 or device_built" --timeout 120 --timeout-method thread > gpurun_out/w8_tests.log 2>&1 || { tail -30 gpurun_out/w8_tests.log; exit 1; }
tail -2 gpurun_out/w8_tests.log
for cfg in "NORI_HIP_WIDE=4" "NORI_HIP_WIDE=8" "NORI_HIP_WIDE=8 NORI_HIP_ANYORDER=near"; do
  echo "CFG $cfg" >> gpurun_out/w8b_bench.txt
  env $cfg timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/w8b_bench.txt 2>&1 || exit 1
done
