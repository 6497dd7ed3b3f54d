#!/bin/bash
# eight-box nodes: correctness (C4 tests at reduced and full size, device-tree fixture tests forced to width 8), then step times
mkdir -p gpurun_out
NORI_HIP_WIDE=8 timeout -k 10 400 python -u -m pytest tests/test_gpu_c4.py tests/test_gpu_parity.py -m gpu -x -q -k "c4 or device_built" --timeout 120 --timeout-method thread > gpurun_out/w8_tests.log 2>&1 || { tail -30 gpurun_out/w8_tests.log; exit 1; }
tail -3 gpurun_out/w8_tests.log
for w in 4 8; do
  echo "WIDE=$w" >> gpurun_out/w8_bench.txt
  NORI_HIP_WIDE=$w timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/w8_bench.txt 2>&1 || exit 1
done
