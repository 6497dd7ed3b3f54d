#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/keys_bench.txt
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputest.log 2>&1 || { tail -40 gpurun_out/gputest.log; exit 1; }
tail -2 gpurun_out/gputest.log
for k in kernel fused kernel fused; do
  echo "KEYS=$k" >> gpurun_out/keys_bench.txt
  NORI_HIP_KEYS=$k timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 4 >> gpurun_out/keys_bench.txt 2>&1 || exit 1
done
