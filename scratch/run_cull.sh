#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputest.log 2>&1 || { tail -30 gpurun_out/gputest.log; exit 1; }
tail -3 gpurun_out/gputest.log
for c in 0 1; do
  echo "CULL=$c" >> gpurun_out/cull_bench.txt
  NORI_HIP_CULL=$c timeout -k 10 120 python bench.py --no-cpu --steps 3 >> gpurun_out/cull_bench.txt 2>&1 || exit 1
  for w in table dragon; do
    echo "CULL=$c $w" >> gpurun_out/cull_bench.txt
    NORI_HIP_CULL=$c timeout -k 10 120 python bench.py --workload $w --no-cpu --steps 5 >> gpurun_out/cull_bench.txt 2>&1 || exit 1
  done
done
