#!/bin/bash
mkdir -p gpurun_out
for k in 0 6 8 12 14; do
  echo "SKIP=$k" >> gpurun_out/sweep_sortbits.txt
  NORI_HIP_SORT_SKIP=$k timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_sortbits.txt 2>&1 || exit 1
done
