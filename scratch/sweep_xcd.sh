#!/bin/bash
# C4 step time against the XCD-aware deal of the trace stage (chunks of the sorted order per XCD; 0 = plain round-robin)
# and against the position of the direction octant in the sort key
mkdir -p gpurun_out
for cfg in "128 0" "255 0" "64 9" "64 8" "64 7" "64 6" "64 5" "0 7"; do
  set -- $cfg
  echo "XCD=$1 KEYMODE=$2" >> gpurun_out/sweep_xcd.txt
  NORI_HIP_XCD=$1 NORI_HIP_KEYMODE=$2 timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_xcd.txt 2>&1 || exit 1
done
