#!/bin/bash
# trace stage against its occupancy (workgroups of 4 waves per CU): how much of its time is latency that more rays in flight would hide
mkdir -p gpurun_out; rm -f gpurun_out/sweep_occ.txt
for b in 5 4 3; do
  echo "BPC=$b" >> gpurun_out/sweep_occ.txt
  NORI_HIP_TRACE_BPC=$b timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_occ.txt 2>&1 || exit 1
done
for t in 512 1024; do
  echo "THREADS=$t" >> gpurun_out/sweep_occ.txt
  NORI_HIP_TRACE_THREADS=$t timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_occ.txt 2>&1 || exit 1
done
