#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/sweep_fin3.txt
for f in 0 800000 1100000 2200000 3000000; do
  echo "W=cbox FINISH=$f" >> gpurun_out/sweep_fin3.txt
  NORI_HIP_FINISH=$f timeout -k 10 120 python bench.py --workload cbox --no-cpu --steps 15 --warmup 2 >> gpurun_out/sweep_fin3.txt 2>&1 || exit 1
done
