#!/bin/bash
mkdir -p gpurun_out
for cfg in "NORI_HIP_FINISH=2097152" "NORI_HIP_FINISH=1000000" "NORI_HIP_FINISH=3000000" "NORI_HIP_FINISH=4000000" "NORI_HIP_POOL=16777216" "NORI_HIP_POOL=16777216 NORI_HIP_FINISH=3000000" "NORI_HIP_POOL=12582912" "NORI_HIP_DYN=25" "NORI_HIP_DYN=8"; do
  echo "CFG $cfg" >> gpurun_out/sweep_fin2.txt
  env $cfg timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_fin2.txt 2>&1 || exit 1
done
