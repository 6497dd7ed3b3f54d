#!/bin/bash
mkdir -p gpurun_out
for cfg in "NORI_HIP_SPEC=0" "NORI_HIP_SPEC=1" "NORI_HIP_BOX2=36" "NORI_HIP_SPEC=1 NORI_HIP_BOX2=36" "NORI_HIP_SPEC=1 NORI_HIP_BOX2=24" "NORI_HIP_THR=1,1,8" "NORI_HIP_THR=1,1,20" "NORI_HIP_LBVH_LEAF=3 NORI_HIP_LBVH_ALIGN=0" "NORI_HIP_SPEC=1 NORI_HIP_LBVH_LEAF=3 NORI_HIP_LBVH_ALIGN=0"; do
  echo "CFG $cfg" >> gpurun_out/sweep_spec4.txt
  env $cfg timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_spec4.txt 2>&1 || exit 1
done
