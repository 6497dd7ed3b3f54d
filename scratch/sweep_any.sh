#!/bin/bash
mkdir -p gpurun_out
for w in cbox table dragon subsurface; do for a in near far; do
  echo "W=$w ANYORDER=$a" >> gpurun_out/sweep_any.txt
  NORI_HIP_ANYORDER=$a timeout -k 10 120 python bench.py --workload $w --no-cpu --steps 5 >> gpurun_out/sweep_any.txt 2>&1 || exit 1
done; done
for w in cbox table dragon; do for a in near far; do
  echo "W=$w/device ANYORDER=$a" >> gpurun_out/sweep_any.txt
  NORI_HIP_ANYORDER=$a timeout -k 10 120 python bench.py --workload $w --bvh device --no-cpu --steps 5 >> gpurun_out/sweep_any.txt 2>&1 || exit 1
done; done
