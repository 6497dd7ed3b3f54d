#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/sweep_pl.txt
for p in 1 4 12 24; do
  echo "PRIM_LANES=$p" >> gpurun_out/sweep_pl.txt
  NORI_HIP_PRIM_LANES=$p timeout -k 10 120 python bench.py --no-cpu --steps 3 >> gpurun_out/sweep_pl.txt 2>&1 || exit 1
done
