#!/bin/bash
mkdir -p gpurun_out
echo "C4" >> gpurun_out/sweep_fw.txt
timeout -k 10 120 python bench.py --no-cpu --steps 3 >> gpurun_out/sweep_fw.txt 2>&1 || exit 1
for w in table dragon subsurface; do
  echo "W=$w" >> gpurun_out/sweep_fw.txt
  timeout -k 10 120 python bench.py --workload $w --no-cpu --steps 5 >> gpurun_out/sweep_fw.txt 2>&1 || exit 1
done
