#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/sweep_fin3.txt
for w in cbox table dragon; do for f in 0 1500000 2097152 3000000 4194304; do
  echo "W=$w FINISH=$f" >> gpurun_out/sweep_fin3.txt
  NORI_HIP_FINISH=$f timeout -k 10 120 python bench.py --workload $w --no-cpu --steps 5 >> gpurun_out/sweep_fin3.txt 2>&1 || exit 1
done; done
for f in 0 3000000 4194304; do
  echo "W=c4 FINISH=$f" >> gpurun_out/sweep_fin3.txt
  NORI_HIP_FINISH=$f timeout -k 10 120 python bench.py --no-cpu --no-secondary --steps 3 >> gpurun_out/sweep_fin3.txt 2>&1 || exit 1
done
