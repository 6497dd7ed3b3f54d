#!/bin/bash
# the tail kernel built for three or four waves per SIMD
mkdir -p gpurun_out; rm -f gpurun_out/sweep_fw.txt
for v in 3 4; do
  echo "FW=$v c4" >> gpurun_out/sweep_fw.txt
  NORI_HIP_FINISH_WAVES=$v timeout -k 10 120 python bench.py --no-cpu --steps 3 >> gpurun_out/sweep_fw.txt 2>&1 || exit 1
  for w in table dragon; do
    echo "FW=$v $w" >> gpurun_out/sweep_fw.txt
    NORI_HIP_FINISH_WAVES=$v timeout -k 10 120 python bench.py --workload $w --no-cpu --steps 5 >> gpurun_out/sweep_fw.txt 2>&1 || exit 1
  done
done
