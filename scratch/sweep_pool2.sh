#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/sweep_pool2.txt
for p in 4194304 6291456 4194304 6291456 5242880 7340032; do
  echo "W=cbox POOL=$p" >> gpurun_out/sweep_pool2.txt
  NORI_HIP_POOL=$p timeout -k 10 120 python bench.py --workload cbox --no-cpu --steps 20 --warmup 2 >> gpurun_out/sweep_pool2.txt 2>&1 || exit 1
done
for p in 7700000 11520000; do
  echo "W=dragon POOL=$p" >> gpurun_out/sweep_pool2.txt
  NORI_HIP_POOL=$p timeout -k 10 120 python bench.py --workload dragon --no-cpu --steps 10 --warmup 2 >> gpurun_out/sweep_pool2.txt 2>&1 || exit 1
done
