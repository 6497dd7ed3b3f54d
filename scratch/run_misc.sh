#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/misc_bench.txt
for args in "--partition tiles" "--workload c5" "--workload proj --steps 1 --warmup 0" "--workload subsurface" "--workload dragon" "--workload cbox --bvh device" "--workload table --bvh device" "--workload cbox --partition tiles" ; do
  echo "ARGS $args" >> gpurun_out/misc_bench.txt
  timeout -k 10 200 python bench.py $args --no-cpu --no-secondary >> gpurun_out/misc_bench.txt 2>&1 || exit 1
done
