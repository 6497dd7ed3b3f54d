#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/keys_bench.txt
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputest.log 2>&1 || { tail -40 gpurun_out/gputest.log; exit 1; }
tail -2 gpurun_out/gputest.log
echo "c4" >> gpurun_out/keys_bench.txt
timeout -k 10 120 python bench.py --no-cpu --steps 4 >> gpurun_out/keys_bench.txt 2>&1 || exit 1
for w in table dragon; do
  echo "$w" >> gpurun_out/keys_bench.txt
  timeout -k 10 120 python bench.py --workload $w --no-cpu --steps 8 >> gpurun_out/keys_bench.txt 2>&1 || exit 1
done
