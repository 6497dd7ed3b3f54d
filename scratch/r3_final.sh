#!/bin/bash
# default bench line, the group driver and the strong-scaling prediction with the final library
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err || { tail -5 gpurun_out/r3_bench_default.err; exit 1; }
cat gpurun_out/r3_bench_default.json
timeout -k 10 200 python bench.py --driver group --gpus 8 --no-cpu > gpurun_out/r3_bench_group_c4.json 2> gpurun_out/r3_bench_group.err || { tail -5 gpurun_out/r3_bench_group.err; exit 1; }
cat gpurun_out/r3_bench_group_c4.json
timeout -k 10 200 python bench.py --driver group --gpus 8 --workload c5 --no-cpu > gpurun_out/r3_bench_group_c5.json 2>> gpurun_out/r3_bench_group.err || { tail -5 gpurun_out/r3_bench_group.err; exit 1; }
cat gpurun_out/r3_bench_group_c5.json
for n in 2 4 8; do
  timeout -k 10 200 python bench.py --as-rank $n --steps 2 --warmup 1 > gpurun_out/r3_asrank_c4_$n.json 2>> gpurun_out/r3_asrank.err || { tail -5 gpurun_out/r3_asrank.err; exit 1; }
  cat gpurun_out/r3_asrank_c4_$n.json
  timeout -k 10 200 python bench.py --as-rank $n --partition tiles --steps 2 --warmup 1 > gpurun_out/r3_asrank_c4_tiles_$n.json 2>> gpurun_out/r3_asrank.err || { tail -5 gpurun_out/r3_asrank.err; exit 1; }
  cat gpurun_out/r3_asrank_c4_tiles_$n.json
done
timeout -k 10 200 python bench.py --as-rank 8 --workload c5 --steps 2 --warmup 1 > gpurun_out/r3_asrank_c5_8.json 2>> gpurun_out/r3_asrank.err || { tail -5 gpurun_out/r3_asrank.err; exit 1; }
cat gpurun_out/r3_asrank_c5_8.json
timeout -k 10 200 python bench.py --as-rank 8 --workload c5 --partition tiles --steps 2 --warmup 1 > gpurun_out/r3_asrank_c5_tiles_8.json 2>> gpurun_out/r3_asrank.err || { tail -5 gpurun_out/r3_asrank.err; exit 1; }
cat gpurun_out/r3_asrank_c5_tiles_8.json
