#!/bin/bash
# final collection of the round: C4 profile with the committed library, its one-pipeline line, then scratch/r3_final.sh (default line, group driver, as-rank table)
set -o pipefail
mkdir -p gpurun_out
bash scratch/profile_r03.sh c4 c4_final > gpurun_out/prof_c4_final.log 2>&1 || { tail -5 gpurun_out/prof_c4_final.log; exit 1; }
grep -c . gpurun_out/prof_c4_final/kernel_stats.csv
NORI_HIP_PIPELINES=1 timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r3_c4_one_pipeline.json 2> gpurun_out/r3_c4_one_pipeline.err || { tail -5 gpurun_out/r3_c4_one_pipeline.err; exit 1; }
cat gpurun_out/r3_c4_one_pipeline.json | cut -c1-400
bash scratch/r3_final.sh > gpurun_out/r3_final.log 2>&1 || { tail -5 gpurun_out/r3_final.log; exit 1; }
tail -c 1500 gpurun_out/r3_final.log
