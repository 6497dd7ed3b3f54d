#!/bin/bash
# profiles of the workloads whose default changed to two pipelines
set -o pipefail
for w in cbox table dragon c5; do
  bash scratch/profile_r03.sh $w ${w}_final > gpurun_out/prof_${w}_final.log 2>&1 || { tail -5 gpurun_out/prof_${w}_final.log; exit 1; }
  echo "$w done"; head -c 300 gpurun_out/prof_${w}_final/bench.json; echo
done
