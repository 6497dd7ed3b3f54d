#!/bin/bash
# usage: scratch/ab.sh libA.so libB.so ... -- benches each library (copied over the in-tree one) on table, cbox, dragon, twice
set -o pipefail
for rep in 1 2; do for lib in "$@"; do
  cp "$lib" cuda-path-tracer_amd/libnori_hip.so || exit 1
  for w in ${AB_WORKLOADS:-table cbox dragon}; do echo "$(basename $lib)_$w NORI_HIP_X=0" | BENCH_ARGS="--workload $w" bash scratch/sweep.sh || exit 1; done
done; done
