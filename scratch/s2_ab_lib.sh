#!/bin/bash
# A/B of two builds of libnori_hip.so on one box: usage s2_ab_lib.sh <a.so> <b.so> <workload>...
a=$1; b=$2; shift; shift
cp cuda-path-tracer_amd/libnori_hip.so /tmp/keep.so
for w in "$@"; do for v in $a $b $a $b; do
  cp $v cuda-path-tracer_amd/libnori_hip.so
  timeout -k 10 200 python bench.py --workload $w --no-cpu --no-secondary --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('$w $v', d['ms_per_step'])"
done; done
cp /tmp/keep.so cuda-path-tracer_amd/libnori_hip.so
