#!/bin/bash
# usage: scratch/s2_pmc2.sh <tag> <workload> <env...>: the two SQ passes only (instruction counts, lane utilisation, wave life)
tag=$1; wl=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in "$@"; do export "$e"; done
i=0
while read -r counters; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu --no-secondary > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_$i.err; exit 1; }
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
PASSES
python3 scratch/pmc_sum.py $tag
