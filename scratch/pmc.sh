#!/bin/bash
# usage: scratch/pmc.sh <tag> <env assignments...> ; runs the PMC passes on a short bench, one rocprofv3 run each
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in "$@"; do export "$e"; done
i=0
while read -r counters; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu $BENCH_ARGS > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || { echo "pass $i failed"; exit 1; }
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE GRBM_TA_BUSY
TCC_HIT TCC_MISS TCC_READ TCC_WRITE
FETCH_SIZE
WRITE_SIZE
PASSES
python3 scratch/pmc_sum.py $tag
