#!/bin/bash
# usage: scratch/profile_r03.sh <workload> <tag> [bench args...]: rocprofv3 kernel stats + six PMC passes of one bench step,
# summaries under gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/)
w=$1; tag=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 python3 -u bench.py --workload $w --steps 3 --warmup 1 "$@" > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -3 $out/bench.err; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-secondary "$@" > $out/kt.json 2> $out/kt.err || { echo "kernel trace failed"; tail -3 $out/kt.err; exit 1; }
cp $out/kt/*/*kernel_stats.csv $out/kernel_stats.csv
i=0
while read -r counters; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $counters --output-format csv -d $out/pmc_$i -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu --no-secondary "$@" > $out/pmc_$i.json 2> $out/pmc_$i.err || { echo "pass $i failed"; tail -3 $out/pmc_$i.err; exit 1; }
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE GRBM_TA_BUSY
TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum
FETCH_SIZE
WRITE_SIZE
PASSES
python3 scratch/profile_r03_sum.py $w $tag
