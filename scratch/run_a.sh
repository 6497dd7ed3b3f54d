#!/bin/bash
# round 2, first GPU call: GPU tests (incl. the WIP no-alloc test) and the FETCH_SIZE probe
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2a/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r2a/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2a/noalloc.log 2>&1; rc=$?
tail -5 gpurun_out/r2a/noalloc.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 ./scratch/fetchprobe > gpurun_out/r2a/fetchprobe.txt 2>&1 || exit 1
cat gpurun_out/r2a/fetchprobe.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2a/pmc_fetch -- ./scratch/fetchprobe > gpurun_out/r2a/pmc_fetch.log 2>&1 || echo "FETCH_SIZE pass failed"
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d gpurun_out/r2a/pmc_rdreq -- ./scratch/fetchprobe > gpurun_out/r2a/pmc_rdreq.log 2>&1 || echo "RDREQ pass failed"
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/r2a/pmc_hit -- ./scratch/fetchprobe > gpurun_out/r2a/pmc_hit.log 2>&1 || echo "HIT pass failed"
python3 - <<'PY'
import csv, glob
for d in ("pmc_fetch", "pmc_rdreq", "pmc_hit"):
    for f in glob.glob("gpurun_out/r2a/%s/*/*counter_collection.csv" % d):
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            print(d, r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
PY
