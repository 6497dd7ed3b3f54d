#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout -k 10 600 python -u -m pytest tests/test_gpu_c4.py tests/test_gpu_parity.py::test_spp_slices_and_tiles_accumulate_to_the_same_film -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2b/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r2b/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u bench.py --steps 2 --warmup 1 > gpurun_out/r2b/bench_c4.json 2> gpurun_out/r2b/bench_c4.err; rc=$?
tail -3 gpurun_out/r2b/bench_c4.err; cat gpurun_out/r2b/bench_c4.json
[ $rc -eq 0 ] || exit $rc
