#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputest.log 2>&1 || { tail -40 gpurun_out/gputest.log; exit 1; }
tail -2 gpurun_out/gputest.log
rm -f gpurun_out/sweep_fw.txt; bash scratch/sweep_fw.sh
