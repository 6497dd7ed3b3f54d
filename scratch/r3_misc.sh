#!/bin/bash
# step times of the workloads other than C4 (no CPU baseline)
for w in "$@"; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$w\", d[\"ms_per_step\"], d[\"value\"])" || exit 1
done
