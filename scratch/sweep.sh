#!/bin/bash
# usage: scratch/sweep.sh  -- bench.py over pipeline knobs; prints ms_per_step for each
set -o pipefail
mkdir -p gpurun_out
run() { # label env...
  local label="$1"; shift
  local out; out=$(env "$@" timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu $BENCH_ARGS 2>gpurun_out/sweep_err.txt | tail -1) || { echo "$label FAILED"; tail -5 gpurun_out/sweep_err.txt; return 1; }
  python - "$label" "$out" <<'PY'
import sys, json
j = json.loads(sys.argv[2]); print('%-40s %8.2f ms  %8.1f %s  frac %.3f' % (sys.argv[1], j['ms_per_step'], j['value'], j['unit'], j['roofline']['frac']))
PY
  grep "profile" gpurun_out/sweep_err.txt | tail -2 | cut -c1-330 || true
}
while read -r label envs; do
  [ -z "$label" ] && continue
  run "$label" $envs || exit 1
done
