#!/bin/bash
# usage: scratch/r3_prof_all.sh <workload>...: scratch/profile_r03.sh for each, log per workload under gpurun_out/
for w in "$@"; do
  bash scratch/profile_r03.sh $w r03_$w > gpurun_out/prof_r03_$w.log 2>&1 || { echo "$w FAILED"; tail -5 gpurun_out/prof_r03_$w.log; exit 1; }
  python3 - <<PY
import json; d=json.load(open("gpurun_out/prof_r03_$w/bench.json")); r=d["roofline"]
print("$w", d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms"], "frac", r["frac"], "traffic_frac", r.get("traffic_frac"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
done
