#!/bin/bash
# kernel timeline of two contexts rendering side by side (which stages overlap)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dual_trace -- python3 scratch/s2_dual.py c4 1 2 > gpurun_out/dual_trace.log 2>&1 || { tail -5 gpurun_out/dual_trace.log; exit 1; }
tail -1 gpurun_out/dual_trace.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/dual_trace/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
def short(n):
    for k in ('k_trace_pool4q', 'k_shade', 'k_finish', 'DeviceRadixSort', 'k_iter_begin', 'k_pool_init', 'k_compact'):
        if k in n: return k
    return n[:30]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
ev.sort()
# the last "dual" step: take the last 400 events and print a compact timeline with relative times in ms
t_end = max(e[1] for e in ev)
sel = [e for e in ev if e[2] in ('k_trace_pool4q', 'k_shade', 'k_finish')]
# find the dual phase: events whose queue differs ... print the last 2.2 steps worth
out = open('gpurun_out/dual_timeline.txt', 'w')
for s, e, k, q in sel:
    out.write("%10.3f %10.3f %8.3f %-16s q=%s\n" % ((s - ev[0][0]) / 1e6, (e - ev[0][0]) / 1e6, (e - s) / 1e6, k, q))
out.close()
print("events", len(sel))
PY
