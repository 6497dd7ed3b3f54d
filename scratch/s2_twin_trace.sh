#!/bin/bash
# kernel timeline of a render call with two pipelines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/twin_trace
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/twin_trace -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 > gpurun_out/twin_trace.json 2> gpurun_out/twin_trace.log || { tail -5 gpurun_out/twin_trace.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/twin_trace/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
def short(n):
    for k in ('k_trace_pool4q', 'k_shade', 'k_finish', 'DeviceRadixSort', 'k_iter_begin', 'k_pool_init', 'k_compact'):
        if k in n: return k
    return n[:30]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows)
t0 = ev[0][0]
fin = [e for e in ev if e[2] == 'k_finish']
out = open('gpurun_out/twin_timeline.txt', 'w')
for s, e, k, q in ev:
    if k in ('k_trace_pool4q', 'k_shade', 'k_finish', 'DeviceRadixSort'):
        out.write("%10.3f %10.3f %8.3f %-16s q=%s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, q))
out.close()
for s, e, k, q in fin: print("finish %.3f -> %.3f (%.3f ms) q=%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q))
PY
cat gpurun_out/twin_trace.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])"
