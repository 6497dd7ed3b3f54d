"""usage: profile_r02_sum.py <workload> <tag> -- per-kernel summary of the passes of scratch/profile_r03.sh and the traffic JSON
(2 x FETCH_SIZE + WRITE_SIZE per launch of the trace-stage kernel; rocprofv3 reports both in KB; doubling: profiles/r02_fetchprobe.txt)"""
import collections, csv, glob, json, os, sys
w, tag = sys.argv[1], sys.argv[2]
out = 'gpurun_out/prof_%s' % tag
def short(k):
    for n in ('k_trace_pool4', 'k_trace_pool', 'k_shade', 'k_finish', 'k_render', 'k_compact_pool', 'k_sort', 'k_bin'):
        if n in k:
            return n
    return None
tot = collections.defaultdict(float); calls = collections.defaultdict(set)
for f in sorted(glob.glob(out + '/pmc_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if n:
            tot[(n, r['Counter_Name'])] += float(r['Counter_Value']); calls[(n, r['Counter_Name'])].add(r['Dispatch_Id'])
lines = []
for k in sorted(set(k for k, _ in tot)):
    g = lambda c: tot.get((k, c), 0.0)
    wc = g('SQ_WAVE_CYCLES') or 1
    hit, miss = g('TCC_HIT_sum'), g('TCC_MISS_sum')
    lines.append('%-14s launches %4d waves %8d  VALU lane util %5.1f%%  wave life: issuing %4.1f%% waiting %4.1f%% (on an instruction issue %4.1f%%)  insts: VALU %.3g SALU %.3g VMEM_RD %.3g WR %.3g  '
                 'L2 hit %.3g miss %.3g (%.1f%% hit)  FETCH_SIZE %.2f GB WRITE_SIZE %.2f GB  GUI_ACTIVE %.3g' % (
        k, len(calls[(k, 'SQ_WAVES')]), g('SQ_WAVES'), 100 * g('SQ_THREAD_CYCLES_VALU') / (64 * (g('SQ_ACTIVE_INST_VALU') or 1)), 100 * g('SQ_ACTIVE_INST_ANY') / wc, 100 * g('SQ_WAIT_ANY') / wc,
        100 * g('SQ_WAIT_INST_ANY') / wc, g('SQ_INSTS_VALU'), g('SQ_INSTS_SALU'), g('SQ_INSTS_VMEM_RD'), g('SQ_INSTS_VMEM_WR'), hit, miss, 100 * hit / max(1.0, hit + miss),
        g('FETCH_SIZE') / 1e6, g('WRITE_SIZE') / 1e6, g('GRBM_GUI_ACTIVE')))
open(out + '/pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
bench = json.load(open(out + '/bench.json'))
kname = bench['roofline']['kernel']
key = 'k_trace_pool4' if kname.startswith('k_trace_pool4') else 'k_trace_pool'
n = len(calls[(key, 'FETCH_SIZE')])
if n:
    fetch, write = 2.0 * tot[(key, 'FETCH_SIZE')] * 1024.0 / n, tot[(key, 'WRITE_SIZE')] * 1024.0 / max(1, len(calls[(key, 'WRITE_SIZE')]))
    tj = {"workload": w, "kernel": kname, "bvh": "device" if "device-built" in bench['config']['bvh'] else "host", "launches": n,
          "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
          "algorithmic_bytes_per_launch": bench['roofline']['algorithmic_bytes_per_launch'],
          "traffic_over_algorithmic": round((fetch + write) / bench['roofline']['algorithmic_bytes_per_launch'], 3),
          "l2_hit_rate": round(tot[(key, 'TCC_HIT_sum')] / max(1.0, tot[(key, 'TCC_HIT_sum')] + tot[(key, 'TCC_MISS_sum')]), 4),
          "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (scratch/profile_r03.sh), bench.py --steps 1 --warmup 0; traffic = 2 x FETCH_SIZE + WRITE_SIZE: "
                  "gfx950 tallies every 128-byte line request at 64 B, for scattered reads as for streaming ones (profiles/r02_fetchprobe.txt)"}
    if tot.get((key, 'SQ_INSTS_VALU')) and tot.get((key, 'GRBM_GUI_ACTIVE')):
        tj["valu_issue_busy"] = round(tot[(key, 'SQ_INSTS_VALU')] * 4.0 / 1024.0 / (tot[(key, 'GRBM_GUI_ACTIVE')] / 8.0), 4)
    json.dump(tj, open(out + '/traffic.json', 'w'), indent=1)
    print(json.dumps(tj))
print(open(out + '/bench.json').read())
os.system("head -12 %s/kernel_stats.csv" % out)
