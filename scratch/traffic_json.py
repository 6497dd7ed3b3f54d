"""usage: python3 scratch/traffic_json.py <pmc tag> <workload> -- profiles/r01_<workload>_traffic.json from the FETCH_SIZE / WRITE_SIZE
passes of scratch/pmc.sh (passes 5 and 6), for the trace-stage kernel of the wavefront pipeline.  rocprofv3 reports both in KB."""
import csv, glob, json, sys
tag, workload = sys.argv[1], sys.argv[2]
tot, calls = {}, {}
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/*/*counter_collection.csv' % tag)):
    for r in csv.DictReader(open(f)):
        if 'k_trace_pool<2' in r['Kernel_Name'] and r['Counter_Name'] in ('SQ_INSTS_VALU', 'GRBM_GUI_ACTIVE'):
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if 'k_trace_pool<2' in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            c = r['Counter_Name']
            tot[c] = tot.get(c, 0.0) + float(r['Counter_Value']) * 1024.0
            calls.setdefault(c, set()).add(r['Dispatch_Id'])
n = len(calls['FETCH_SIZE'])
assert n == len(calls['WRITE_SIZE']), (n, len(calls['WRITE_SIZE']))
fetch, write = 2.0 * tot['FETCH_SIZE'] / n, tot['WRITE_SIZE'] / n
out = {"workload": workload, "kernel": "k_trace_pool<TK_BOTH>", "launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": fetch + write,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (scratch/pmc.sh), bench.py --steps 1 --warmup 0; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
# share of the kernel's cycles in which a SIMD's vector ALU is issuing: wave-instructions x 4 cycles (64 lanes on a 16-lane
# SIMD) over 1024 SIMDs, against GRBM_GUI_ACTIVE (which rocprofv3 sums over the 8 XCDs)
if tot.get('SQ_INSTS_VALU') and tot.get('GRBM_GUI_ACTIVE'):
    out["valu_issue_busy"] = round(tot['SQ_INSTS_VALU'] * 4.0 / 1024.0 / (tot['GRBM_GUI_ACTIVE'] / 8.0), 4)
    out["valu_wave_instructions_per_launch"] = tot['SQ_INSTS_VALU'] / n
json.dump(out, open('gpurun_out/r01_%s_traffic.json' % workload, 'w'), indent=1)
print(json.dumps(out))
