import csv, glob, collections, sys
tag = sys.argv[1]
tot = collections.defaultdict(float)
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/*/*counter_collection.csv' % tag)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = 'k_render' if 'k_render' in k else ('k_shade' if 'k_shade' in k else ('shadow' if 'k_trace_pool<1' in k else ('trace' if 'k_trace_pool<2' in k else ('extend' if 'k_trace_pool<0' in k else ('finish' if 'k_finish' in k else None)))))
        if name: tot[(name, r['Counter_Name'])] += float(r['Counter_Value'])
ks = sorted(set(k for k, _ in tot))
for k in ks:
    g = lambda c: tot.get((k, c), 0.0)
    wc = g('SQ_WAVE_CYCLES') or 1
    print('%-8s waves %8d  VALU util %5.1f%%  active %4.1f%% wait_any %4.1f%% wait_inst %4.1f%%  insts: VALU %.3g SALU %.3g VMEM_RD %.3g WR %.3g  TCC hit %.3g miss %.3g  FETCH %.1f GB WRITE %.1f GB  TCP acc %.3g  TA busy %.3g / GUI %.3g' % (
        k, g('SQ_WAVES'), 100 * g('SQ_THREAD_CYCLES_VALU') / (64 * (g('SQ_ACTIVE_INST_VALU') or 1)), 100 * g('SQ_ACTIVE_INST_ANY') / wc, 100 * g('SQ_WAIT_ANY') / wc,
        100 * g('SQ_WAIT_INST_ANY') / wc, g('SQ_INSTS_VALU'), g('SQ_INSTS_SALU'), g('SQ_INSTS_VMEM_RD'), g('SQ_INSTS_VMEM_WR'), g('TCC_HIT'), g('TCC_MISS'),
        g('FETCH_SIZE') / 1e6, g('WRITE_SIZE') / 1e6, g('TCP_TOTAL_CACHE_ACCESSES'), g('TA_TA_BUSY'), g('GRBM_GUI_ACTIVE')))
