#!/usr/bin/env python3
"""Sum the counters of a rocprofv3 --pmc run per kernel: python tools/pmc_summary.py <output dir> [name filter]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
fs = glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if flt not in k:
            continue
        k = k[:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
for k, v in sorted(agg.items()):
    n = len(disp[k])
    print('%s  (%d launches)' % (k, n))
    print('   ' + '  '.join('%s=%.4g' % (c, x / n) for c, x in sorted(v.items())))
    if 'SQ_ACTIVE_INST_VALU' in v and v.get('SQ_THREAD_CYCLES_VALU'):
        print('   lane util %.3f' % (v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_ACTIVE_INST_VALU'])))
    if v.get('SQ_WAVES') and 'SQ_INSTS_VALU' in v:
        print('   per wave: VALU %.0f  SALU %.0f  VMEM_RD %.0f  LDS %.0f' % (v['SQ_INSTS_VALU'] / v['SQ_WAVES'], v.get('SQ_INSTS_SALU', 0) / v['SQ_WAVES'], v.get('SQ_INSTS_VMEM_RD', 0) / v['SQ_WAVES'], v.get('SQ_INSTS_LDS', 0) / v['SQ_WAVES']))
    if v.get('SQ_WAVE_CYCLES'):
        print('   of wave cycles: waiting %.2f, wait_inst %.2f' % (v.get('SQ_WAIT_ANY', 0) / v['SQ_WAVE_CYCLES'], v.get('SQ_WAIT_INST_ANY', 0) / v['SQ_WAVE_CYCLES']))
