"""gpurun_out/final/ (written by tools/gpu_final.sh on the GPU box) -> the round's files under profiles/."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'final')
DST = os.path.join(ROOT, 'profiles')
R = sys.argv[1] if len(sys.argv) > 1 else 'r1'


def one(pattern):
    fs = glob.glob(os.path.join(SRC, pattern))
    if not fs:
        raise SystemExit('missing ' + pattern)
    return max(fs, key=os.path.getmtime)


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
        launches[r['Kernel_Name']].add(r['Dispatch_Id'])
    return agg, {k: len(v) for k, v in launches.items()}


line = open(os.path.join(SRC, 'bench_default.json')).read().strip()
json.loads(line)
open(os.path.join(DST, R + '_bench_default.json'), 'w').write(line + '\n')
shutil.copy(one('stats/*/*_kernel_stats.csv'), os.path.join(DST, R + '_kernel_stats.csv'))

traffic = {}
for name, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg, n = counters(one(name + '/*/*_counter_collection.csv'))
    with open(os.path.join(DST, '%s_pmc_%s_size.csv' % (R, name)), 'w') as f:
        f.write('Kernel_Name,Launches,%s_total_KB_as_reported,%s_KB_per_launch\n' % (ctr, ctr))
        for k in sorted(agg, key=lambda k: -agg[k][ctr]):
            f.write('"%s",%d,%.1f,%.1f\n' % (k, n[k], agg[k][ctr], agg[k][ctr] / n[k]))
    for k in agg:
        if k.startswith('void exec::trace_kernel<false, false'):
            traffic[ctr] = agg[k][ctr] / n[k]
            traffic['kernel'] = k.split('(')[0]
            traffic['launches'] = n[k]
out = {
    'kernel': traffic['kernel'],
    'workload': 'bunny_box 1024x1024, max_bounces 4 (python bench.py --steps 1 --warmup 0 --spp 2)',
    'launches': traffic['launches'],
    'FETCH_SIZE_KB_per_launch_as_reported': traffic['FETCH_SIZE'],
    'fetch_correction_gfx950': 2.0,
    'WRITE_SIZE_KB_per_launch': traffic['WRITE_SIZE'],
    'hbm_bytes_per_launch': (traffic['FETCH_SIZE'] * 2.0 + traffic['WRITE_SIZE']) * 1024.0,
    'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section: '
              'gfx950 FETCH_SIZE counts 64 B per 128-B request, hence x2)',
    'source_files': ['profiles/%s_pmc_fetch_size.csv' % R, 'profiles/%s_pmc_write_size.csv' % R],
}
json.dump(out, open(os.path.join(DST, R + '_traffic.json'), 'w'), indent=1)

agg, n = counters(one('sq/*/*_counter_collection.csv'))
cols = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
        'SQ_INSTS_VALU', 'SQ_THREAD_CYCLES_VALU']
with open(os.path.join(DST, R + '_pmc_sq.csv'), 'w') as f:
    f.write('Kernel_Name,Launches,' + ','.join(cols) + ',VALU_lane_utilisation\n')
    for k in sorted(agg, key=lambda k: -agg[k]['SQ_WAVE_CYCLES']):
        v = agg[k]
        util = v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_ACTIVE_INST_VALU']) if v['SQ_ACTIVE_INST_VALU'] else 0.0
        f.write('"%s",%d,' % (k, n[k]) + ','.join('%d' % v[c] for c in cols) + ',%.3f\n' % util)
print('profiles/%s_* refreshed; traffic %.1f MB per closest-hit launch' % (R, out['hbm_bytes_per_launch'] / 1e6))
