"""gpurun_out/final_r2/ (written by tools/gpu_final_r2.sh on the GPU box) -> profiles/r2_*."""
import json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, 'gpurun_out', 'final_r2'), os.path.join(ROOT, 'profiles')
for src, dst in (('bench_default.json', 'r2_bench_default.json'), ('bench_living_room_standin.json', 'r2_bench_living_room_standin.json'),
                 ('bench_living_room_standin_envmap.json', 'r2_bench_living_room_standin_envmap.json'), ('bench_living_512spp.json', 'r2_bench_living_room_standin_512spp.json'), ('bench_two_workers.json', 'r2_bench_two_workers.json'),
                 ('kernel_stats.csv', 'r2_kernel_stats.csv'), ('kernel_stats_alone.csv', 'r2_kernel_stats_alone.csv'),
                 ('pmc_sq.csv', 'r2_pmc_sq.csv'), ('pmc_sq2.csv', 'r2_pmc_sq2.csv'), ('parity_report.jsonl', 'r2_parity_report.jsonl'),
                 ('small_loop.log', 'r2_small_loop.txt')):
    p = os.path.join(SRC, src)
    if not os.path.exists(p):
        print('missing', src)
        continue
    if src.endswith('.json'):
        json.loads(open(p).read())
    shutil.copy(p, os.path.join(DST, dst))
d = json.load(open(os.path.join(DST, 'r2_bench_default.json')))
print('value %.2f Msamples/s, roofline frac %.3f' % (d['value'], d['roofline']['frac']))
