#!/bin/bash
# the default bench (as the driver runs it) + the config-5 stand-in line
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
python -c "import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print(round(d['value'],2), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step; roofline frac', round(r['frac'],3), 'rays/s', r['rays_per_s'], 'hbm meas', r['hbm_frac_measured'], 'lane util', r['valu_lane_util']); print(json.dumps(r['kernels'], indent=0)[:3000]); print(d.get('cpu_baseline'))"
timeout 900 python bench.py --workload living_room_standin --spp 32 --steps 2 --no-cpu-baseline 2> $OUT/bench_living.err | tail -1 > $OUT/bench_living.json
python -c "import json; d=json.load(open('$OUT/bench_living.json')); r=d['roofline']; print('living', round(d['value'],2), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step; roofline frac', round(r['frac'],3)); print({k:(round(v['mean_launch_ms_alone'],3), round(v['valu_lane_util'],2)) for k,v in (r['kernels'] or {}).items()})"
tail -n 3 $OUT/bench_default.err; tail -n 3 $OUT/bench_living.err
