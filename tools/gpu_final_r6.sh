#!/bin/bash
# Round-6 measurements on the GPU box -> gpurun_out/final_r6/ (the files copied into profiles/r6_* come from here).
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r6
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
export RDR_RCCL_LOG=$OUT/rccl
timeout 1700 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider 2>&1 | tail -25 > $OUT/pytest.log
unset RDR_PARITY_REPORT RDR_RCCL_LOG
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json ) 2> $OUT/bench_default.time
timeout 600 python bench.py --workload living_room_standin --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check 2> /dev/null | tail -1 > $OUT/bench_living_room_standin.json
timeout 600 python bench.py --workload living_room_standin_envmap --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check 2> /dev/null | tail -1 > $OUT/bench_living_room_standin_envmap.json
RDR_POOL_CAP_MB=8192 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-profile --no-self-check --no-large-leg 2> /dev/null | tail -1 > $OUT/bench_pool_cap_8g.json
RDR_POOL_CAP_MB=16384 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-profile --no-self-check --no-large-leg 2> /dev/null | tail -1 > $OUT/bench_pool_cap_16g.json
{ for cfg in "256 4" "256 4 move" "256 16" "128 8" "512 4"; do echo "== $cfg"; python tools/small_loop_timing.py $cfg 2>&1 | tail -4; done; } > $OUT/small_loop.txt
{ echo "== 256 4 RDR_NO_NEE_COMPACT=1 RDR_PICKH_ONE_LAUNCH=1"; RDR_NO_NEE_COMPACT=1 RDR_PICKH_ONE_LAUNCH=1 python tools/small_loop_timing.py 256 4 2>&1 | tail -4; } >> $OUT/small_loop.txt
python tools/scene_build_timing.py 2>&1 | tail -3 > $OUT/scene_build.txt
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 32 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
RDR_NO_OVERLAP=1 RDR_WORKERS=1 RDR_BATCH=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_alone -- $P > $OUT/stats_alone.log 2>&1
cp $OUT/stats_alone/*/*_kernel_stats.csv $OUT/kernel_stats_alone.csv
rm -rf $OUT/stats $OUT/stats_alone
cd $GRAFT_REPO_ROOT
tail -12 $OUT/pytest.log; cat $OUT/bench_default.time | tail -3
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['roofline']
print('value %.2f Msamples/s  ms/step %.1f  frac %.3f (per launch %.3f, alone %s)  traffic %s  schedule %sx%s cap %s' % (d['value'], d['ms_per_step'], r['frac'], r['per_launch']['frac'], r['alone'] and '%.3f' % r['alone']['frac'], r['traffic'], d['config']['sample_workers'], d['config']['samples_per_launch'], d['config']['pool_cap_mb']))
print('cpu_baseline', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
print('gpu_vs_reference', json.dumps(d.get('cpu_baseline', {}).get('gpu_vs_reference')))
print('self_check', d.get('self_check'))
big = d.get('roofline_large') or {}
print('roofline_large frac %s rays/s %s counters %s' % (big.get('frac'), big.get('rays_per_s'), json.dumps(big.get('counters'))))
k = r.get('kernels') or {}
for n, v in sorted(k.items(), key=lambda kv: -kv[1]['launches_per_sample'] * kv[1]['mean_launch_ms_alone']):
    print('   %-22s %.3f x %.3f ms | lane util %.2f valu %.2f waiting %.2f hbm %.2f' % (n, v['launches_per_sample'], v['mean_launch_ms_alone'], v['valu_lane_util'], v['valu_frac_of_peak'] or 0, v['wave_cycles_waiting_frac'] or 0, v['hbm_frac_of_peak'] or 0))
PY
for f in living_room_standin living_room_standin_envmap pool_cap_8g pool_cap_16g; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read()); print('$f', round(d['value'],2), 'roofline', round(d['roofline']['frac'],3), d['config'].get('sample_workers'), d['config'].get('samples_per_launch'))"; done
grep -E "==|iteration" $OUT/small_loop.txt; cat $OUT/scene_build.txt
