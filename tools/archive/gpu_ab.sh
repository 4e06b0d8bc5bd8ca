#!/bin/bash
# A/B harness on the GPU box: tools/gpu_ab.sh <outdir> "<name>:<ENV=..>[,ENV=..]" ...   (per variant: short bench + rocprofv3 kernel stats)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $OUT/pytest.log; tail -2 $OUT/pytest.log
fi
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg"
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --no-alone-leg"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; E=$(echo "${spec#*:}" | tr ',' ' '); [ -z "$E" ] && E="X=1"
  env $E timeout 300 $B 2> $OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  python -c "import json,sys; d=json.load(open('$OUT/bench_$v.json')); print('$v', round(d['value'],2), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step')"
  if [ -z "$SKIP_PROF" ]; then
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$v -- $P > $OUT/stats_$v.log 2>&1
  cp $OUT/stats_$v/*/*_kernel_stats.csv $OUT/kernel_stats_$v.csv
  rm -rf $OUT/stats_$v
  python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats_$v.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:${TOPN:-12}]:
    n=r['Name'].replace('void exec::stage_kernel<rdr::','').replace('rdr::','')[:48]
    print('   %-48s calls %4s avg %8.3f max %8.3f ms %5.1f%%' % (n, r['Calls'], float(r['AverageNs'])/1e6, float(r['MaxNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
print('   total kernel ms', round(tot/1e6,1))
PY
  fi
done
cd $GRAFT_REPO_ROOT
if [ -n "$SMALL_LOOP" ]; then python tools/small_loop_timing.py 256 4 2>&1 | tail -4 | tee $OUT/small_loop.log; fi
