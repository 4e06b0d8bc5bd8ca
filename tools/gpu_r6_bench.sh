#!/bin/bash
# Round 6: the default bench line (as the driver runs it) + the rocprofv3 kernel statistics of the same command (short form).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_h; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r = d['roofline']
print('value %.2f Msamples/s  ms/step %.1f  frac %.3f (per launch %.3f, alone %s)  traffic %s' % (d['value'], d['ms_per_step'], r['frac'], r['per_launch']['frac'], r['alone'] and '%.3f' % r['alone']['frac'], r['traffic']))
print('cpu_baseline', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
print('gpu_vs_reference', json.dumps(d.get('cpu_baseline', {}).get('gpu_vs_reference')))
print('self_check', d.get('self_check'))
print('roofline_large', json.dumps(d.get('roofline_large')))
PY
export TMPDIR=/tmp
for v in over alone; do
  if [ $v = alone ]; then export RDR_NO_OVERLAP=1; else unset RDR_NO_OVERLAP; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg --no-large-leg > /dev/null 2>&1)
  cp $(ls $OUT/prof_$v/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$v.csv
  rm -rf $OUT/prof_$v
done
head -12 $OUT/kernel_stats_over.csv | cut -c1-160
