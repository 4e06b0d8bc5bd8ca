#!/bin/bash
# Round 6, call 2: the split hierarchical pick + sorted refill: parity subset on both builds, A/B grid, per-kernel time.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_b; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_tuning.py tests/test_backward_parity.py tests/test_config_parity.py tests/test_raytri.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest_subset.log
tail -4 $OUT/pytest_subset.log
tools/gpu_r6_exp.sh "1x8 one-launch pick|RDR_WORKERS=1 RDR_PICKH_ONE_LAUNCH=1" "2x4 one-launch pick|RDR_PICKH_ONE_LAUNCH=1" "2x4 split pick|X=0" "2x4 split pick sort1|RDR_REFILL_SORT=1" "2x4 split k8|RDR_PICKH_REFILL=8,16,8" "2x4 split idle32|RDR_PICKH_REFILL=4,32,8" "2x4 split idle8 steps4|RDR_PICKH_REFILL=4,8,4" "1x8 split pick|RDR_WORKERS=1"
export TMPDIR=/tmp
for v in one split; do
  if [ $v = one ]; then export RDR_PICKH_ONE_LAUNCH=1; else unset RDR_PICKH_ONE_LAUNCH; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg > /dev/null 2>&1)
  f=$(ls $OUT/prof_$v/*/*kernel_stats.csv | head -1)
  python - "$f" $v <<'PY' | tee $OUT/kernel_stats_$v.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(sys.argv[2], 'total kernel ms', tot / 1e6)
for r in rows[:16]:
    print('%-90s calls %5s total %9.2f ms avg %8.3f ms %5.1f %%' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
PY
  cp $f $OUT/kernel_stats_$v.csv
  rm -rf $OUT/prof_$v
done
