#!/bin/bash
# rocprofv3 per-kernel statistics of a short benchmark run -> gpurun_out/$1/kernel_stats.csv (top kernels printed)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-kstats}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --no-alone-leg --no-profile ${@:2}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/stats
python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print('%-70s calls %6s  avg %9.1f us  total %8.2f ms  %5.1f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
PY
