#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_standin; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in living_room_standin living_room_standin_envmap; do
P="python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 1 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check"
RDR_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$w -- $P > $OUT/$w.log 2>&1
cp $OUT/s_$w/*/*_kernel_stats.csv $OUT/kernel_stats_alone_$w.csv; rm -rf $OUT/s_$w
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats_alone_$w.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$w total kernel ms', tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print('%5.1f%% %6d calls %9.3f ms avg  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e6, r['Name'][:90]))
PY
done
