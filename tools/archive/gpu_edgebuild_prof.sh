#!/bin/bash
# rocprofv3 kernel statistics of the optimisation loop (tools/small_loop_timing.py): what the device-side edge hierarchy build costs
OUT=$GRAFT_REPO_ROOT/gpurun_out/edgebuild
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/tools/small_loop_timing.py > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/stats
python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats.csv')))
keys = ('edge_bounds', 'scene_bounds', 'codes_kernel', 'radix', 'init_nodes', 'bounds_up', 'treelet', 'leaf_rank', 'fatten', 'gather_leaf', 'reset_counters', 'single_leaf', 'rocprim', 'onesweep', 'histogram', 'sort')
for r in rows:
    if any(k in r['Name'] for k in keys):
        print('%-90s calls %5s avg %9.1f us total %8.2f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
