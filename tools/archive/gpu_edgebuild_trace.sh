#!/bin/bash
# per-dispatch durations of the treelet pass (one Scene build): rocprofv3 --kernel-trace of tools/scene_build_phases.py
OUT=$GRAFT_REPO_ROOT/gpurun_out/edgetrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RDR_SYNC_EDGES=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/scene_build_phases.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/tr/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tl = [r for r in rows if 'treelet_level' in r['Kernel_Name']]
per = len(tl) // 4
last = tl[-per:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    print('grid %7s  start %8.1f us  dur %7.1f us' % (r.get('Grid_Size', r.get('Grid_Size_X', '?')), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
