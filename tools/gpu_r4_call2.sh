#!/bin/bash
# Round 4, second GPU call: chain-mode batches (mip-mapped scenes with edge sampling) -- parity and throughput.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call2
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 > $OUT/pytest.log
unset RDR_PARITY_REPORT
tail -22 $OUT/pytest.log
B="python bench.py --steps 2 --warmup 1 --spp 32 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg"
show() { python -c "
import json,sys
d=json.loads(open('$1').read()); r=d['roofline']
print('$1'.split('/')[-1], '%.2f Msamples/s  %.0f ms/step  closest %.3f ms/launch frac %.3f' % (d['value'], d['ms_per_step'], r['mean_launch_ms'], r['frac']))"; }
for w in living_room_standin living_room_standin_envmap; do
  timeout 300 $B --workload $w 2>/dev/null | tail -1 > $OUT/bench_$w.json; show $OUT/bench_$w.json
  RDR_BATCH=1 timeout 300 $B --workload $w 2>/dev/null | tail -1 > $OUT/bench_${w}_unbatched.json; show $OUT/bench_${w}_unbatched.json
done
RDR_BATCH=4 timeout 300 $B --workload living_room_standin 2>/dev/null | tail -1 > $OUT/bench_living_room_standin_b4.json; show $OUT/bench_living_room_standin_b4.json
