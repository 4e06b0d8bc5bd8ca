#!/bin/bash
# config-5 stand-ins: throughput + per-kernel time (kernel trace of a short job)
OUT=$GRAFT_REPO_ROOT/gpurun_out/standin; rm -rf $OUT; mkdir -p $OUT
for w in living_room_standin living_room_standin_envmap; do
  timeout 300 python bench.py --workload $w --spp 32 --steps 2 --no-cpu-baseline --no-profile --no-alone-leg --no-self-check 2>/dev/null | tail -1 > $OUT/$w.json
  python -c "
import json; d=json.loads(open('$OUT/$w.json').read()); print('$w', round(d['value'],2), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step')"
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-alone-leg --no-self-check > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/trace_timeline.py $OUT/trace_$w | head -24
  find $OUT -name "*.csv" -size +2M -delete
done
