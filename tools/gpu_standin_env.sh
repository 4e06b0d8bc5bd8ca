#!/bin/bash
# throughput of the config-5 stand-ins under different environment settings: bash tools/gpu_standin_env.sh "A=1" "B=2" ...
for v in "$@"; do
  for w in living_room_standin living_room_standin_envmap; do
    env $v timeout 300 python bench.py --workload $w --spp 32 --steps 2 --no-cpu-baseline --no-profile --no-alone-leg --no-self-check 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] $w', round(d['value'],2), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step')"
  done
done
