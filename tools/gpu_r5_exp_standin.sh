#!/bin/bash
# config-5 stand-ins: two builds against each other (bench.py --workload ..., 64 spp, 2 steps)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp; mkdir -p $OUT
for w in living_room_standin living_room_standin_envmap; do for v in "$GRAFT_REPO_ROOT/variants/$1" ""; do
  REDNER_AMD_LIB=$v python bench.py --workload $w --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-30s lib=[%s] %6.2f Msamples/s' % ('$w', '$v'.split('/')[-1], d['value']))"
done; done 2>&1 | tee -a $OUT/exp_standin.txt
