#!/bin/bash
# config-5 stand-ins: wave-level pre-sum rounds for texel gradients (RDR_TEXEL_ROUNDS), now that a round serves an rgb triple
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp; mkdir -p $OUT
for w in living_room_standin living_room_standin_envmap; do for r in 8 3 5 12 16; do
  RDR_TEXEL_ROUNDS=$r python bench.py --workload $w --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-30s rounds %2d %6.2f Msamples/s' % ('$w', $r, d['value']))"
done; done 2>&1 | tee -a $OUT/exp_texel_rounds.txt
