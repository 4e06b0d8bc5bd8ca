#!/bin/bash
# A/B of two builds of the library in one GPU session: bench.py (short), alternating.  usage: gpu_r5_ab.sh <variant.so> [bench args]
V=$GRAFT_REPO_ROOT/variants/$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_ab; mkdir -p $OUT
for rep in 1 2; do for v in "$V" ""; do
  REDNER_AMD_LIB=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=[%s]' % '$v'.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'closest ms', round(d['roofline']['mean_launch_ms'],4))"
done; done 2>&1 | tee -a $OUT/ab.txt
