#!/bin/bash
# Round 6, call 21b: as call 21, in 8-sample launch sets (the benchmark's), one and two sample workers.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_u; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/gcount2.so
for spec in "new sizes, one worker|RDR_BATCH=8 RDR_WORKERS=1" "old sizes, one worker|RDR_BATCH=8 RDR_WORKERS=1 RDR_GATHER_CAPS=8192,131072" "old sizes, two workers|RDR_BATCH=8 RDR_GATHER_CAPS=8192,131072"; do
  label=${spec%%|*}; envs=${spec#*|}
  echo "== $label"
  env $envs python bench.py --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg 2>&1 | grep "\[gather\]" | sort | uniq -c | sort -rn | head -12
done 2>&1 | tee $OUT/gather_marks_8.txt
