#!/bin/bash
# Round 6, call 21: slots of the next-event-mode pick that the gather marks for the reference-order walk, before and after the walk
# (variants/gcount2.so prints the gather book's counters), with the lists sized by the launch set and with the old fixed sizes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_u; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/gcount2.so
for spec in "lists sized by the launch set|X=0" "old sizes (8192 heavy slots, 131072 work items)|RDR_GATHER_CAPS=8192,131072"; do
  label=${spec%%|*}; envs=${spec#*|}
  echo "== $label"
  env $envs python bench.py --spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg 2>&1 | grep "\[gather\]" | sort | uniq -c | sort -rn | head -12
done 2>&1 | tee $OUT/gather_marks.txt
