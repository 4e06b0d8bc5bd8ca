#!/bin/bash
# Round 6, call 9: small frames under the round's new stages; register caps of the adjoint / walk kernels (variant builds).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_i; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for spec in "default|X=0" "no-nee-compact|RDR_NO_NEE_COMPACT=1" "one-launch-pick|RDR_PICKH_ONE_LAUNCH=1" "both-off|RDR_NO_NEE_COMPACT=1 RDR_PICKH_ONE_LAUNCH=1"; do
    label=${spec%%|*}; envs=${spec#*|}
    echo -n "$label: "; env $envs python tools/small_loop_timing.py 256 4 2>&1 | grep iteration
  done
done | tee $OUT/small_loop_ab.txt
tools/gpu_r6_exp.sh "base|X=0" "AdjBounceScatter 4 blocks|REDNER_AMD_LIB=variants/scat4.so" "AdjBounceNee 3 blocks|REDNER_AMD_LIB=variants/nee3.so" "chunked walks 4 blocks|REDNER_AMD_LIB=variants/chunk4.so" "pick lazy loads|RDR_PICKH_LAZY=1"
