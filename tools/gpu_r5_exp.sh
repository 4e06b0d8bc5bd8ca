#!/bin/bash
# Round 5: several builds / build parameters against the default, bench.py short runs in one GPU session.
# usage: gpu_r5_exp.sh "<label>|<env assignments>" ...     (REDNER_AMD_LIB=variants/x.so RDR_BVH_LEAF=2 ...)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp; mkdir -p $OUT
run() { env $2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s %6.2f Msamples/s  step %7.1f ms  closest %.4f ms  nodes/ray %.2f tris/ray %.2f' % ('$1', d['value'], d['ms_per_step'], r['mean_launch_ms'], r['nodes_per_ray'], r['tris_per_ray']))"; }
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    envs=${envs//variants\//$GRAFT_REPO_ROOT/variants/}
    run "$label" "$envs"
  done
done 2>&1 | tee -a $OUT/exp.txt
