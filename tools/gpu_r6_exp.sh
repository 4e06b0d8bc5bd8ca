#!/bin/bash
# Round 6: schedules / settings / builds against each other, short bench.py runs alternating in one GPU session.
# usage: gpu_r6_exp.sh [--spp N] "<label>|<env assignments>" ...
SPP=128; if [ "$1" = "--spp" ]; then SPP=$2; shift 2; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_exp; mkdir -p $OUT
run() { env $2 python bench.py --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s %6.2f Msamples/s  step %7.1f ms  closest: union frac %.3f per-launch frac %.3f mean %.4f ms overlap %.2f  cap %d MiB' % ('$1', d['value'], d['ms_per_step'], r['frac'], r['per_launch']['frac'], r['mean_launch_ms'], r['launch_overlap'], d['config']['pool_cap_mb']))"; }
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    envs=${envs//variants\//$GRAFT_REPO_ROOT/variants/}
    run "$label" "$envs"
  done
done 2>&1 | tee -a $OUT/exp.txt
