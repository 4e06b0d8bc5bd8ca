#!/bin/bash
# Round 6, call 5: persistent-wave traversal (chunks off a counter) against the chunk-per-wave refill kernel.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_e; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/trace_ab.py 2048 -- "" "RDR_TRACE_PERSIST=1" "RDR_TRACE_PERSIST=1 RDR_REFILL_SORT=1" "RDR_REFILL_SORT=1" "RDR_TRACE_PERSIST=2" "RDR_TRACE_PERSIST=1 RDR_TRACE_REFILL=4,16,4" "RDR_TRACE_PERSIST=1 RDR_TRACE_REFILL=4,32,4" "RDR_TRACE_PERSIST=1 RDR_TRACE_REFILL=4,24,8" 2>&1 | tee $OUT/trace_persist_ab.txt | tail -20
timeout 900 python -m pytest tests/test_tuning.py tests/test_backward_parity.py tests/test_config_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
tools/gpu_r6_exp.sh "2x4 refill no-nee-compact|RDR_NO_NEE_COMPACT=1" "2x4 refill|X=0" "2x4 refill sort1|RDR_REFILL_SORT=1" "2x4 persist|RDR_TRACE_PERSIST=1" "2x4 persist sort1|RDR_TRACE_PERSIST=1 RDR_REFILL_SORT=1" "1x8 persist sort1|RDR_WORKERS=1 RDR_TRACE_PERSIST=1 RDR_REFILL_SORT=1" "1x8 refill|RDR_WORKERS=1"
for spec in "default|X=0" "persist|RDR_TRACE_PERSIST=1" "persist_sort|RDR_TRACE_PERSIST=1 RDR_REFILL_SORT=1"; do
  label=${spec%%|*}; envs=${spec#*|}
  for lv in 0 4; do echo -n "$label " ; env $envs timeout 600 python tools/large_scene_trace.py $lv 8 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['triangles'], 'tris  frac %.3f  %.2f Grays/s  mean launch %.3f ms  fwd %.1f ms' % (d['frac'], d['rays_per_s'] / 1e9, d['mean_launch_ms'], d['forward_ms']))"; done
done | tee $OUT/large_scene_persist.txt
