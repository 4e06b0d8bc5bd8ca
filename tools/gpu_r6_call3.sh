#!/bin/bash
# Round 6, call 3: dynamic chunks for the descent, leaves as a walk, sorted refill; a hierarchy beyond the L2.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_tuning.py tests/test_backward_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
tools/gpu_r6_exp.sh "2x4 one-launch pick|RDR_PICKH_ONE_LAUNCH=1" "2x4 split dyn k4|X=0" "2x4 split dyn k4 sort1|RDR_REFILL_SORT=1" "2x4 split dyn k2|RDR_PICKH_REFILL=2,16,8" "2x4 split dyn k1 idle8|RDR_PICKH_REFILL=1,8,8" "2x4 split dyn k4 idle8|RDR_PICKH_REFILL=4,8,8" "2x4 split dyn k4 leaves-walk|RDR_PICKH_LEAVES_WALK=1" "2x4 split dyn k2 leaves-walk sort1|RDR_PICKH_LEAVES_WALK=1 RDR_PICKH_REFILL=2,16,8 RDR_REFILL_SORT=1"
for lv in 0 3 4; do timeout 600 python tools/large_scene_trace.py $lv 8 2>&1 | tail -1 | tee -a $OUT/large_scene.txt; done
