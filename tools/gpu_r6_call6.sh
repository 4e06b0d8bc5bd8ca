#!/bin/bash
# Round 6, call 6: next-event adjoint over the compacted list of lanes that have something to differentiate.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_f; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tuning.py tests/test_backward_parity.py tests/test_config_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
tools/gpu_r6_exp.sh "2x4 no-nee-compact|RDR_NO_NEE_COMPACT=1" "2x4 nee-compact|X=0" "2x4 nee-compact sort1|RDR_REFILL_SORT=1" "1x8 nee-compact|RDR_WORKERS=1" "2x8 nee-compact sort1 32GiB|RDR_REFILL_SORT=1 RDR_POOL_CAP_MB=32768"
