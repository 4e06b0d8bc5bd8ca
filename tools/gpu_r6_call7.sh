#!/bin/bash
# Round 6, call 7: whole GPU suite on both builds (parity report), then the settings grid on the final tree.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_g; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
RDR_PARITY_REPORT=$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $OUT/pytest.log; tail -6 $OUT/pytest.log
tools/gpu_r6_exp.sh "default (2 workers, 16 GiB, octant order)|X=0" "queue order|RDR_REFILL_SORT=0" "32 GiB|RDR_POOL_CAP_MB=32768" "64 GiB|RDR_POOL_CAP_MB=65536" "1 worker|RDR_WORKERS=1" "r5 schedule+kernels|RDR_WORKERS=1 RDR_REFILL_SORT=0 RDR_PICKH_ONE_LAUNCH=1 RDR_NO_NEE_COMPACT=1"
