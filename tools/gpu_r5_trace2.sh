#!/bin/bash
# refill kernel at 8 waves per SIMD (64 VGPRs + 24 B scratch) vs shipped (72 VGPRs, 7 waves): tools/trace_ab.py, then bench
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_trace
mkdir -p $OUT
export RDR_TRACE_REFILL_ALL=1 TRACE_AB_TIMEOUT=300
{ python tools/trace_ab.py 2048 -- "" "REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/refill_w8.so" "" "REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/refill_w8.so"; } 2>&1 | tee $OUT/ab_w8.txt
unset RDR_TRACE_REFILL_ALL
for v in "" "$GRAFT_REPO_ROOT/variants/refill_w8.so" "" "$GRAFT_REPO_ROOT/variants/refill_w8.so"; do
  REDNER_AMD_LIB=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=[$v]', round(d['value'],2), 'closest ms', round(d['roofline']['mean_launch_ms'],4))"
done 2>&1 | tee -a $OUT/ab_w8.txt
