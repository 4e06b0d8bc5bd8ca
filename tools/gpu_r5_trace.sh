#!/bin/bash
# Round 5: traversal kernel A/B (tools/trace_ab.py) -- refilling kernel vs speculative leaves
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_trace
mkdir -p $OUT
export RDR_TRACE_REFILL_ALL=1 TRACE_AB_TIMEOUT=300
python tools/trace_ab.py 2048 -- "" "RDR_TRACE_SPEC=32,16" "RDR_TRACE_SPEC=24,12" "RDR_TRACE_SPEC=16,8" "RDR_TRACE_SPEC=48,24" "RDR_TRACE_SPEC=32,16 RDR_TRACE_REFILL=4,24,8" "RDR_TRACE_SPEC=32,16 RDR_TRACE_REFILL=8,24,4" "RDR_TRACE_SPEC=32,16 RDR_TRACE_REFILL=4,16,4" "RDR_TRACE_SPEC=1,1" 2>&1 | tee $OUT/ab1.txt
