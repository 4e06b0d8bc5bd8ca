#!/bin/bash
# The whole GPU suite on the round's final tree, in one run (the record behind profiles/r4_pytest_gpu_final.log).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_final
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 170 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -30 > $OUT/pytest.log
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | tail -8
