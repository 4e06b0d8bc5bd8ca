#!/bin/bash
# Round 4, final GPU call: the whole GPU suite on the final tree (branred restated; flip budget 0), then the per-kernel times.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_libm2
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 230 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -120 > $OUT/pytest.log
unset RDR_PARITY_REPORT
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | tail -12
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check"
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/stats
head -8 $OUT/kernel_stats.csv | cut -c1-160
