#!/bin/bash
# Round 3, first GPU call: the new parity tests + baseline numbers before the kernel work.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3_call1
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 900 python -m pytest tests -m gpu -q -k "not config_size" 2>&1 | tail -15 > $OUT/pytest.log
unset RDR_PARITY_REPORT
cat $OUT/pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2> $OUT/bench.err | tail -1 > $OUT/bench.json
cut -c1-400 $OUT/bench.json; echo
python tools/small_loop_timing.py 256 4 > $OUT/small_loop.log 2>&1; tail -4 $OUT/small_loop.log
