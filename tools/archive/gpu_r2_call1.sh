#!/bin/bash
# round 2, call 1: full GPU test suite with the parity report, then the default bench
mkdir -p gpurun_out/r2c1
export RDR_PARITY_REPORT=$PWD/gpurun_out/r2c1/parity.jsonl
rm -f $RDR_PARITY_REPORT
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -150 > gpurun_out/r2c1/pytest.log
unset RDR_PARITY_REPORT
timeout 600 python bench.py > gpurun_out/r2c1/bench.json 2> gpurun_out/r2c1/bench.err
tail -5 gpurun_out/r2c1/pytest.log
cat gpurun_out/r2c1/bench.json
