#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_call3; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_default_library_gpu.py tests/test_libm_exact.py tests/test_config_parity.py tests/test_edge_cases.py tests/test_distributed.py -m gpu -q -x --durations=5 2>&1 | tail -15 | tee $OUT/pytest.log
for rep in 1 2; do for m in "" exact; do
  REDNER_AMD_LIBM=$m python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-self-check --no-profile --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('libm=[$m]', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1))"
done; done 2>&1 | tee $OUT/ab_libm.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
