#!/bin/bash
# Round 5, call 1: the new GPU tests (unmodified pyredner on libredner_amd.so, bench.py self-launch, empty scenes) + a baseline bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_call1
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_dropin_pyredner_gpu.py tests/test_distributed.py tests/test_edge_cases.py tests/test_tuning.py -m gpu -q -x --durations=5 2>&1 | tail -30 > $OUT/pytest_new.log
cat $OUT/pytest_new.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-self-check 2> $OUT/bench.err | tail -1 > $OUT/bench_baseline.json
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_call1/bench_baseline.json').read())
print('value',d['value'],'frac',d['roofline']['frac'])
for k,v in (d['roofline'].get('kernels') or {}).items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})
PY
