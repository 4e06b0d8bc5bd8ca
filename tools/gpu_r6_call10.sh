#!/bin/bash
# Round 6, call 10: the size-gated stage forms: small-frame loop (default = small-frame forms) against the large-frame forms, whole suite.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_j; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for spec in "default (small-frame forms)|X=0" "large-frame forms|RDR_LARGE_FRAME_FORMS=1"; do
    label=${spec%%|*}; envs=${spec#*|}
    echo -n "$label: "; env $envs python tools/small_loop_timing.py 256 4 2>&1 | grep iteration
  done
done | tee $OUT/small_loop_ab.txt
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
python bench.py --spp 128 --steps 2 --no-cpu-baseline --no-profile --no-self-check --no-large-leg --no-alone-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['config']['sample_workers'], d['config']['samples_per_launch'])"
