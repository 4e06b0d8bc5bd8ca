#!/bin/bash
# A/B of library builds inside one GPU session: tools/gpu_ab_builds.sh "<bench args>" variants/a.so variants/b.so ...
# (alternating, 3 rounds; REDNER_AMD_LIB selects the build, redner_amd/_capi.py)
ARGS=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    REDNER_AMD_LIB=$GRAFT_REPO_ROOT/$lib python bench.py $ARGS --no-cpu-baseline --no-profile --no-alone-leg 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), 'Msamples/s  frac', round(d['roofline']['frac'],3))"
  done
done
