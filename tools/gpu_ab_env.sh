#!/bin/bash
# bench.py (short job) and the small-frame loop under different environment settings: bash tools/gpu_ab_env.sh "A=1" "B=2 C=3" ...
for v in "$@"; do
  echo "== [$v]"
  env $v timeout 300 python bench.py --spp ${AB_SPP:-32} --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  %.2f Msamples/s  closest %.4f ms/launch (%.0f k rays, frac %.3f)  traversal share %.2f' % (d['value'], r['mean_launch_ms'], r['rays_per_launch']/1e3, r['frac'], r['traversal_share_of_step']))"
  env $v timeout 120 python tools/small_loop_timing.py 256 4 2>/dev/null | tail -1
done
