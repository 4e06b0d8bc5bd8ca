#!/bin/bash
# A/B of the host thread pool's size / placement on the Scene build (edge structures) and the 256x256 loop
mkdir -p /tmp/dd
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node|L3" | head -12
for cfg in "X=1" "RDR_POOL_THREADS=15" "RDR_POOL_THREADS=7" "RDR_POOL_THREADS=15 RDR_POOL_PIN=16" "RDR_POOL_THREADS=7 RDR_POOL_PIN=8" "RDR_POOL_THREADS=31 RDR_POOL_PIN=32" "RDR_POOL_THREADS=15 RDR_POOL_PIN=8"; do
  echo "== $cfg"
  env $cfg RDR_DEBUG_DUMP=/tmp/dd python tools/scene_build_phases.py 2>&1 | grep "scene build: edge structures" | tail -2
  env $cfg python tools/small_loop_timing.py 2>&1 | tail -1
done
