#!/bin/bash
# Round 6, call 19: where a small optimisation-loop iteration (256 x 256 x 4 spp) spends its time: kernel trace of the loop's last
# iterations (per-kernel totals, GPU-busy time against wall time).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "256 4" "512 4"; do
  rm -rf $OUT/tr
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/small_loop_timing.py $cfg > $OUT/loop.log 2>&1
  grep iteration $OUT/loop.log
  echo "== $cfg: last 45 ms of the trace"
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/tr 45
done 2>&1 | tee $OUT/small_loop_timeline.txt
rm -rf $OUT/tr
