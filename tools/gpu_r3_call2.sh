#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3_call2
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "not config_size" 2>&1 | tail -5
for v in "RDR_BATCH=1" "RDR_X=1" "RDR_BATCH=2" "RDR_WORKERS=1" "RDR_BATCH=2 RDR_WORKERS=2"; do
  echo "== [$v]"; env $v python tools/small_loop_timing.py 256 4 2>&1 | tail -4
done
echo "== 256 x 16 spp"; for v in "RDR_BATCH=1" "RDR_X=1" "RDR_BATCH=4" "RDR_BATCH=8 RDR_WORKERS=2"; do echo "[$v]"; env $v python tools/small_loop_timing.py 256 16 2>&1 | tail -2; done
echo "== 128 x 4 spp"; for v in "RDR_BATCH=1" "RDR_X=1"; do echo "[$v]"; env $v python tools/small_loop_timing.py 128 4 2>&1 | tail -2; done
