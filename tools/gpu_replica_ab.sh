#!/bin/bash
# per-kernel time of the adjoint stages on the config-5 stand-ins under different large-tier replica budgets:
#   bash tools/gpu_replica_ab.sh "X=0" "RDR_REPLICA_MB=1024"
OUT=$GRAFT_REPO_ROOT/gpurun_out/replica; rm -rf $OUT; mkdir -p $OUT
i=0
for v in "$@"; do
  for w in ${AB_WORKLOADS:-living_room_standin}; do
  i=$((i+1))
  cd /tmp; export TMPDIR=/tmp
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -- python $GRAFT_REPO_ROOT/bench.py --workload $w --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-alone-leg --no-self-check > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== $v $w"
  python tools/trace_timeline.py $OUT/t$i | grep -i "launches in\|AdjBounce\|AdjPrimary\|PrimaryEdgeDer\|SecondaryEdgeDer\|BounceContrib\|FlushGrad\|fill"
  find $OUT -name "*.csv" -size +2M -delete
  done
done
