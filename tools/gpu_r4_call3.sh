#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call3
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --spp 32 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg --workload living_room_standin"
show() { python -c "
import json,sys
d=json.loads(open('$1').read()); r=d['roofline']
print('$1'.split('/')[-1], '%.2f Msamples/s  %.0f ms/step  closest %.3f ms/launch frac %.3f' % (d['value'], d['ms_per_step'], r['mean_launch_ms'], r['frac']))"; }
for v in "RDR_POOL_CAP_MB=200000" "RDR_POOL_CAP_MB=200000 RDR_BATCH=8" "RDR_POOL_CAP_MB=200000 RDR_BATCH=2" "RDR_POOL_CAP_MB=200000 RDR_BATCH=1"; do
  n=$(echo $v | tr ' =' '__')
  env $v RDR_DEBUG_BATCH=1 timeout 300 $B 2>$OUT/err_$n.txt | tail -1 > $OUT/b_$n.json; echo "== $v"; show $OUT/b_$n.json; grep chain $OUT/err_$n.txt | head -2
done
cd /tmp && export TMPDIR=/tmp
RDR_POOL_CAP_MB=200000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 16 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg --workload living_room_standin > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats_chain16.csv; rm -rf $OUT/stats
head -25 $OUT/kernel_stats_chain16.csv | cut -c1-150
