#!/bin/bash
# round 2, call 2: GPU tests with the new edge-pick kernels + A/B benches + per-kernel stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c2
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity.jsonl
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.log
unset RDR_PARITY_REPORT
tail -3 $OUT/pytest.log
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg"
for v in default walk fused both; do
  case $v in
    default) E="";;
    walk) E="RDR_PICKN_WALK=1";;
    fused) E="RDR_PICKH_FUSED=1";;
    both) E="RDR_PICKN_WALK=1 RDR_PICKH_FUSED=1";;
  esac
  env $E timeout 300 $B 2> $OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  python -c "import json,sys; d=json.load(open('$OUT/bench_$v.json')); print('$v', round(d['value'],2), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step')"
done
RDR_DEBUG_DUMP=/tmp/dump_unused python tools/scene_build_timing.py 2>&1 | grep -v "^\[redner_amd\] render" | tail -40 > $OUT/scene_build.log
python tools/small_loop_timing.py 256 4 > $OUT/small_loop.log 2>&1; cat $OUT/small_loop.log
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --no-alone-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/stats
head -25 $OUT/kernel_stats.csv | cut -c1-150
