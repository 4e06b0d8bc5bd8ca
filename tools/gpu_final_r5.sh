#!/bin/bash
# Round-5 measurements on the GPU box -> gpurun_out/final_r5/ (the files copied into profiles/r5_* come from here).
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r5
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 1700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 > $OUT/pytest.log
unset RDR_PARITY_REPORT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 1200 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json ) 2> $OUT/bench_default.time
REDNER_AMD_LIBM=exact timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg 2> /dev/null | tail -1 > $OUT/bench_exact_libm.json
timeout 600 python bench.py --workload living_room_standin --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check 2> /dev/null | tail -1 > $OUT/bench_living_room_standin.json
timeout 600 python bench.py --workload living_room_standin_envmap --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check 2> /dev/null | tail -1 > $OUT/bench_living_room_standin_envmap.json
RDR_POOL_CAP_MB=65536 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-profile --no-self-check 2> /dev/null | tail -1 > $OUT/bench_pool_cap_64g.json
{ for cfg in "256 4" "256 4 move" "256 16" "128 8" "512 4"; do echo "== $cfg"; python tools/small_loop_timing.py $cfg 2>&1 | tail -4; done; } > $OUT/small_loop.txt
# the multi-rank path of bench.py started WITHOUT a launcher, two ranks on the ONE GPU of this box (gloo; not a measurement)
RDR_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --spp 32 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg 2> $OUT/bench_two_ranks_shared_gpu.err | tail -1 > $OUT/bench_two_ranks_shared_gpu.json
python tools/scene_build_timing.py 2>&1 | tail -3 > $OUT/scene_build.txt
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
RDR_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_alone -- $P > $OUT/stats_alone.log 2>&1
cp $OUT/stats_alone/*/*_kernel_stats.csv $OUT/kernel_stats_alone.csv
rm -rf $OUT/stats $OUT/stats_alone
cd $GRAFT_REPO_ROOT
tail -12 $OUT/pytest.log; cat $OUT/bench_default.time | tail -3; cut -c1-260 $OUT/bench_default.json; echo
for f in exact_libm living_room_standin living_room_standin_envmap pool_cap_64g two_ranks_shared_gpu; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read()); print('$f', round(d['value'],2), 'roofline', round(d['roofline']['frac'],3))"; done
grep -E "==|iteration" $OUT/small_loop.txt; cat $OUT/scene_build.txt
