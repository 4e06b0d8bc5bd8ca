#!/bin/bash
# Round-end measurements on the GPU box -> gpurun_out/final_r2/ ; tools/collect_profiles_r2.py copies them into profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r2
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest.log
unset RDR_PARITY_REPORT
timeout 900 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
timeout 600 python bench.py --workload living_room_standin --spp 32 --steps 2 --no-cpu-baseline 2> $OUT/bench_living.err | tail -1 > $OUT/bench_living_room_standin.json
timeout 600 python bench.py --workload living_room_standin_envmap --spp 32 --steps 2 --no-cpu-baseline 2> $OUT/bench_living_envmap.err | tail -1 > $OUT/bench_living_room_standin_envmap.json
timeout 900 python bench.py --workload living_room_standin --spp 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alone-leg 2> /dev/null | tail -1 > $OUT/bench_living_512spp.json
RDR_WORKERS=2 RDR_HELPER_LOW=1 timeout 600 python bench.py --no-cpu-baseline --no-profile 2> /dev/null | tail -1 > $OUT/bench_two_workers.json
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python tools/small_loop_timing.py 256 4 > $OUT/small_loop.log 2>&1
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --no-alone-leg --no-profile"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/stats
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh final_r2_pmc > $OUT/pmc.log 2>&1
cp gpurun_out/final_r2_pmc/kernel_stats_alone.csv gpurun_out/final_r2_pmc/pmc_sq.csv gpurun_out/final_r2_pmc/pmc_sq2.csv $OUT/
cat $OUT/pytest.log; cut -c1-300 $OUT/bench_default.json; echo; cut -c1-200 $OUT/bench_living_room_standin.json; echo; tail -4 $OUT/small_loop.log
