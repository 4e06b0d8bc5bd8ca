#!/bin/bash
# Round 3, after the batch cap went to 2^24 lanes: the default benchmark line, the kernel statistics of the same job, the GPU tests.
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r3; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 900 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
cut -c1-200 $OUT/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 32 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
RDR_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_alone -- $P > $OUT/stats_alone.log 2>&1
cp $OUT/stats_alone/*/*_kernel_stats.csv $OUT/kernel_stats_alone.csv
rm -rf $OUT/stats $OUT/stats_alone
cd $GRAFT_REPO_ROOT
python tools/small_loop_timing.py 256 4 2>&1 | tail -1
head -8 $OUT/kernel_stats_alone.csv | cut -c1-160
