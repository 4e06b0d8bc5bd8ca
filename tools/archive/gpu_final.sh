# Round-end measurements on the GPU box; tools/collect_profiles.py turns gpurun_out/final/ into profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --no-alone-leg"
S="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp 2 --no-cpu-baseline --no-alone-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $S > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $S > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq -- $S > $OUT/sq.log 2>&1
rm -f $OUT/*/*/*_kernel_trace.csv        # large; the summaries are what is kept
cat $OUT/bench_default.json | cut -c1-400
