#!/bin/bash
# Round 6, call 20: the heavy-slot lists of the next-event-mode pick sized by the launch set.  (1) a build that prints the gather
# book's counters (variants/gcount.so, the caps as before) on a short job; (2) fixtures on both builds; (3) the previous library
# against the in-tree one, per-kernel time alone.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_t; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/gcount.so RDR_WORKERS=1 python bench.py --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg 2>&1 | grep "\[gather\]" | sort | uniq -c | head -8 | tee $OUT/gather_counts.txt
timeout 1500 python -m pytest tests/test_config_parity.py tests/test_backward_parity.py tests/test_default_library_gpu.py tests/test_sample_batches.py tests/test_fuzz_parity.py tests/test_tuning.py tests/test_edge_cases.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_subset.log
tools/gpu_r6_exp.sh "previous commit|REDNER_AMD_LIB=variants/prev.so" "+ heavy-slot lists sized by the launch set|X=0"
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg"
for lib in variants/prev.so redner_amd/lib/libredner_amd.so; do
  REDNER_AMD_LIB=$GRAFT_REPO_ROOT/$lib RDR_NO_OVERLAP=1 RDR_WORKERS=1 RDR_BATCH=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $P > $OUT/st.log 2>&1
  echo "== $lib"
  python - $OUT/st <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if 'SecEdgeGather' in n or 'SecEdgePickNWalk' in n:
            print('  %-70s calls %4s avg %9.1f us total %8.2f ms  min %8.1f max %8.1f us' % (n.replace('void exec::', '').replace('rdr::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  rm -rf $OUT/st
done 2>&1 | tee $OUT/gather_caps_ab.txt
