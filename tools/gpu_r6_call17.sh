#!/bin/bash
# Round 6, call 17: BounceContrib does not evaluate a next-event term nobody consumes (camera paths of a gradient render): fixtures on
# both builds, the previous commit's library (variants/prev.so) against the in-tree one, per-kernel time alone.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_q; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_config_parity.py tests/test_backward_parity.py tests/test_default_library_gpu.py tests/test_sample_batches.py tests/test_fuzz_parity.py tests/test_tuning.py tests/test_edge_cases.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_subset.log
tools/gpu_r6_exp.sh "previous commit|REDNER_AMD_LIB=variants/prev.so" "+ unconsumed next-event term not evaluated|X=0"
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
        if 'BounceContrib' in n:
            print('  %-70s calls %4s avg %9.1f us total %8.2f ms  min %8.1f max %8.1f us' % (n.replace('void exec::', '').replace('rdr::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  rm -rf $OUT/st
done 2>&1 | tee $OUT/unconsumed_nee_ab.txt
