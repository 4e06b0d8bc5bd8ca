#!/bin/bash
# Round 6, call 11: AdjBounceScatter over lanes ordered by the kind of shape involved -- a partitioned list (RDR_SCATTER_SORT=1/2) or
# each workgroup's 256 items regrouped in LDS (exec::stage_kernel_grouped, RDR_SCATTER_SORT=11..14): per-kernel time alone, then throughput.
LIB=${1:-variants/grp.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_k; mkdir -p $OUT
export REDNER_AMD_LIB=$GRAFT_REPO_ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 16 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg"
for m in 0 1 2 11 12 13 14; do
  RDR_SCATTER_SORT=$m RDR_NO_OVERLAP=1 RDR_WORKERS=1 RDR_BATCH=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$m -- $P > $OUT/st_$m.log 2>&1
  echo "== RDR_SCATTER_SORT=$m"
  python - $OUT/st_$m <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if 'AdjBounceScatter' in n or 'AdjBounceNee' in n or 'KeepNextKind' in n or 'BounceContrib' in n:
            print('  %-70s calls %4s avg %9.1f us total %8.2f ms' % (n.replace('void exec::', '').replace('rdr::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  rm -rf $OUT/st_$m
done 2>&1 | tee $OUT/scatter_kernel_ab.txt
cd $GRAFT_REPO_ROOT
tools/gpu_r6_exp.sh "base|X=0" "grouped next|RDR_SCATTER_SORT=11" "grouped own|RDR_SCATTER_SORT=12" "grouped own,next|RDR_SCATTER_SORT=13" "grouped next,own|RDR_SCATTER_SORT=14" "list by next|RDR_SCATTER_SORT=1"
