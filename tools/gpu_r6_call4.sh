#!/bin/bash
# Round 6, call 4: counters of the pick kernels (bench.py's own rocprofv3 passes) under three pick forms; the large hierarchy
# under the 4-wide records / without refill, and its HBM counters.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_d; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for spec in "one|RDR_PICKH_ONE_LAUNCH=1" "split_k4|X=0" "split_k1|RDR_PICKH_REFILL=1,8,8"; do
  label=${spec%%|*}; envs=${spec#*|}
  env $envs python bench.py --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-self-check 2>/dev/null | tail -1 > $OUT/bench_$label.json
  python - $OUT/bench_$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d['roofline']['kernels'] or {}
print(sys.argv[2], '%.2f Msamples/s' % d['value'], 'frac %.3f alone %.3f' % (d['roofline']['frac'], (d['roofline']['alone'] or {}).get('frac', 0)))
for n in ('SecEdgePickH', 'SecEdgePickHDescend', 'SecEdgePickHLeaves', 'SecEdgeSetup', 'SecEdgeGatherN', 'trace_closest_refill', 'trace_any_refill'):
    if n in k:
        v = k[n]
        print('   %-22s alone %.3f ms overlapped %s launches/sample %.3f lane util %.3f valu frac %.3f waiting %.2f hbm frac %.3f' % (
            n, v['mean_launch_ms_alone'], v['mean_launch_ms_overlapped'] and '%.3f' % v['mean_launch_ms_overlapped'], v['launches_per_sample'], v['valu_lane_util'], v['valu_frac_of_peak'] or 0,
            v['wave_cycles_waiting_frac'] or 0, v['hbm_frac_of_peak'] or 0))
PY
done 2>&1 | tee $OUT/pick_counters.txt
for spec in "default|X=0" "wide|RDR_WIDE_MAX=2000000000" "norefill|RDR_TRACE_REFILL=0" "refill8|RDR_TRACE_REFILL=8,24,4"; do
  label=${spec%%|*}; envs=${spec#*|}
  for lv in 3 4; do echo -n "$label " ; env $envs timeout 600 python tools/large_scene_trace.py $lv 8 2>&1 | tail -1; done
done | tee $OUT/large_scene_variants.txt
cd /tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python $GRAFT_REPO_ROOT/tools/large_scene_trace.py 4 8 > /dev/null 2>&1
  f=$(ls $OUT/pmc_$tag/*/*counter_collection.csv | head -1)
  python - "$f" <<'PY' | tee -a $OUT/large_scene_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'trace_' not in k: continue
    k = k[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k, v in agg.items():
    print(k, 'launches', len(n[k]), {c: x / len(n[k]) for c, x in v.items()})
PY
  rm -rf $OUT/pmc_$tag
done
