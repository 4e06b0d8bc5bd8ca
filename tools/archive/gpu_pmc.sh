#!/bin/bash
# isolated (single-stream) per-kernel durations + SQ counters:  tools/gpu_pmc.sh <outdir>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RDR_NO_OVERLAP=1
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spp 4 --no-cpu-baseline --no-alone-leg"
S="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp 2 --no-cpu-baseline --no-alone-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats_alone.csv; rm -rf $OUT/stats
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq -- $S > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -- $S > $OUT/sq2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ('sq','sq2'):
    fs = glob.glob('$OUT/%s/*/*_counter_collection.csv' % d)
    if not fs: print('no counters for', d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']].add(r['Dispatch_Id'])
    cols = sorted({c for k in agg for c in agg[k]})
    with open('$OUT/pmc_%s.csv' % d, 'w') as f:
        f.write('Kernel_Name,Launches,' + ','.join(cols) + '\n')
        for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
            f.write('"%s",%d,' % (k[:110], len(n[k])) + ','.join('%d' % agg[k][c] for c in cols) + '\n')
PY
rm -rf $OUT/sq $OUT/sq2
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats_alone.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    n=r['Name'].replace('void exec::stage_kernel<rdr::','').replace('rdr::','')[:48]
    print('   %-48s calls %4s avg %8.3f max %8.3f ms %5.1f%%' % (n, r['Calls'], float(r['AverageNs'])/1e6, float(r['MaxNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
print('   total kernel ms', round(tot/1e6,1))
PY
