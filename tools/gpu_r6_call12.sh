#!/bin/bash
# Round 6, call 12: what the latency-bound stage kernels wait for -- vector-memory latency (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM),
# issue mix, cache hit rates, per kernel, each kernel alone (one worker, one stream).  Counter passes only (--pmc + --kernel-trace).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || rocprofv3-avail list > $OUT/counters_available.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9a-z\[\]]*\|TCC_[A-Z_0-9a-z\[\]]*\|TA_[A-Z_0-9a-z\[\]]*" $OUT/counters_available.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
export RDR_NO_OVERLAP=1 RDR_WORKERS=1 RDR_BATCH=8
P="python $GRAFT_REPO_ROOT/bench.py --inner --res 1024 --max-bounces 4 --workload bunny_box"
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- $P > $OUT/p$i.log 2>&1 || echo "pass $i failed: $pmc"
done
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
names = ['AdjBounceScatter', 'AdjBounceNee', 'BounceContrib', 'BounceSample', 'AdjPrimary', 'SecEdgePickHDescend', 'trace_refill_kernel<false', 'trace_refill_kernel<true', 'PrimaryEdgeDerivatives', 'SecEdgeGatherN', 'SecEdgeSetup']
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for k in names:
            if k in r['Kernel_Name']:
                agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
with open(out + '/stage_counters.txt', 'w') as fo:
    for k in names:
        if k not in agg: continue
        fo.write('== %s\n' % k); print('==', k)
        for c in sorted(agg[k]):
            line = '  %-44s %18.0f  (%d launches)' % (c, agg[k][c], len(n[(k, c)])); fo.write(line + '\n'); print(line)
        a = agg[k]
        def g(x): return a.get(x, 0.0)
        if g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR') > 0:
            line = '  -> mean latency of a vector-memory instruction: %.0f cycles; per wave: %.1f vmem instructions, %.0f wave cycles' % (
                g('SQ_INST_LEVEL_VMEM') / (g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')), (g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / max(g('SQ_WAVES'), 1), g('SQ_WAVE_CYCLES') / max(g('SQ_WAVES'), 1))
            fo.write(line + '\n'); print(line)
PY
rm -rf $OUT/p[0-9]*/
