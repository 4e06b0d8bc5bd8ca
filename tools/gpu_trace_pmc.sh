#!/bin/bash
# SQ / TCP counters of the traversal kernels on the queues of tools/trace_ab.py: binary records vs 4-wide records.
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in binary wide; do
  if [ $v = binary ]; then export RDR_TRACE_BINARY=1; else unset RDR_TRACE_BINARY; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/${v}_sq -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 1024 /tmp/x.npz > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/${v}_sq2 -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 1024 /tmp/x.npz > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/${v}_tcp -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 1024 /tmp/x.npz > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum --kernel-trace --output-format csv -d $OUT/${v}_ta -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 1024 /tmp/x.npz > /dev/null 2>&1
  for k in sq sq2 tcp ta; do echo "== $v $k"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${v}_$k trace_ ; done
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
