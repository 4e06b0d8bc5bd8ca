#!/bin/bash
# SQ counters of the refilling traversal kernel vs the speculative-leaf kernel on the 4 M-ray queues of tools/trace_ab.py 2048
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_trace_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RDR_TRACE_REFILL_ALL=1 TRACE_AB_REPS=4
for v in refill spec; do
  if [ $v = spec ]; then export RDR_TRACE_SPEC=${SPEC:-24,12}; else unset RDR_TRACE_SPEC; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/${v}_sq -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 2048 /tmp/x.npz > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/${v}_sq2 -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 2048 /tmp/x.npz > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN --kernel-trace --output-format csv -d $OUT/${v}_sq3 -- python $GRAFT_REPO_ROOT/tools/trace_ab.py --worker 2048 /tmp/x.npz > /dev/null 2>&1
  for k in sq sq2 sq3; do echo "== $v $k"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${v}_$k trace_ ; done
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
