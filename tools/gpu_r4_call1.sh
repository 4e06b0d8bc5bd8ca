#!/bin/bash
# Round 4, first GPU call: the whole -m gpu suite (new: fuzz vs the live oracle, 1024x1024x16 batched fixture, tuning fields,
# two ranks on one GPU, caller stream), parked bytes under the new default cap, and bench.py A/B: default / big cache / one stream.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call1
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 > $OUT/pytest.log
unset RDR_PARITY_REPORT
cat $OUT/pytest.log | tail -30
python tools/parked_bytes.py 2>&1 | tail -1 | tee $OUT/parked.txt
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg"
show() { python -c "
import json,sys
d=json.loads(open('$1').read()); r=d['roofline']
print('$1'.split('/')[-1], '%.2f Msamples/s  %.0f ms/step  closest %.3f ms/launch frac %.3f' % (d['value'], d['ms_per_step'], r['mean_launch_ms'], r['frac']))"; }
for rep in 1 2; do
timeout 300 $B 2>/dev/null | tail -1 > $OUT/bench_default_$rep.json; show $OUT/bench_default_$rep.json
RDR_POOL_CAP_MB=98304 timeout 300 $B 2>/dev/null | tail -1 > $OUT/bench_bigcache_$rep.json; show $OUT/bench_bigcache_$rep.json
RDR_POOL_CAP_MB=98304 RDR_NO_OVERLAP=1 timeout 300 $B 2>/dev/null | tail -1 > $OUT/bench_onestream_$rep.json; show $OUT/bench_onestream_$rep.json
done
RDR_BATCH_LANES=4194304 timeout 300 $B 2>/dev/null | tail -1 > $OUT/bench_lanes22.json; show $OUT/bench_lanes22.json
