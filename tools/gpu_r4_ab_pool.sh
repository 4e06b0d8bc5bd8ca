B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg"
show() { python -c "
import json,sys
d=json.loads(open('$1').read()); r=d['roofline']
print('$1'.split('/')[-1], '%.2f Msamples/s  %.0f ms/step  closest %.3f ms/launch' % (d['value'], d['ms_per_step'], r['mean_launch_ms']))"; }
for v in "X=1" "RDR_POOL_CAP_MB=98304" "RDR_BATCH_LANES=4194304" "RDR_BATCH_LANES=8388608" "X=2"; do
  n=$(echo $v | tr ' =' '__'); env $v timeout 300 $B 2>/dev/null | tail -1 > /tmp/b_$n.json; echo "== $v"; show /tmp/b_$n.json
done
