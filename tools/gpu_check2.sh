timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-alone-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'])"; done
RDR_COMPACT_3PASS=1 timeout 300 python bench.py --no-cpu-baseline --no-alone-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench 3-pass', d['value'], d['ms_per_step'])"
timeout 250 python tools/small_loop_timing.py 256 4 2>&1 | tail -3
RDR_COMPACT_3PASS=1 timeout 250 python tools/small_loop_timing.py 256 4 2>&1 | tail -1
