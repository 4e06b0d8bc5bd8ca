timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -1
for pa in 0 1; do echo pass_ahead=$pa; RDR_PASS_AHEAD=$pa timeout 250 python tools/small_loop_timing.py 256 4 2>&1 | grep backward; RDR_PASS_AHEAD=$pa timeout 250 python tools/small_loop_timing.py 512 4 2>&1 | grep backward; done
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'])"
