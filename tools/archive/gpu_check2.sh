timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline']['alone']['frac'])"
