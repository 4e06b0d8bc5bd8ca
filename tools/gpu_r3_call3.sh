#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -k "not config_size" 2>&1 | tail -5
for cfg in "256 4" "256 16" "128 8" "512 4" "64 4"; do set -- $cfg; echo "== $1 x $1, $2 spp"; python tools/small_loop_timing.py $1 $2 2>&1 | tail -4; done
timeout 300 python bench.py --spp 32 --steps 2 --warmup 1 --no-profile --no-alone-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['self_check'], json.dumps(d['cpu_baseline'])[:1500])"
