#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -k "not config_size" 2>&1 | tail -3
for cfg in "256 4" "256 4 move" "256 16" "256 16 move" "128 8 move"; do echo "== $cfg"; python tools/small_loop_timing.py $cfg 2>&1 | tail -4; done
echo "== no edge cache"; RDR_NO_EDGE_CACHE=1 python tools/small_loop_timing.py 256 4 2>&1 | tail -2
