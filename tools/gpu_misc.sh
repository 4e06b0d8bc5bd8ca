timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 250 python tools/small_loop_timing.py 256 4 2>&1 | grep backward
timeout 250 python tools/small_loop_timing.py 256 16 2>&1 | grep backward
timeout 250 python tools/small_loop_timing.py 512 8 2>&1 | grep backward
