timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in 1 2 4; do echo "workers=$w"; RDR_WORKERS=$w timeout 250 python tools/small_loop_timing.py 256 4 2>&1 | grep backward;  RDR_WORKERS=$w timeout 250 python tools/small_loop_timing.py 256 16 2>&1 | grep backward; RDR_WORKERS=$w timeout 250 python tools/small_loop_timing.py 512 8 2>&1 | grep backward; done
