cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -- python $GRAFT_REPO_ROOT/tools/small_loop_timing.py 256 4 2>&1 | tail -5
