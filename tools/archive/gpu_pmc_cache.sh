cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD)_[A-Z0-9_]+(\[[0-9]+\])?" | sed 's/\[[0-9]*\]//' | sort -u | tr '\n' ' ' | head -c 6000
echo
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cache/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-80; }
run a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run b TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run c TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
