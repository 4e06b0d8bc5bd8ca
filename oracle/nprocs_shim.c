/* TEST INFRASTRUCTURE.  LD_PRELOAD shim: the process sees ORACLE_NPROCS processors (default 1).
 * The reference sizes its thread pool by std::thread::hardware_concurrency() (/root/reference/src/parallel.cpp:228-255), which
 * libstdc++ takes from get_nprocs().
 *   ORACLE_NPROCS=1 (tests/golden/make_ref_order.py): no worker threads, every parallel_for runs on the calling thread, chunk after
 *     chunk, index after index -- the order of the reference's fp32 atomic adds (src/atomic.h:43-66) is then DEFINED: sample by
 *     sample, kernel by kernel, lane by lane.
 *   ORACLE_NPROCS=16 / 32 / 64 (bench.py: cpu_baseline): the reference on fewer threads than the box has -- its per-call pool
 *     does not scale to 256 threads on small frames, so the fair CPU baseline is the best of a sweep. */
#include <stdlib.h>
static int shim_nprocs(void) {
    const char *e = getenv("ORACLE_NPROCS");
    int n = e ? atoi(e) : 1;
    return n > 0 ? n : 1;
}
int get_nprocs(void) { return shim_nprocs(); }
int get_nprocs_conf(void) { return shim_nprocs(); }
