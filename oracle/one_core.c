/* TEST INFRASTRUCTURE.  LD_PRELOAD shim: the process sees ONE processor, so the reference's thread pool
 * (/root/reference/src/parallel.cpp:228-255: std::thread::hardware_concurrency() - 1 workers + the caller) runs every
 * parallel_for on the calling thread, chunk after chunk, index after index -- the order of its fp32 atomic adds
 * (/root/reference/src/atomic.h:43-66) is then DEFINED: sample by sample, kernel by kernel, lane by lane.
 * libstdc++'s hardware_concurrency() is get_nprocs(). */
int get_nprocs(void) { return 1; }
int get_nprocs_conf(void) { return 1; }
