// exec.h (gfx950 build) -- how stage bodies become kernels on MI355X.
//
// The renderer's stage bodies (stages_*.h) are plain functors `void operator()(int lane)`.
// This header supplies, for the HIP build:
//   * RDR_FN            -- decoration of every math/stage function
//   * rdr::accum        -- gradient scatter: hardware fp64 atomic add (global_atomic_add_f64)
//   * exec::launch      -- one lane per thread, 256-thread workgroups (4 wave64 per workgroup)
//   * exec::DeviceBuf   -- HBM allocations through hipMalloc (no unified memory: the reference's
//                          cudaMallocManaged scratch, src/buffer.h:53-56, page-faults per launch)
//   * exec::compact / exec::trace_* -- hand-written kernels in compact.hip / trace.hip
// The only other implementation of this interface is the single-threaded debugging harness under
// tests/hostsim/, which is test infrastructure and is never loaded by the product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdexcept>
#include <string>

#define RDR_FN __host__ __device__ inline

namespace rdr {
__device__ inline void accum(double *p, double v) { unsafeAtomicAdd(p, v); }
__host__ inline void accum(double *p, double v) { *p += v; }   // host instantiation is never executed
}

namespace exec {

inline void check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

struct Context {
    hipStream_t stream = nullptr;
};
Context &ctx();

inline void *dmalloc(size_t bytes) {
    void *p = nullptr;
    check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc");
    return p;
}
inline void dfree(void *p) { if (p) (void)hipFree(p); }
inline void zero(void *p, size_t bytes) { if (bytes) check(hipMemsetAsync(p, 0, bytes, ctx().stream), "hipMemsetAsync"); }
inline void upload(void *dst, const void *src, size_t bytes) {
    if (bytes) check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream), "upload");
    check(hipStreamSynchronize(ctx().stream), "upload sync");
}
inline void download(void *dst, const void *src, size_t bytes) {
    if (bytes) check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream), "download");
    check(hipStreamSynchronize(ctx().stream), "download sync");
}
inline void sync() { check(hipStreamSynchronize(ctx().stream), "sync"); }

template <class F>
__global__ void __launch_bounds__(256) stage_kernel(F f, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) f(i);
}

template <class F>
inline void launch(int n, const F &f) {
    if (n <= 0) return;
    int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(stage_kernel<F>, dim3(blocks), dim3(256), 0, ctx().stream, f, n);
    check(hipGetLastError(), "stage launch");
}

} // namespace exec
