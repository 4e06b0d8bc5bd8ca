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
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

#define RDR_FN __host__ __device__ inline
#define RDR_INLINE_CALL [[clang::always_inline]]
// Per-lane traversal stacks live in LDS (one column per thread of the 256-thread workgroup, entry k
// of lane t at [k * 256 + t]: conflict-free), not in scratch: large scratch frames make the HSA
// runtime re-allocate scratch per dispatch (tens of ms, see profiles/r1_notes.md).
#define RDR_DEV_FN __device__ inline
#define RDR_STACK_DECL(T, name, N) __shared__ T name##_lds[(N) * 256]; T *name = name##_lds + threadIdx.x
#define RDR_STACK_AT(name, k) name[(k) * 256]
// Stack of a resumable walk (persistent kernels): an LDS column on the GPU, a member of the walk state on the host.
#define RDR_WALK_STACK_MEMBER(T, name, N)
#define RDR_WALK_STACK(st, T, name, N, TAG) (rdr::lds_column<T, N, TAG>())
#define RDR_WALK_AT(stk, k) stk[(k) * 256]

namespace rdr {
template <class T, int N, int TAG>
__device__ inline T *lds_column() {
    __shared__ T tile[N * 256];
    return tile + threadIdx.x;
}
__host__ inline void *lds_column_host_stub() { return nullptr; }

// Gradient scatter.  Many lanes of a wave usually add to the SAME address (all pixels of a wall
// hit the same 4 vertices / the same constant albedo; every pixel adds to the camera), which would
// serialise 64 fp64 atomics on one L2 line.  So: the lanes that target the first lane's address sum
// their values first (xor-butterfly when all 64 lanes are active, otherwise a scalar loop over
// those lanes in lane order) and issue ONE atomic; that is repeated for up to kAccumRounds distinct
// addresses (lanes on different materials / walls), whoever is left issues its own hardware fp64 atomic.
//
// The accumulators are replicated (GradStore in render.cpp) and a wave adds to the replica its id selects, which spreads
// the atomics on one logical address over many lines / channels.  Two tiers, because the replica count that the memory
// budget allows depends on the size of what is replicated: SMALL tensors (camera, light intensities, constant albedos,
// the vertices of low-poly shapes, the top mip levels) are the ones every wave adds to, and get 256 replicas whatever
// else the scene holds; LARGE ones (image textures, big meshes) get as many as fit the budget.  Replica 0 of the small
// tier lies below `hot_end`, everything of the large tier above it: the address tells the tier.
struct ReplicaLayout {
    const double *hot_end;           // accumulators (replica 0) below this address belong to the small tier
    unsigned long long hot_stride;   // doubles between replicas of the small tier
    unsigned long long stride;       // ... of the large tier (0 = no replicas)
    unsigned hot_mask, mask;         // replicas - 1 (powers of two)
};
static __device__ ReplicaLayout g_rep = {nullptr, 0, 0, 0, 0};
// (callers that add a GROUP of consecutive accumulators -- accum_triple, accum_block -- call this for the first address only:
//  a tensor never straddles hot_end, render.cpp: GradStore checks its layout)
__device__ inline double *replica_of(double *p) {
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const bool hot = p < g_rep.hot_end;
    return p + (size_t)(wave & (hot ? g_rep.hot_mask : g_rep.mask)) * (hot ? g_rep.hot_stride : g_rep.stride);
}
constexpr int kAccumRounds = 3;                              // distinct addresses summed across the wave per call

// Sum of x over the 64 lanes of a wave, returned in every lane; ALL lanes must be active.  Four data-parallel-primitive
// steps inside each row of 16 lanes (swap neighbours, swap pairs, mirror the half row, mirror the row: after each the lane
// holds the sum of a group twice as large) -- two v_mov_b32_dpp and one v_add_f64 per step, no LDS traffic, no address
// arithmetic -- then the four row totals are read into scalar registers.  (__shfl_xor on a double is two ds_bpermute_b32
// through the LDS crossbar per step, six steps.)
__device__ inline double wave_sum(double x) {
#define RDR_DPP_ADD(ctrl)                                                                                  \
    {                                                                                                      \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xf, 0xf, true);         \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xf, 0xf, true);         \
        x += __hiloint2double(hi_, lo_);                                                                   \
    }
    RDR_DPP_ADD(0xB1)        // quad_perm:[1,0,3,2]
    RDR_DPP_ADD(0x4E)        // quad_perm:[2,3,0,1]
    RDR_DPP_ADD(0x141)       // row_half_mirror
    RDR_DPP_ADD(0x140)       // row_mirror
#undef RDR_DPP_ADD
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// fp64 add to an accumulator in DEVICE memory.  The accumulators' addresses reach the stages through tables in memory, so the
// compiler knows no address space for them and unsafeAtomicAdd() becomes flat_atomic_add_f64 (the flat path tests every lane's
// address against the LDS / scratch apertures before it goes to the L2); they are hipMalloc'ed blocks, always: global_atomic_add_f64.
typedef __attribute__((address_space(1))) double *GlobalDouble;
__device__ inline void global_add_f64(double *p, double v) { (void)__builtin_amdgcn_global_atomic_fadd_f64((GlobalDouble)p, v); }
__device__ inline void accum_rounds(double *p, double v, int ROUNDS);
__device__ inline void accum(double *p, double v) { accum_rounds(p, v, kAccumRounds); }
// Texel gradients: the lanes of a wave (64 neighbouring pixels) fall on a handful of texels of a magnified or coarse level,
// several lanes each: more distinct addresses are worth summing across the wave than for per-material / per-vertex data.
static __device__ int g_texel_rounds = 8;                    // (set_replicas uploads RDR_TEXEL_ROUNDS when it is set: experiments)
__device__ inline void accum_texel(double *p, double v) { accum_rounds(p, v, g_texel_rounds); }
__device__ inline void accum_rounds(double *p, double v, int ROUNDS) {
    p = replica_of(p);
    const unsigned long long act = __ballot(1);
    const unsigned long long addr = (unsigned long long)p;
    const int lane = threadIdx.x & 63;
    const int vlo = __double2loint(v), vhi = __double2hiint(v);
    unsigned long long rem = act;
    bool mine = true;                         // this lane's value has not been added yet
    for (int round = 0; round < ROUNDS && rem != 0; ++round) {      // wave-uniform trip count
        const int l = __ffsll((long long)rem) - 1;
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)addr, l);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(addr >> 32), l);
        const bool same = addr == (((unsigned long long)hi << 32) | lo);
        const unsigned long long m = __ballot(same) & rem;
        rem &= ~m;
        if (__popcll(m) < 2) continue;        // a lone lane adds for itself below
        double s;
        if (act == ~0ull) {
            s = wave_sum(same ? v : 0.0);
        } else {
            s = 0;
            unsigned long long mm = m;
            while (mm) {
                const int k = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                s += __hiloint2double(__builtin_amdgcn_readlane(vhi, k), __builtin_amdgcn_readlane(vlo, k));
            }
        }
        if (lane == l) global_add_f64(p, s);
        if (same) mine = false;
    }
    if (mine) global_add_f64(p, v);
}
// Three consecutive accumulators (a colour, a position): the lanes that share p also share p + 1 and p + 2, so the search for
// them is done ONCE for the triple (accum() three times repeated it per component: ~25 instructions per round and component).
__device__ inline void accum_triple_rounds(double *p, double x, double y, double z, int ROUNDS) {
    p = replica_of(p);
    const unsigned long long act = __ballot(1);
    const unsigned long long addr = (unsigned long long)p;
    const int lane = threadIdx.x & 63;
    unsigned long long rem = act;
    bool mine = true;
    for (int round = 0; round < ROUNDS && rem != 0; ++round) {      // wave-uniform trip count
        const int l = __ffsll((long long)rem) - 1;
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)addr, l);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(addr >> 32), l);
        const bool same = addr == (((unsigned long long)hi << 32) | lo);
        const unsigned long long m = __ballot(same) & rem;
        rem &= ~m;
        if (__popcll(m) < 2) continue;        // a lone lane adds for itself below
        double sx, sy, sz;
        if (act == ~0ull) {
            sx = wave_sum(same ? x : 0.0); sy = wave_sum(same ? y : 0.0); sz = wave_sum(same ? z : 0.0);
        } else {
            sx = sy = sz = 0;
            const int xlo = __double2loint(x), xhi = __double2hiint(x), ylo = __double2loint(y), yhi = __double2hiint(y);
            const int zlo = __double2loint(z), zhi = __double2hiint(z);
            unsigned long long mm = m;
            while (mm) {
                const int k = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                sx += __hiloint2double(__builtin_amdgcn_readlane(xhi, k), __builtin_amdgcn_readlane(xlo, k));
                sy += __hiloint2double(__builtin_amdgcn_readlane(yhi, k), __builtin_amdgcn_readlane(ylo, k));
                sz += __hiloint2double(__builtin_amdgcn_readlane(zhi, k), __builtin_amdgcn_readlane(zlo, k));
            }
        }
        if (lane == l) { global_add_f64(p, sx); global_add_f64(p + 1, sy); global_add_f64(p + 2, sz); }
        if (same) mine = false;
    }
    if (mine) { global_add_f64(p, x); global_add_f64(p + 1, y); global_add_f64(p + 2, z); }
}
__device__ inline void accum_triple(double *p, double x, double y, double z) { accum_triple_rounds(p, x, y, z, kAccumRounds); }
__device__ inline void accum_texel_triple(double *p, double x, double y, double z) { accum_triple_rounds(p, x, y, z, g_texel_rounds); }      // an rgb texel
__host__ inline void accum_triple(double *p, double x, double y, double z) { p[0] += x; p[1] += y; p[2] += z; }
__host__ inline void accum_texel_triple(double *p, double x, double y, double z) { p[0] += x; p[1] += y; p[2] += z; }
__host__ inline void accum(double *p, double v) { *p += v; }   // host instantiation is never executed
__host__ inline void accum_texel(double *p, double v) { *p += v; }
// The same add without the search for lanes that share the address: for per-vertex / per-texel data of large meshes the lanes
// of a wave almost never do, and the three search rounds cost ~25 scalar + vector instructions each, 18 times per lane in the
// bounce adjoint (6 000 of its 13 500 instructions per wave, profiles/r2_pmc_sq2.csv).
__device__ inline void accum_plain(double *p, double v) {
    p = replica_of(p);
    global_add_f64(p, v);
}
__host__ inline void accum_plain(double *p, double v) { *p += v; }
// fp64 atomic add on an ordinary buffer (no replicas: NOT for the gradient accumulators)
__device__ inline void atomic_add_f64(double *p, double v) { global_add_f64(p, v); }
__host__ inline void atomic_add_f64(double *p, double v) { *p += v; }
__device__ inline int atomic_fetch_add(int *p, int v) { return atomicAdd(p, v); }
__host__ inline int atomic_fetch_add(int *p, int v) { int o = *p; *p += v; return o; }
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
// Caching allocator behind every per-call buffer (trace.hip): blocks go back to a per-device free list instead of
// hipFree, so a render() / Scene of the same shape as an earlier one allocates nothing (hipMalloc + hipFree of the ~60
// arrays of a 1024 x 1024 render cost milliseconds and hipFree synchronises the device).  pool_trim() releases the cache.
void *pool_alloc(size_t bytes);
void pool_free(void *p);
void pool_trim();                     // hipFree of every parked block (synchronises the device first)
size_t pool_cached_bytes();           // bytes parked in the free lists right now
size_t memory_available();            // what a call can still get: free device memory + what is parked in the cache
double memory_held_by_others();       // fraction of the device's memory that neither is free nor came from this pool (torch's allocator, other processes)
void pool_set_cap(long long bytes);   // bound of the cache per device; negative = the default (rdr_set_pool_cap_mb)
size_t pool_cap_bytes();              // that bound
size_t pool_device_mallocs();          // number of hipMalloc calls made by the pool so far (tests: steady state adds none)
inline std::atomic<size_t> &host_count_reads_ref() { static std::atomic<size_t> n{0}; return n; }
inline size_t host_count_reads() { return host_count_reads_ref().load(); }     // live-lane counts read back by the host so far
inline void zero(void *p, size_t bytes) { if (bytes) check(hipMemsetAsync(p, 0, bytes, ctx().stream), "hipMemsetAsync"); }
inline void upload(void *dst, const void *src, size_t bytes) {
    if (bytes) check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream), "upload");
    check(hipStreamSynchronize(ctx().stream), "upload sync");
}
inline void download(void *dst, const void *src, size_t bytes) {
    if (bytes) check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream), "download");
    check(hipStreamSynchronize(ctx().stream), "download sync");
}
inline void copy_dev(void *dst, const void *src, size_t bytes) {          // device -> device, stream-ordered
    if (bytes) check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().stream), "copy_dev");
}
inline void sync() { check(hipStreamSynchronize(ctx().stream), "sync"); }
constexpr bool kDeviceEdgeTrees = true;         // the edge hierarchies are built by kernels (edges_gpu.cpp)
constexpr bool kDeviceBvh = true;               // ... and so is the triangle hierarchy (bvh_gpu.cpp)
__host__ __device__ inline void gather_stats_add(long, long, int, int) {}      // a hook of the CPU debugging harness
inline void device_sync() { (void)hipDeviceSynchronize(); }        // every stream of the device (error paths; never throws)
// Batched transfers for the Scene build (trace.hip): every array goes through one pinned staging buffer, the copies are
// queued on the stream and ONE synchronisation ends the batch -- a pageable hipMemcpy + sync per array cost 30-200 us each,
// ~50 of them per Scene.
void upload_async(void *dst, const void *src, size_t bytes);      // src may be reused as soon as this returns
void upload_flush();                                              // all queued uploads have landed
struct DownloadItem { void *dst; const void *src; size_t bytes; };
void download_batch(const DownloadItem *items, int n);            // device -> host, one synchronisation
// Constant tables shared by every Scene on a device (Sobol' direction numbers, LTC matrices): uploaded once per device.
const void *device_constant(const void *host, size_t bytes);
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }

// Side streams for stages that do not depend on each other (the edge-pick walks and the bounce adjoint of one path
// depth): each of them keeps a fraction of the lanes busy and waits on dependent loads, so they fill each other's gaps.
// StreamScope redirects every launch made inside it; Fence orders two streams (record on the producer, gate the consumer).
hipStream_t side_stream(int k);          // trace.hip: two non-blocking streams per device, created on first use
struct StreamScope {
    hipStream_t saved;
    explicit StreamScope(hipStream_t s) : saved(ctx().stream) { ctx().stream = s; }
    ~StreamScope() { ctx().stream = saved; }
    StreamScope(const StreamScope &) = delete;
};
// Helper host threads that live as long as the process; each runs one job at a time on a non-blocking stream of the
// device the caller is on (render() drives some of the samples of a gradient render from them).  Their thread-local
// scratch (compaction counters, side streams) is created once and reused by later jobs.
constexpr int kMaxHelpers = 3;
class SecondThread {
public:
    static SecondThread &get(int k = 0) {                          // helper k OF THE CALLING THREAD'S DEVICE; never destroyed
        static std::mutex lock;
        static SecondThread *t[16][kMaxHelpers] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(lock);
        SecondThread *&p = t[dev & 15][k];
        if (!p) p = new SecondThread();
        return *p;
    }
    void start(std::function<void()> job) {
        int dev = 0;
        check(hipGetDevice(&dev), "hipGetDevice");
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return state_ == 0; });              // never overwrite a job that has not been taken / finished
        job_ = std::move(job); device_ = dev; state_ = 1;
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return state_ == 0; });
        if (failure_) { std::exception_ptr f = failure_; failure_ = nullptr; std::rethrow_exception(f); }
    }
private:
    SecondThread() { std::thread([this] { loop(); }).detach(); }
    void loop() {
        hipStream_t streams[16] = {};
        for (;;) {
            std::function<void()> job; int dev;
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return state_ == 1; }); job = std::move(job_); dev = device_; state_ = 2; }
            std::exception_ptr failure;
            try {
                check(hipSetDevice(dev), "hipSetDevice");
                hipStream_t &s = streams[dev & 15];
                if (!s) {
                    static const bool low = std::getenv("RDR_HELPER_LOW") != nullptr;        // experiment: helpers at the lowest priority
                    int least = 0, greatest = 0;
                    if (low && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
                        check(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least), "hipStreamCreateWithPriority");
                    else check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
                }
                ctx().stream = s;
                job();
                check(hipStreamSynchronize(s), "second stream sync");
            } catch (...) { failure = std::current_exception(); }
            { std::unique_lock<std::mutex> lk(m_); failure_ = failure; state_ = 0; }
            cv_.notify_all();
        }
    }
    std::mutex m_; std::condition_variable cv_;
    std::function<void()> job_; int device_ = 0; int state_ = 0;      // 0 idle, 1 job posted, 2 running
    std::exception_ptr failure_;
};

struct Fence {
    hipEvent_t e = nullptr;
    Fence() { check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); }
    ~Fence() { if (e) (void)hipEventDestroy(e); }
    Fence(const Fence &) = delete;
    void after(hipStream_t producer) { check(hipEventRecord(e, producer), "hipEventRecord"); }
    void gate(hipStream_t consumer) { check(hipStreamWaitEvent(consumer, e, 0), "hipStreamWaitEvent"); }
};

// Replicated gradient accumulators (see GradStore in render.cpp): a power of two, at most 256 and as many as fit `budget`.
// Must be called from the translation unit that instantiates the stage kernels.
inline int choose_replicas(size_t replica_bytes, size_t budget) {
    int r = 256;
    while (r > 1 && replica_bytes * (size_t)r > budget) r >>= 1;
    return r;
}
// Large tier: 256 MiB; a long job (pixels x samples of this call) may take 64 bytes per sample, up to 1 GiB -- the texel
// gradients of a scene with several image textures per material are > 100 MB per replica, and one or two replicas leave
// the adjoint stages waiting on the texels' lines (config-5 stand-in with environment map: 17.1 -> 18.5 Msamples/s at
// 1 GiB).  Zeroing and summing 1 GiB costs ~0.5 ms.  RDR_REPLICA_MB: fixed budget, for experiments.
inline size_t replica_budget(size_t job_samples) {
    static const size_t mb = [] { const char *e = std::getenv("RDR_REPLICA_MB"); return e ? (size_t)std::atoi(e) : (size_t)0; }();
    if (mb) return mb << 20;
    const size_t lo = (size_t)256 << 20, hi = (size_t)1 << 30, want = job_samples * 64;
    return want < lo ? lo : (want > hi ? hi : want);
}
inline void set_replicas(const rdr::ReplicaLayout &layout) {
    check(hipMemcpyToSymbolAsync(HIP_SYMBOL(rdr::g_rep), &layout, sizeof(layout), 0, hipMemcpyHostToDevice, ctx().stream), "set_replicas");
    static const int texel_rounds = [] { const char *e = std::getenv("RDR_TEXEL_ROUNDS"); return e ? std::atoi(e) : 0; }();
    if (texel_rounds > 0) check(hipMemcpyToSymbolAsync(HIP_SYMBOL(rdr::g_texel_rounds), &texel_rounds, sizeof(int), 0, hipMemcpyHostToDevice, ctx().stream), "set_replicas");
    check(hipStreamSynchronize(ctx().stream), "set_replicas sync");
}

// How many items a launch covers: `upper` is what the host knows (an upper bound, or the exact number when `dev` is null);
// `dev` points at the exact number in device memory, written by the compaction that produced the list.  The per-depth
// live-lane counts never come back to the host (src/pathtracer.cpp:292,590,833 read them after every stage): grids are sized by
// the bound and every kernel trims itself.
struct Count {
    const int *dev; int upper;
    Count(int n) : dev(nullptr), upper(n) {}
    Count(const int *d, int u) : dev(d), upper(u) {}
};

// A stage may ask for a minimum number of workgroups per CU (registers are then capped, spilling if need be):
// `static constexpr int kMinBlocksPerCU` in the functor; default: whatever the body needs.
template <class F, class = void> struct MinBlocks { static constexpr int value = 1; };
template <class F> struct MinBlocks<F, decltype((void)F::kMinBlocksPerCU)> { static constexpr int value = F::kMinBlocksPerCU; };

template <class F>
__global__ void __launch_bounds__(256, MinBlocks<F>::value) stage_kernel(F f, int n, const int *count) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (count) { const int c = *count; n = c < n ? c : n; }
    // a stage body that is instantiated twice (plain + LeanStage) must still be inlined into each kernel:
    // an out-of-line call would pass the whole functor through scratch
    if (i < n) { RDR_INLINE_CALL f(i); }
}

// Persistent waves with lane refill, for walks whose length varies wildly from lane to lane (the NEE-mode edge pick:
// median 20 steps, 95th percentile 640 -- a one-item-per-lane kernel keeps 25 % of the SIMD busy).  W provides
//   State; bool begin(int item, State&)   set up the walk; false = nothing to walk (finish is still called)
//          bool step(State&)              one step; true when the walk is complete
//          void finish(State&)            write the result
// A wave keeps running: whenever >= kRefillIdle lanes are idle they take the next items off a global counter
// (one atomic per refill), then every busy lane advances kWalkSteps steps.
constexpr int kRefillIdle = 16;
constexpr int kWalkSteps = 16;
// W may have `bool gate_closed() const`: nothing of this launch has anything to do (decided by a device-side flag an earlier stage
// wrote): the kernel returns before it touches the item counter -- otherwise EVERY wave takes its items off that one counter, an
// atomic per 64 items on one address (8.4 M slots: 131 k serialised atomics = 4.5 ms to find out that no slot wanted a walk).
template <class W, class = void> struct WalkGate { __device__ static bool closed(const W &) { return false; } };
template <class W> struct WalkGate<W, decltype((void)&W::gate_closed)> { __device__ static bool closed(const W &w) { return w.gate_closed(); } };
template <class W>
__global__ void __launch_bounds__(256) persistent_kernel(W w, int n, const int *count, int *next_item) {
    if (WalkGate<W>::closed(w)) return;
    if (count) { const int c = *count; n = c < n ? c : n; }
    typename W::State st;
    bool busy = false;
    const int lane = threadIdx.x & 63;
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        const int nidle = __popcll(idle);
        if (nidle >= kRefillIdle) {
            const int leader = __ffsll((long long)idle) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(next_item, nidle);
            base = __shfl(base, leader);
            if (!busy) {
                const int item = base + __popcll(idle & ((1ull << lane) - 1ull));
                if (item < n) {
                    if (w.begin(item, st)) busy = true;
                    else w.finish(st);
                }
            }
            if (__ballot(busy) == 0ull && base + nidle >= n) break;
        }
#pragma unroll 1
        for (int it = 0; it < kWalkSteps; ++it) {
            if (busy && w.step(st)) { w.finish(st); busy = false; }
        }
    }
}
// The same walk protocol with WAVE-LOCAL refill (the scheme of trace_refill_kernel, trace.hip) over chunks that the waves take
// off a global counter: a wave owns 64 x K consecutive items at a time and its idle lanes take the next unclaimed ones of that
// chunk -- a ballot and a popcount; when the chunk is handed out the wave takes the next chunk (ONE atomic per 64 K items:
// 16 k per 4 M items, where persistent_kernel above issues one per refill) while its busy lanes go on, so no lane waits for the
// longest walk of a chunk, only for the longest walk of the launch.  Items that are neighbours in the list (neighbouring pixels:
// walks of similar length through the same tree nodes) stay in one wave.  For walks that are long and uneven (the hierarchical
// edge pick: 20 ... 300 steps).
#ifndef RDR_CHUNKED_MIN_BLOCKS             // (variant builds: workgroups per CU the chunked walks are compiled for)
#define RDR_CHUNKED_MIN_BLOCKS 1
#endif
template <class W>
__global__ void __launch_bounds__(256, RDR_CHUNKED_MIN_BLOCKS) chunked_kernel(W w, int n, const int *count, int items_per_lane, int idle_min, int steps, int *next_chunk) {
    if (count) { const int c = *count; n = c < n ? c : n; }
    const int chunk = 64 * items_per_lane;
    const int lane = threadIdx.x & 63;
    int next = 0, end = 0;                                  // wave-uniform: the part of the current chunk not handed out yet
    bool more = true;                                       // chunks may be left on the counter
    typename W::State st;
    bool busy = false;
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        const int nidle = __popcll(idle);
        if (nidle >= idle_min || nidle == 64) {
            if (next >= end && more) {
                int base = 0;
                if (lane == 0) base = atomicAdd(next_chunk, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n) more = false;
                else { next = base; end = base + chunk < n ? base + chunk : n; }
            }
            if (next < end) {
                if (!busy) {
                    const int item = next + __popcll(idle & ((1ull << lane) - 1ull));
                    if (item < end) {
                        if (w.begin(item, st)) busy = true;
                        else w.finish(st);
                    }
                }
                next += nidle;
            }
        }
        if (__ballot(busy) == 0ull) { if (next >= end && !more) break; continue; }
#pragma unroll 1
        for (int it = 0; it < steps; ++it) {
            if (busy && w.step(st)) { w.finish(st); busy = false; }
        }
    }
}
int *persistent_counter();          // trace.hip: ring of zeroed ints, one per launch
template <class W>
inline void launch_chunked(Count n, const W &w, int items_per_lane = 4, int idle_min = 16, int steps = 8) {
    if (n.upper <= 0) return;
    const int per_block = 4 * 64 * items_per_lane;
    // as many workgroups as there are chunks of four, but no more than a few per CU: later ones would find the counter exhausted
    const int blocks = (int)std::min<long long>(((long long)n.upper + per_block - 1) / per_block, 256 * 6);
    hipLaunchKernelGGL(chunked_kernel<W>, dim3(blocks), dim3(256), 0, ctx().stream, w, n.upper, n.dev, items_per_lane, idle_min, steps, persistent_counter());
    check(hipGetLastError(), "chunked launch");
}
template <class W>
inline void launch_persistent(Count n, const W &w) {
    if (n.upper <= 0) return;
    int blocks = std::min((n.upper + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(persistent_kernel<W>, dim3(blocks), dim3(256), 0, ctx().stream, w, n.upper, n.dev, persistent_counter());
    check(hipGetLastError(), "persistent launch");
}

template <class F>
inline void launch(Count n, const F &f) {
    if (n.upper <= 0) return;
    int blocks = (n.upper + 255) / 256;
    hipLaunchKernelGGL(stage_kernel<F>, dim3(blocks), dim3(256), 0, ctx().stream, f, n.upper, n.dev);
    check(hipGetLastError(), "stage launch");
}

} // namespace exec

// ---------------------------------------------------------------------------------------------
// Order-preserving stream compaction of the live-lane list (replaces thrust::copy_if /
// remove_if, src/active_pixels.cpp:17-49).  Three small kernels, wave64-native:
//   count   : each 256-thread workgroup handles 1024 candidates; per-wave __ballot + popcount,
//             4 wave totals combined through LDS -> one count per workgroup
//   scan    : one workgroup turns the per-workgroup counts into exclusive offsets (+ total)
//   scatter : recomputes the ballots; lane rank = mbcnt(ballot) + wave offset + workgroup offset
// Stability (needed because Sobol' slots of the secondary-edge sampler are assigned by compacted
// rank, src/pathtracer.cpp:504-505) follows from ranks being prefix sums in lane order.
// ---------------------------------------------------------------------------------------------
#include "../bvh.h"
namespace exec {

constexpr int kCompactItems = 4;                       // candidates per thread
constexpr int kCompactTile = 256 * kCompactItems;      // per workgroup

__device__ inline int lane_prefix(unsigned long long ballot) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0));
}

template <class P>
__global__ void __launch_bounds__(256) compact_count(const int *in, int n, const int *count, P pred, int *block_counts) {
    __shared__ int wave_tot[4];
    if (count) { const int c = *count; n = c < n ? c : n; }
    int base = blockIdx.x * kCompactTile;
    int cnt = 0;
    for (int it = 0; it < kCompactItems; ++it) {
        int i = base + it * 256 + threadIdx.x;
        bool keep = false;
        if (i < n) { int p = in ? in[i] : i; keep = pred(p); }
        unsigned long long b = __ballot(keep);
        cnt += __popcll(b);
    }
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// exclusive scan of up to 256*16 workgroup counts by one workgroup.  Publishes the number kept in `total_dev` (device memory,
// for the kernels that consume the list; `base` is added: the list may continue an earlier one) and in `total` (mapped host
// memory + ticket, for the rare host read).  `dyn` (optional): the edge sampler's dimension counter advances by `inc` when this
// compaction closes a bounce that had lanes to run (src/pathtracer.cpp:590-706: `used += 7` per executed iteration).
static __global__ void __launch_bounds__(256) compact_scan(int *block_counts, int nblocks, int *total, int ticket, int *total_dev,
                                                           const int *base, int n_in, const int *count_in, int *dyn, int inc) {
    __shared__ int part[256];
    int per = (nblocks + 255) / 256;
    int beg = threadIdx.x * per, end = min(beg + per, nblocks);
    int s = 0;
    for (int i = beg; i < end; ++i) s += block_counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { int t = part[i]; part[i] = run; run += t; }
        const int all = run + (base ? *base : 0);
        if (total_dev) *total_dev = all;
        if (dyn) {
            if (count_in) { const int c = *count_in; n_in = c < n_in ? c : n_in; }
            if (n_in > 0) *dyn += inc;
        }
        total[0] = all;
        // the host spins on the ticket (mapped pinned memory): release at system scope publishes the count first
        __hip_atomic_store(total + 1, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = beg; i < end; ++i) { int t = block_counts[i]; block_counts[i] = run; run += t; }
}

template <class P>
__global__ void __launch_bounds__(256) compact_scatter(const int *in, int n, const int *count, P pred, const int *block_offsets, int *out,
                                                        const int *out_base, int *pos_out) {
    __shared__ int wave_tot[kCompactItems][4];
    if (count) { const int c = *count; n = c < n ? c : n; }
    if (out_base) { out += *out_base; if (pos_out) pos_out += *out_base; }      // appended lists: positions line up with the items
    int base = blockIdx.x * kCompactTile;
    int wave = threadIdx.x >> 6;
    int val[kCompactItems]; bool keep[kCompactItems]; int rank[kCompactItems];
    for (int it = 0; it < kCompactItems; ++it) {
        int i = base + it * 256 + threadIdx.x;
        keep[it] = false; val[it] = 0;
        if (i < n) { val[it] = in ? in[i] : i; keep[it] = pred(val[it]); }
        unsigned long long b = __ballot(keep[it]);
        rank[it] = lane_prefix(b);
        if ((threadIdx.x & 63) == 0) wave_tot[it][wave] = __popcll(b);
    }
    __syncthreads();
    int off = block_offsets[blockIdx.x];
    for (int it = 0; it < kCompactItems; ++it) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wave_tot[it][w];
        if (keep[it]) {
            out[off + before + rank[it]] = val[it];
            if (pos_out) pos_out[off + before + rank[it]] = base + it * 256 + (int)threadIdx.x;      // where the item came from
        }
        off += wave_tot[it][0] + wave_tot[it][1] + wave_tot[it][2] + wave_tot[it][3];
    }
}

// `total` lives in pinned, device-mapped host memory: the scan kernel stores the count straight into
// it, so reading it back costs one stream synchronisation and no copy launch (a pageable 4-byte
// hipMemcpy measured ~195 us per call in the rocprofv3 trace, profiles/r1_notes.md).
struct CompactScratch { int *block_counts = nullptr; int *total = nullptr; volatile int *total_host = nullptr; int capacity = 0; int ticket = 0; };
CompactScratch &compact_scratch(int nblocks, int which = 0);      // `which`: compactions that may be in flight together (two streams of one host thread) use different scratch

// `out` must not alias `in`: a workgroup may scatter into a tile that an earlier-numbered workgroup has not read yet.
// compact_dev: nothing comes back to the host.  The kept items go to out[*append_at ...] (append_at null: out[0 ...]) and
// the returned Count says how many items `out` now holds (in device memory) and what the host can bound it by.
int *new_count();                    // trace.hip: a device int from a per-thread ring, for one compaction's result
// `pos_out` (optional): pos_out[k] = position in the input list of the k-th kept item.
template <class P>
inline Count compact_dev(const int *in, Count n, int *out, const P &pred, const Count *append_at = nullptr, int *dyn = nullptr, int inc = 0,
                         int *pos_out = nullptr, int scratch = 0) {
    const int base_upper = append_at ? append_at->upper : 0;
    if (n.upper <= 0) return append_at ? *append_at : Count(0);
    int nblocks = (n.upper + kCompactTile - 1) / kCompactTile;
    if (nblocks > 256 * 4096) throw std::runtime_error("compact: input too large");
    CompactScratch &sc = compact_scratch(nblocks, scratch);
    hipStream_t st = ctx().stream;
    int *result = new_count();
    const int *base = append_at ? append_at->dev : nullptr;
    if (append_at && !append_at->dev) throw std::runtime_error("compact: append position must live on the device");
    hipLaunchKernelGGL(compact_count<P>, dim3(nblocks), dim3(256), 0, st, in, n.upper, n.dev, pred, sc.block_counts);
    const int ticket = ++sc.ticket;
    hipLaunchKernelGGL(compact_scan, dim3(1), dim3(256), 0, st, sc.block_counts, nblocks, sc.total, ticket, result, base, n.upper, n.dev, dyn, inc);
    hipLaunchKernelGGL(compact_scatter<P>, dim3(nblocks), dim3(256), 0, st, in, n.upper, n.dev, pred, sc.block_counts, out, base, pos_out);
    check(hipGetLastError(), "compact launch");
    return Count(result, n.upper + base_upper);
}
// The host-visible form (loop decisions the host has to take itself): the same kernels, then a spin on the ticket the scan
// kernel publishes in mapped pinned memory -- a few microseconds; hipStreamSynchronize wakes the thread ~20 us after the
// stream drains and also waits for the scatter kernel, which the next launch is ordered behind anyway.
template <class P>
inline int compact(const int *in, int n, int *out, const P &pred) {
    if (n <= 0) return 0;
    host_count_reads_ref()++;
    CompactScratch &sc = compact_scratch((n + kCompactTile - 1) / kCompactTile);
    (void)compact_dev(in, Count(n), out, pred);
    hipStream_t st = ctx().stream;
    const int ticket = sc.ticket;
    for (long spins = 0; sc.total_host[1] != ticket; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(st) != hipErrorNotReady) {   // finished or failed without publishing
            check(hipStreamSynchronize(st), "compact sync");
            if (sc.total_host[1] != ticket) throw std::runtime_error("compact: count was not published");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return sc.total_host[0];
}
// k times a device-side count (the two lanes per slot of an edge pass), as a new device-side count.
static __global__ void scale_count_kernel(const int *src, int upper, int k, int *dst) {
    const int c = *src;
    *dst = k * (c < upper ? c : upper);
}
inline Count scaled_count(Count c, int k) {
    if (!c.dev) return Count(k * c.upper);
    int *dst = new_count();
    hipLaunchKernelGGL(scale_count_kernel, dim3(1), dim3(1), 0, ctx().stream, c.dev, c.upper, k, dst);
    return Count(dst, k * c.upper);
}
// A count the host needs after all (a skip decision, a debug dump): synchronise and read it.
inline int read_count(Count c) {
    if (!c.dev) return c.upper;
    host_count_reads_ref()++;
    int v = 0;
    download(&v, c.dev, sizeof(int));
    return v < c.upper ? v : c.upper;
}

// ---- traversal kernels (trace.hip) --------------------------------------------------------------
struct TraceStats {
    double closest_ms = 0, any_ms = 0;                 // sum of the launches' durations
    double closest_union_ms = 0, any_union_ms = 0;     // time with at least one launch of the kind in flight
    uint64_t closest_launches = 0, any_launches = 0, closest_rays = 0, any_rays = 0, nodes[2] = {0, 0}, tris[2] = {0, 0}, wide_nodes[2] = {0, 0};   // node records: 32-byte binary / 128-byte 4-wide
    bool timing = false, counting = false;
};
TraceStats &trace_stats();
void trace_stats_collect();     // folds pending hipEvent pairs / device counters into trace_stats()
// `coherent`: neighbouring queue slots hold neighbouring rays that finish together (camera rays): the plain kernel, whatever the size
void trace(const rt::BvhD &bvh, const rt::RayRec *rays, rt::HitRec *hits, Count n, bool any, bool coherent = false);
// Host threads render() may drive samples from (rdr_tuning::workers = 1 turns the second one off).
inline int sample_workers(int lanes, int samples, bool batches = false) {
    // (rdr_tuning::workers overrides all of this, render.cpp)
    // sample batches (render.cpp): two chains of launches in flight, more do not help (tools/gpu_batch_grid.sh)
    // ... also of large batches (round 6): at 1024 x 1024 two workers with 4-sample batches beat one worker with 8-sample
    // batches in the same memory, 69.3 -> 70.8 Msamples/s (profiles/r5_notes.md "two chains of launches in flight")
    if (batches) return samples >= 2 ? 2 : 1;
    // measured (bunny_box backward, round 2): 256x256x4 spp 13.8 / 14.6 / 15.2 ms with 2 / 3 / 4 workers, 256x256x16 spp
    // 52.1 / 47.9 / 46.7 / 48.7 ms with 3 / 4 / 6 / 8; 512x512x8 spp 92 -> 83 ms with a second worker; at 1024x1024 the second
    // worker adds 2-4 % and stretches every kernel it shares the GPU with
    if (lanes >= (1 << 19)) return 1;
    return (lanes <= (1 << 17) && samples >= 8) ? std::min(4, 1 + kMaxHelpers) : 2;
}
void select_device(int use_gpu, int gpu_index);

} // namespace exec

namespace rdr {
// What GradStore (render.cpp) tells its accumulator backend: the small tier has been laid out / a small tensor is about to be
// folded into the caller's floats / the fold is done.  The fp64 replicas need none of it (tests/hostsim/exec.h keeps the
// reference's fp32 accumulation order beside them through these three).
inline void accumulators_laid_out(double *, size_t) {}
inline void accumulator_before_fold(double *, double *, size_t) {}
inline void accumulators_folded() {}
__device__ inline void accum_f32(float *p, float v) { atomicAdd(p, v); }
__host__ inline void accum_f32(float *p, float v) { *p += v; }
}
