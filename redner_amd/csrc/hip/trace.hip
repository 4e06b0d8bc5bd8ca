// trace.hip -- closest-hit / any-hit traversal kernels for gfx950 and the small amount of device
// state behind exec.h (stream, compaction scratch, timing).
//
// One ray per lane, 256-thread workgroups, per-lane traversal stack in an LDS column (16-bit entries when the
// hierarchy allows); nodes and triangles are read through L1/L2 (the whole bunny_box hierarchy is ~1 MB and lives in
// the 4 MiB per-XCD L2).  rt::traverse<> is the shared per-ray routine, so results are bit-identical to the
// brute-force rule in raytri.h.  Three kernels, chosen per launch by exec::trace():
//   trace_kernel          binary 32-byte records, one ray per lane                       (coherent queues; 2^19 < n < 2^22)
//   trace_wide_kernel     4-wide 128-byte records, one ray per lane: half the steps      (queues of <= 2^19 rays)
//   trace_refill_kernel   binary records, idle lanes take the next rays of the wave's    (incoherent queues sized for >= 2^22
//                         own chunk                                                       lanes)
// Measured numbers, what bounds a launch and the loop shapes that were tried and rejected are in DESIGN.md section 3
// ("Traversal kernel", "Round 3") and profiles/r1_notes.md, r2_notes.md, r3_notes.md.
#include "exec.h"
#include "../tuning.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <unordered_map>
#include <vector>

namespace exec {

Context &ctx() { static thread_local Context c; return c; }

void select_device(int use_gpu, int gpu_index) {
    if (!use_gpu)
        throw std::runtime_error("redner_amd renders on an MI355X (gfx950) GPU only: Scene(use_gpu=False) "
                                 "is not supported and there is no CPU fallback");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw std::runtime_error("redner_amd: no HIP device is visible (hipGetDeviceCount) -- a gfx950 GPU is required");
    if (gpu_index < 0) gpu_index = 0;
    if (gpu_index >= count) throw std::runtime_error("redner_amd: gpu_index out of range");
    check(hipSetDevice(gpu_index), "hipSetDevice");
}

namespace {
struct Pool {
    std::mutex lock;
    std::multimap<size_t, void *> free_blocks[16];              // per device, by capacity
    std::unordered_map<void *, std::pair<size_t, int>> live;    // block -> (capacity, device)
    size_t parked[16] = {};                                     // bytes in free_blocks[d] (kept in step: no walk per free)
    size_t from_driver[16] = {};                                // bytes this pool currently holds from hipMalloc (live + parked)
    size_t device_mallocs = 0;
    long long cap_override = -1;                                // rdr_set_pool_cap_mb
};
Pool &pool() { static Pool *p = new Pool(); return *p; }       // never destroyed: blocks may be returned during exit

// The cache is bounded: a torch process shares the device with torch's own allocator, which cannot reclaim what is parked
// here.  Default: a quarter of the device's memory but no more than 8 GiB (round 6; 16 GiB in rounds 4-5; round 3 parked up
// to 72 GB -- the 48 GB of a 2^24-lane sample batch -- for +4 % at 1024 x 1024).  More is a decision of the caller:
// rdr_set_pool_cap_mb / RDR_POOL_CAP_MB (bench.py owns its GPU, raises the bound and says so in its line);
// rdr_trim_cache() releases everything.
size_t pool_cap(const Pool &pl) {
    if (pl.cap_override >= 0) return (size_t)pl.cap_override;
    static const size_t dflt = [] {
        if (const char *e = std::getenv("RDR_POOL_CAP_MB")) return (size_t)std::max(0, std::atoi(e)) << 20;
        const size_t eight = (size_t)8192 << 20;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return eight; }
        return std::min(total_b / 4, eight);
    }();
    return dflt;
}
}

size_t pool_cap_bytes() { Pool &pl = pool(); std::lock_guard<std::mutex> lk(pl.lock); return pool_cap(pl); }

void pool_set_cap(long long bytes) {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.lock);
    pl.cap_override = bytes;
}

// RDR_POOL_POISON=1 (debugging): every block handed out is filled with 0xFF bytes (NaN as a double or float, -1 as an int) on
// the calling thread's stream -- a buffer that is read before it is written then fails the same way every time instead of
// depending on what the pool last kept in it (fresh device memory is zero; the CPU harness gets zero pages from malloc).
static void *pool_alloc_raw(size_t bytes);
static void pool_release(int only_dev);
void *pool_alloc(size_t bytes) {
    void *p = pool_alloc_raw(bytes);
    static const bool poison = std::getenv("RDR_POOL_POISON") != nullptr;
    if (poison) check(hipMemsetAsync(p, 0xFF, std::max<size_t>(bytes, 16), ctx().stream), "pool poison");
    return p;
}
static void *pool_alloc_raw(size_t bytes) {
    const size_t want = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Pool &pl = pool();
    {
        std::lock_guard<std::mutex> lk(pl.lock);
        auto &fl = pl.free_blocks[dev & 15];
        auto it = fl.lower_bound(want);
        if (it != fl.end() && it->first <= want + want / 4 + 4096) {          // close fit: reuse
            void *p = it->second;
            pl.live[p] = {it->first, dev};
            pl.parked[dev & 15] -= it->first;
            fl.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {             // out of memory with blocks parked in the cache: release them and retry once
        (void)hipGetLastError();
        pool_release(dev);             // this device's parked blocks only: the other devices' callers are not disturbed
        check(hipMalloc(&p, want), "hipMalloc");
    }
    std::lock_guard<std::mutex> lk(pl.lock);
    pl.device_mallocs++;
    pl.from_driver[dev & 15] += want;
    pl.live[p] = {want, dev};
    return p;
}

void pool_free(void *p) {
    if (!p) return;
    Pool &pl = pool();
    std::unique_lock<std::mutex> lk(pl.lock);
    auto it = pl.live.find(p);
    if (it == pl.live.end()) { lk.unlock(); (void)hipFree(p); return; }
    const size_t bytes = it->second.first;
    const int d = it->second.second & 15;
    pl.live.erase(it);
    if (pl.parked[d] + bytes > pool_cap(pl)) {
        pl.from_driver[d] -= bytes;
        lk.unlock();
        (void)hipFree(p);              // waits for the device: safe whatever is in flight; outside the lock: the other workers go on
        return;
    }
    pl.free_blocks[d].emplace(bytes, p);
    pl.parked[d] += bytes;
}

// Parked blocks of ONE device (-1: of every device) back to the driver.  The blocks are taken off the lists under the pool's
// lock and released outside it: hipFree waits for the device, and another device's allocations must not wait with it
// (one process driving several devices from several threads).
static void pool_release(int only_dev) {
    Pool &pl = pool();
    std::vector<std::pair<int, void *>> gone;
    {
        std::lock_guard<std::mutex> lk(pl.lock);
        for (int d = 0; d < 16; ++d) {
            if (only_dev >= 0 && d != (only_dev & 15)) continue;
            auto &fl = pl.free_blocks[d];
            for (auto &kv : fl) gone.push_back({d, kv.second});
            fl.clear();
            pl.from_driver[d] -= pl.parked[d];
            pl.parked[d] = 0;
        }
    }
    if (gone.empty()) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int at = dev;
    for (auto &g : gone) {
        if (g.first != at) { (void)hipSetDevice(g.first); at = g.first; }
        (void)hipFree(g.second);
    }
    if (at != dev) (void)hipSetDevice(dev);
}
void pool_trim() { pool_release(-1); }

size_t pool_cached_bytes() {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.lock);
    size_t total = 0;
    for (size_t b : pl.parked) total += b;
    return total;
}

size_t memory_available() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return ~(size_t)0; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    Pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.lock);
    return free_b + pl.parked[dev & 15];
}

double memory_held_by_others() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); return 0.0; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    Pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.lock);
    const double used = (double)(total_b - free_b), mine = (double)pl.from_driver[dev & 15];
    return used > mine ? (used - mine) / (double)total_b : 0.0;
}

size_t pool_device_mallocs() { Pool &pl = pool(); std::lock_guard<std::mutex> lk(pl.lock); return pl.device_mallocs; }

namespace {
struct Staging { char *base = nullptr; size_t cap = 0, used = 0; };
Staging &staging() { static thread_local Staging s; return s; }
char *staging_reserve(size_t bytes) {          // 256-byte aligned slice of the pinned buffer; grows when idle
    Staging &st = staging();
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (st.used + need > st.cap) {
        sync();                                 // nothing in flight may still read the old buffer
        st.used = 0;
        if (need > st.cap) {
            if (st.base) (void)hipHostFree(st.base);
            st.cap = std::max<size_t>(need * 2, (size_t)32 << 20);
            check(hipHostMalloc((void **)&st.base, st.cap, hipHostMallocDefault), "hipHostMalloc (staging)");
        }
    }
    char *p = st.base + st.used;
    st.used += need;
    return p;
}
}

void upload_async(void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    char *stage = staging_reserve(bytes);
    std::memcpy(stage, src, bytes);
    check(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, ctx().stream), "upload_async");
}
void upload_flush() { sync(); staging().used = 0; }

void download_batch(const DownloadItem *items, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (items[i].bytes + 255) & ~(size_t)255;
    if (!total) return;
    upload_flush();                              // the staging buffer is shared with queued uploads
    char *stage = staging_reserve(total);
    size_t at = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].bytes) check(hipMemcpyAsync(stage + at, items[i].src, items[i].bytes, hipMemcpyDeviceToHost, ctx().stream), "download_batch");
        at += (items[i].bytes + 255) & ~(size_t)255;
    }
    sync();
    at = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].bytes) std::memcpy(items[i].dst, stage + at, items[i].bytes);
        at += (items[i].bytes + 255) & ~(size_t)255;
    }
    staging().used = 0;
}

const void *device_constant(const void *host, size_t bytes) {
    static std::mutex lock;
    static std::map<std::pair<const void *, int>, void *> table;      // (host table, device) -> device copy, kept for the process
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(lock);
    void *&p = table[{host, dev}];
    if (!p) {
        p = dmalloc(bytes);
        check(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice), "device_constant");
    }
    return p;
}

CompactScratch &compact_scratch(int nblocks, int which) {
    static thread_local CompactScratch per_device[16][2];
    int dev = 0;
    (void)hipGetDevice(&dev);
    CompactScratch &s = per_device[dev & 15][which & 1];
    if (s.capacity < nblocks) {
        if (s.block_counts) (void)hipFree(s.block_counts);
        if (!s.total) {
            void *h = nullptr, *d = nullptr;
            check(hipHostMalloc(&h, 64, hipHostMallocMapped), "hipHostMalloc");
            check(hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer");
            s.total_host = (volatile int *)h;
            s.total = (int *)d;
            s.total_host[0] = 0; s.total_host[1] = 0;       // [count, ticket of the compaction that wrote it]
        }
        s.capacity = nblocks + 1024;
        s.block_counts = (int *)dmalloc(sizeof(int) * s.capacity);
    }
    return s;
}

// STACK: entries of the per-lane LDS stack column.  The kernel waits on node fetches about two thirds of the
// time (profiles/r1_notes.md), so waves per SIMD matter: 40 entries allow 4, 24 allow 6, 16 allow 8.  The host
// picks the smallest instantiation that covers the scene's hierarchy depth.
constexpr int kTopNodes = 255;          // root + 127 sibling pairs (pairs start at odd indices, so none straddles): 8 KiB of LDS
// node record i: from the workgroup's LDS copy of the top levels, else from global memory.  The LDS pointer keeps its address
// space in its type: with two generic pointers the compiler merges the paths into a select + flat_load.
typedef const __attribute__((address_space(3))) float *LdsFloats;
struct FetchStaged {
    const rt::Node *nodes; LdsFloats top; int ntop;
    // the two children of an inner record: adjacent records that start at an odd index, so both lie in the LDS copy or neither
    // does -- ONE branch for the pair (two operator() calls cost two, each with both paths compiled in)
    __device__ void pair(int i, rt::Node &l, rt::Node &r) const {
        if (i < ntop) {
            LdsFloats p = top + 8 * i;
            l.lo[0] = p[0]; l.lo[1] = p[1]; l.lo[2] = p[2]; l.a = __float_as_int(p[3]);
            l.hi[0] = p[4]; l.hi[1] = p[5]; l.hi[2] = p[6]; l.b = __float_as_int(p[7]);
            r.lo[0] = p[8]; r.lo[1] = p[9]; r.lo[2] = p[10]; r.a = __float_as_int(p[11]);
            r.hi[0] = p[12]; r.hi[1] = p[13]; r.hi[2] = p[14]; r.b = __float_as_int(p[15]);
        } else { l = nodes[i]; r = nodes[i + 1]; }
    }
    __device__ rt::Node operator()(int i) const {
        rt::Node n;
        if (i < ntop) {
            LdsFloats p = top + 8 * i;
            n.lo[0] = p[0]; n.lo[1] = p[1]; n.lo[2] = p[2]; n.a = __float_as_int(p[3]);
            n.hi[0] = p[4]; n.hi[1] = p[5]; n.hi[2] = p[6]; n.b = __float_as_int(p[7]);
        } else n = nodes[i];
        return n;
    }
};
// Tallies of the counting variants (untimed roofline pass of bench.py): summed across the wave first -- one atomic per wave and
// counter instead of one per lane (15 M lanes x 3 counters on one line cost 13-15 ms per launch, profiles/r3_notes.md).
__device__ inline void wave_tally(unsigned long long *p, unsigned long long v) {
    const unsigned long long act = __ballot(1);
    double s;
    if (act == ~0ull) s = rdr::wave_sum((double)v);          // exact: the tallies stay far below 2^53
    else {
        s = 0;
        const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
        for (unsigned long long m = act; m; m &= m - 1) {
            const int k = __ffsll((long long)m) - 1;
            s += (double)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, k) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, k));
        }
    }
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(p, (unsigned long long)s);
}

template <bool ANY, bool COUNT, int STACK, class IDX, bool TOP>
__global__ void __launch_bounds__(256) trace_kernel(rt::BvhD bvh, const rt::RayRec *__restrict__ rays,
                                                    rt::HitRec *__restrict__ hits, int n, const int *count,
                                                    unsigned long long *counters) {
    if (count) { const int c = *count; n = c < n ? c : n; }
    if ((int)(blockIdx.x * 256) >= n) return;                 // the whole workgroup lies beyond the queue (grids are sized by an upper bound)
    __shared__ IDX stack_tile[STACK * 256];                   // per-lane stack columns (16-bit when the node count allows)
    IDX *stack = stack_tile + threadIdx.x;
    // the top of the hierarchy (breadth-first order: the first records are its upper levels), staged once per workgroup
    __shared__ rt::Node top[kTopNodes + 1];
    const int ntop = TOP ? (bvh.num_nodes < kTopNodes ? bvh.num_nodes : kTopNodes) : 0;
    if (TOP) {
        if ((int)threadIdx.x < ntop) top[threadIdx.x] = bvh.nodes[threadIdx.x];
        __syncthreads();
    }
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    rt::RayRec r = rays[i];
    rt::Hit h{0.f, -1, -1};
    rt::Counters c{0, 0};
    if (!(r.tmax < 0.f)) {
        float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
        if (COUNT) {
            h = TOP ? rt::traverse_with<ANY, IDX>(bvh, o, d, r.tmin, r.tmax, stack, 256, &c, FetchStaged{bvh.nodes, (LdsFloats)(const float *)top, ntop})
                    : rt::traverse<ANY, IDX>(bvh, o, d, r.tmin, r.tmax, stack, 256, &c);
        } else {
            h = TOP ? rt::traverse_with<ANY, IDX>(bvh, o, d, r.tmin, r.tmax, stack, 256, (rt::Counters *)nullptr, FetchStaged{bvh.nodes, (LdsFloats)(const float *)top, ntop})
                    : rt::traverse<ANY, IDX>(bvh, o, d, r.tmin, r.tmax, stack, 256, nullptr);
        }
    }
    if (COUNT) {                                                // (the lanes of the wave that hold a queue slot are all here)
        wave_tally(&counters[ANY ? 3 : 4], 1ull);               // queue slots (base is g_counters, +2 for any-hit): [4] / [5]
        wave_tally(&counters[0], c.nodes);
        wave_tally(&counters[1], c.tris);
    }
    hits[i] = rt::HitRec{h.shape, h.shape >= 0 ? h.prim : -1};
}

// ---- the walk over the 4-wide records (bvh.h: Node4) ---------------------------------------------------------------------
// One ray per lane; a step fetches ONE 128-byte record (eight 16-byte loads issued together, one dependent round trip),
// tests its four child boxes, orders the children that are hit by entry distance (a five-exchange network), pushes the
// farther ones and descends into the nearest; a leaf child carries its triangle range in the link, so its triangles are
// tested without another fetch.  Stack entries are links (32 bit) in an LDS column.  Hits are decided by rt::ray_triangle /
// rt::closer alone (raytri.h), so the result equals the binary walk's and the brute-force rule's whatever the order.
template <bool ANY, bool COUNT, int STACK>
__global__ void __launch_bounds__(256) trace_wide_kernel(rt::BvhD bvh, const rt::RayRec *__restrict__ rays, rt::HitRec *__restrict__ hits,
                                                         int n, const int *count, unsigned long long *counters) {
    if (count) { const int c = *count; n = c < n ? c : n; }
    if ((int)(blockIdx.x * 256) >= n) return;
    __shared__ int stack_tile[STACK * 256];
    int *stack = stack_tile + threadIdx.x;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const rt::RayRec r = rays[i];
    rt::Hit h{0.f, -1, -1};
    rt::Counters c{0, 0};
    if (!(r.tmax < 0.f)) {
        const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
        h = rt::traverse_wide<ANY>(bvh, o, d, r.tmin, r.tmax, stack, 256, COUNT ? &c : nullptr);
    }
    // counters = base of this query kind ({nodes, tris} at [0, 1]); the 4-wide records are tallied at g_counters[6] / [7]
    if (COUNT) { wave_tally(&counters[ANY ? 3 : 4], 1ull); wave_tally(&counters[ANY ? 5 : 6], c.nodes); wave_tally(&counters[1], c.tris); }
    hits[i] = rt::HitRec{h.shape, h.shape >= 0 ? h.prim : -1};
}

// ---- lanes that take the next ray when theirs is done -----------------------------------------------------------------------
// A wave owns 64 x K consecutive rays of the queue.  A lane whose ray is finished idles only until `idle_min` lanes of the wave
// are idle; then the idle lanes take the next unclaimed rays of the wave's chunk (a ballot and a popcount: no atomics, no
// barrier, nothing moves -- the stack column of a lane is reused by its next ray).  Between two such checks every live lane
// advances `steps` steps.  Every ray takes the same steps in the same order as in the plain kernel; hits are written by ray.
// Measured (tools/trace_ab.py 2048: queues of 2.6-4.2 M rays, profiles/r3_notes.md), 4 rays per lane / 24 idle lanes / 4 steps:
// closest-hit on bounce rays 0.948 -> 0.818 ms and 0.750 -> 0.629 ms (-14 ... -16 %), any-hit on bounce rays -13 % / -2 %; on
// COHERENT queues (camera rays, their shadow rays) +25 % / +19 % -- the lanes of a wave start neighbours and finish together,
// refilling only mixes them; and on a million rays there are too few waves left to fill the GPU (+13 % with two rays per lane,
// +28 % with four).  So: queues sized for >= 2^22 lanes that the caller does not mark coherent, four rays per lane; in the
// 1024 x 1024 benchmark 60.4 -> 62.8 (closest-hit queues) -> 63.5 Msamples/s (shadow-ray queues of bounce vertices too);
// vector-ALU lane utilisation of these launches 0.19 -> 0.28.  (A vote per step on top -- lanes on leaves park, a step of the
// wave is either the box body or the triangle body -- was measured as well: 8-10 % slower than refilling alone, whatever the
// number of parked lanes it waits for.)
// LDSN > 0 (hybrid stack, big hierarchies): the first LDSN entries of a lane's stack are its LDS column, deeper ones go to a
// private array (scratch) -- the walk is at depth < 16 nearly always, and 16 int entries instead of 32 / 40 let six workgroups
// share a CU's LDS where four / three did: a hierarchy beyond the L2 is latency-bound and wants the waves (profiles/r6_notes.md).
constexpr int kHybridLds = 16, kHybridSpill = rt::kTraverseStack - kHybridLds;
template <bool ANY, class IDX, int LDSN, class Fetch>
__device__ inline bool traverse_some(const rt::BvhD &bvh, const float o[3], const float d[3], const float inv[3], float tnear, float tfar,
                                     IDX *stack, int (&spill)[LDSN > 0 ? kHybridSpill : 1], rt::Hit &best, int &cur, int &sp, int budget, const Fetch &fetch) {
    using namespace rt;
    for (int it = 0; it < budget; ++it) {
        const Node n = fetch(cur);
        if (n.b > 0) {
            for (int k = 0; k < n.b; ++k) {
                const int slot = n.a + k;
                const float *t = bvh.tris + 9 * slot;
                float th;
                if (ray_triangle(o, d, tnear, tfar, t, t + 3, t + 6, &th)) {
                    const int s = bvh.ids[2 * slot], p = bvh.ids[2 * slot + 1];
                    if (ANY) { best = Hit{th, s, p}; return true; }
                    if (closer(th, s, p, best)) best = Hit{th, s, p};
                }
            }
        } else {
            Node l, r;
            fetch.pair(n.a, l, r);
            const float lim = best.shape < 0 ? tfar : best.t * 1.0000004f + 1e-30f;     // closed at best.t: equal-t candidates are still visited
            float tl, tr;
            const bool hl = ray_box_once(o, inv, tnear, lim, l.lo, l.hi, &tl);
            const bool hr = ray_box_once(o, inv, tnear, lim, r.lo, r.hi, &tr);
            if (hl && hr) {
                int near = n.a, far = n.a + 1;
                if (tr < tl) { near = n.a + 1; far = n.a; }
                if (LDSN == 0 || sp < LDSN) stack[sp * 256] = (IDX)far; else spill[sp - LDSN] = far;
                ++sp;
                cur = near;
                continue;
            } else if (hl) { cur = n.a; continue; }
            else if (hr) { cur = n.a + 1; continue; }
        }
        if (sp == 0) return true;
        --sp;
        cur = (LDSN == 0 || sp < LDSN) ? (int)stack[sp * 256] : spill[sp - LDSN];
    }
    return false;
}

// `sort_mode` (round 6): the wave hands its 256 rays out in the order of a direction key instead of queue
// order -- 1: octant of the direction (Gray-code order: neighbouring bins differ in one sign), 2: octant x dominant axis
// (24 bins), dead slots (tmax < 0) last.  A counting sort of the 256 indices by ballots, one byte per index in LDS; nothing
// else moves: rays are read and hits written by queue slot, every ray takes the same steps, so the hit ids are the same.
template <bool ANY, int STACK, class IDX, bool SORT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STACK <= 16 ? 6 : (STACK <= 24 ? 7 : (STACK <= 32 ? 4 : 1)), 8))) trace_refill_kernel(rt::BvhD bvh, const rt::RayRec *__restrict__ rays, rt::HitRec *__restrict__ hits,
                                                           int n, const int *count, int rays_per_lane, int idle_min, int steps, int sort_mode) {
    if (count) { const int c = *count; n = c < n ? c : n; }
    const int chunk = 64 * rays_per_lane;
    if ((long long)blockIdx.x * 4 * chunk >= n) return;
    __shared__ IDX stack_tile[STACK * 256];
    IDX *stack = stack_tile + threadIdx.x;
    // (32-entry int stacks + the order bytes: 31 fewer pairs of the top staged, so that four workgroups still fit a CU's LDS)
    constexpr int kTop = (SORT && STACK == 32) ? 191 : kTopNodes;
    __shared__ rt::Node top[kTop + 1];
    __shared__ unsigned char order_tile[SORT ? 4 * 256 : 4];
    const int ntop = bvh.num_nodes < kTop ? bvh.num_nodes : kTop;
    if ((int)threadIdx.x < ntop) top[threadIdx.x] = bvh.nodes[threadIdx.x];
    __syncthreads();
    const FetchStaged fetch{bvh.nodes, (LdsFloats)(const float *)top, ntop};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long first = ((long long)blockIdx.x * 4 + wave) * chunk;
    if (first >= n) return;
    const int end = (int)(first + chunk < n ? first + chunk : n);
    unsigned char *order = order_tile + wave * 256;
    const bool sorted = SORT && sort_mode > 0 && rays_per_lane == 4;
    if (sorted) {
        const unsigned long long below = (1ull << lane) - 1ull;
        unsigned keys = 0;                                  // byte j: key of ray first + 64 j + lane (255: beyond the queue)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long r = first + 64 * j + lane;
            unsigned key = 255u;
            if (r < end) {
                const rt::RayRec ray = rays[r];
                if (ray.tmax < 0.f) key = 254u;             // dead slot: handed out last (it finishes at once)
                else {
                    const unsigned oct = (ray.dx < 0.f ? 1u : 0u) | (ray.dy < 0.f ? 2u : 0u) | (ray.dz < 0.f ? 4u : 0u);
                    const unsigned gray = oct ^ (oct >> 1);
                    key = gray;
                    if (sort_mode >= 2) {
                        const float ax = fabsf(ray.dx), ay = fabsf(ray.dy), az = fabsf(ray.dz);
                        const unsigned axis = ax >= ay ? (ax >= az ? 0u : 2u) : (ay >= az ? 1u : 2u);
                        key = gray * 3u + ((gray & 1u) ? 2u - axis : axis);       // boustrophedon: the axis order turns at every octant
                    }
                }
            }
            keys |= key << (8 * j);
        }
        const int nbins = sort_mode >= 2 ? 24 : 8;
        int off = 0;
        for (int b = 0; b <= nbins; ++b) {                  // bin `nbins` = the dead slots
            const unsigned want = b == nbins ? 254u : (unsigned)b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = ((keys >> (8 * j)) & 255u) == want;
                const unsigned long long m = __ballot(in);
                if (in) order[off + __popcll(m & below)] = (unsigned char)(64 * j + lane);
                off += __popcll(m);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    int next = (int)first;                                  // wave-uniform
    bool live = false;
    int slot = 0, cur = 0, sp = 0;
    rt::Hit best{0.f, -1, -1};
    float o[3] = {0, 0, 0}, d[3] = {1, 1, 1}, inv[3] = {1, 1, 1}, tmin = 0, tmax = -1;
    constexpr int kLdsN = (STACK == kHybridLds && sizeof(IDX) == 4) ? kHybridLds : 0;       // 16 int entries: the hybrid stack
    int spill[kLdsN > 0 ? kHybridSpill : 1];
    for (;;) {
        const unsigned long long idle = __ballot(!live);
        const int nidle = __popcll(idle);
        if (next < end && (nidle >= idle_min || nidle == 64)) {
            if (!live) {
                int r = next + __popcll(idle & ((1ull << lane) - 1ull));
                if (r < end) {
                    if (SORT && sorted) r = (int)first + (int)order[r - (int)first];
                    const rt::RayRec ray = rays[r];
                    slot = r; cur = 0; sp = 0;
                    o[0] = ray.ox; o[1] = ray.oy; o[2] = ray.oz; d[0] = ray.dx; d[1] = ray.dy; d[2] = ray.dz;
                    inv[0] = 1.f / d[0]; inv[1] = 1.f / d[1]; inv[2] = 1.f / d[2];
                    tmin = ray.tmin; tmax = ray.tmax;
                    best = rt::Hit{tmax, -1, -1};
                    bool miss = tmax < 0.f || bvh.num_nodes == 0;
                    if (!miss) { float tn; const rt::Node root = fetch(0); miss = !rt::ray_box(o, inv, tmin, tmax, root.lo, root.hi, &tn); }
                    if (miss) hits[r] = rt::HitRec{-1, -1}; else live = true;
                }
            }
            next += nidle;
        }
        if (__ballot(live) == 0ull) { if (next >= end) break; continue; }
        if (live) {
            if (traverse_some<ANY, IDX, kLdsN>(bvh, o, d, inv, tmin, tmax, stack, spill, best, cur, sp, steps, fetch)) {
                hits[slot] = rt::HitRec{best.shape, best.shape >= 0 ? best.prim : -1};
                live = false;
            }
        }
    }
}

hipStream_t side_stream(int k) {
    // k = 0, 1: two non-blocking streams; k = 2, 3: two more at the LOWEST priority.  Where a kernel of the calling stream (a
    // sample's critical path: the continuation-ray traversal, the bounce adjoints) and one of a low-priority side stream
    // (shadow rays, edge picks) both have workgroups waiting, the calling stream's are dispatched first: closest-hit launch
    // 0.334 -> 0.307 ms in the 1024 x 1024 benchmark with the shadow-ray launch beside it, throughput +0.3 %.  Small renders
    // (one wave per SIMD per launch, several samples and the edge builder in flight) lose 20 % that way: render.cpp asks for
    // the low-priority pair only for large frames (RDR_NO_SIDE_PRIORITY=1: never).
    static thread_local hipStream_t streams[16][4] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipStream_t &s = streams[dev & 15][k & 3];
    if (!s) {
        static const bool allowed = std::getenv("RDR_NO_SIDE_PRIORITY") == nullptr;
        int least = 0, greatest = 0;
        if ((k & 2) && allowed && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            check(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least), "hipStreamCreateWithPriority");
        else
            check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    }
    return s;
}

int *new_count() {
    static thread_local int *rings[16] = {};
    static thread_local int slots[16] = {};
    constexpr int kRing = 8192;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int *&ring = rings[dev & 15];
    int &slot = slots[dev & 15];
    if (!ring) ring = (int *)dmalloc(sizeof(int) * kRing);
    int *p = ring + slot;
    slot = (slot + 1) % kRing;
    return p;
}

int *persistent_counter() {
    // per host thread and per device; the slot is zeroed ON THE STREAM OF THE LAUNCH that is about to use it.  (Rounds 3-5 zeroed
    // the whole ring once per lap on whatever stream was current then: a kernel launched on ANOTHER stream right after the
    // ring's allocation could read its counter before that fill had run -- harmless while the only user was the gated overflow
    // walk, a wrong pick once in a while as soon as the hierarchical pick's descent took its chunks off such a counter; found as
    // an intermittent failure of the first test after a library load, round 6.)
    static thread_local int *rings[16] = {};
    static thread_local int slots[16] = {};
    constexpr int kRing = 4096;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int *&ring = rings[dev & 15];
    int &slot = slots[dev & 15];
    if (!ring) ring = (int *)dmalloc(sizeof(int) * kRing);
    int *p = ring + slot;
    slot = (slot + 1) % kRing;
    check(hipMemsetAsync(p, 0, sizeof(int), ctx().stream), "persistent_counter");
    return p;
}

namespace {
struct Pending { hipEvent_t a, b; bool any; };
std::mutex g_stats_lock;                 // launches come from every sample worker's host thread
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;
unsigned long long *g_counters = nullptr;

hipEvent_t get_event() {
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e;
    check(hipEventCreate(&e), "hipEventCreate");
    return e;
}
}

TraceStats &trace_stats() { static TraceStats s; return s; }

void trace_stats_collect() {           // call between render() calls: every worker's stream has been joined by then
    TraceStats &st = trace_stats();
    std::lock_guard<std::mutex> lk(g_stats_lock);
    if (!g_pending.empty()) {
        check(hipDeviceSynchronize(), "stats sync");
        // Per launch: its duration (event pair on the launch stream).  Per kind: the UNION of the launches' busy intervals --
        // the time during which at least one launch of the kind was in flight.  With one chain of launches the two agree; with
        // two sample workers two closest-hit launches share the GPU, each takes longer and the same rays are traced at the same
        // total rate: bytes over the union is the schedule-invariant rate (bench.py: roofline.frac).  Interval ends are placed
        // relative to the first launch's start event (same device clock on every stream).
        std::vector<std::pair<double, double>> iv[2];
        const hipEvent_t ref = g_pending.front().a;
        for (Pending &p : g_pending) {
            float ms = 0, at = 0;
            check(hipEventElapsedTime(&ms, p.a, p.b), "hipEventElapsedTime");
            (p.any ? st.any_ms : st.closest_ms) += ms;
            if (p.a != ref && hipEventElapsedTime(&at, ref, p.a) != hipSuccess) { (void)hipGetLastError(); at = 0.f; }
            iv[p.any ? 1 : 0].push_back({(double)at, (double)at + (double)ms});
        }
        for (int k = 0; k < 2; ++k) {
            std::sort(iv[k].begin(), iv[k].end());
            double busy = 0, lo = 0, hi = -1;
            for (const auto &x : iv[k]) {
                if (hi < lo || x.first > hi) { if (hi >= lo) busy += hi - lo; lo = x.first; hi = x.second; }
                else if (x.second > hi) hi = x.second;
            }
            if (hi >= lo) busy += hi - lo;
            (k ? st.any_union_ms : st.closest_union_ms) += busy;
        }
        for (Pending &p : g_pending) { g_free_events.push_back(p.a); g_free_events.push_back(p.b); }
        g_pending.clear();
    }
    if (g_counters) {
        unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        download(c, g_counters, sizeof(c));
        st.nodes[0] += c[0]; st.tris[0] += c[1]; st.nodes[1] += c[2]; st.tris[1] += c[3];
        st.wide_nodes[0] += c[6]; st.wide_nodes[1] += c[7];
        st.closest_rays += c[4]; st.any_rays += c[5];          // queue lengths live on the device: the counting kernels tally them
        zero(g_counters, sizeof(c));
        sync();
    }
}

void trace(const rt::BvhD &bvh, const rt::RayRec *rays, rt::HitRec *hits, Count cnt, bool any, bool coherent) {
    const int n = cnt.upper;
    const int *n_dev = cnt.dev;
    if (n <= 0) return;
    TraceStats &st = trace_stats();
    hipStream_t s = ctx().stream;
    int blocks = (n + 255) / 256;
    Pending p{};
    if (st.timing) {
        { std::lock_guard<std::mutex> lk(g_stats_lock); p.a = get_event(); p.b = get_event(); }
        p.any = any;
        check(hipEventRecord(p.a, s), "hipEventRecord");
    }
    // Staging pays on big queues (closest-hit 0.330 -> 0.321 ms per 956 k rays); on a 256 x 256 frame the 8 KiB copy + barrier per
    // 256 rays costs more than the L1-hot top levels save (optimisation-loop iteration +2 ms).  RDR_TUNE_TRACE_NO_LDS_TOP: never.
    const rdr::Tuning &tune = rdr::tuning();
    const bool stage_top = !tune.has(RDR_TUNE_TRACE_NO_LDS_TOP) && n >= (1 << 18);
    // Which form of the hierarchy: queues of up to RDR_WIDE_MAX rays (default 2^19) walk the 4-wide records (measured,
    // tools/trace_ab.py, profiles/r3_notes.md: half the dependent steps per ray pays where a launch is one or two waves per
    // SIMD -- closest-hit 0.119 -> 0.102 ms, any-hit 0.078 -> 0.066 ms per 65 k / 50 k rays; on queues of a million rays and
    // more both forms issue the same number of vector instructions per wave and the binary records, at 8 instead of 5 waves
    // per SIMD, are 0-10 % ahead).  RDR_TUNE_TRACE_BINARY: never the wide records.
    const bool wide_allowed = !tune.has(RDR_TUNE_TRACE_BINARY);
    const int wide_max = tune.wide_max;
    if (wide_allowed && bvh.wide != nullptr && bvh.wide_stack_need <= 48 && n <= wide_max) {
        unsigned long long *ctr = nullptr;
        if (st.counting) {
            std::lock_guard<std::mutex> lk(g_stats_lock);
            if (!g_counters) {
                g_counters = (unsigned long long *)dmalloc(64);
                check(hipMemset(g_counters, 0, 64), "hipMemset");
            }
            ctr = any ? g_counters + 2 : g_counters;
        }
#define RDR_WIDE_LAUNCH(ANY_, COUNT_, STACK_) \
        hipLaunchKernelGGL((trace_wide_kernel<ANY_, COUNT_, STACK_>), dim3(blocks), dim3(256), 0, s, bvh, rays, hits, n, n_dev, ctr)
        const int wneed = bvh.wide_stack_need;
#define RDR_WIDE_BY_STACK(ANY_, COUNT_)                                        \
        do {                                                                   \
            if (wneed <= 12) RDR_WIDE_LAUNCH(ANY_, COUNT_, 12);  \
            else if (wneed <= 16) RDR_WIDE_LAUNCH(ANY_, COUNT_, 16); \
            else if (wneed <= 20) RDR_WIDE_LAUNCH(ANY_, COUNT_, 20); \
            else if (wneed <= 24) RDR_WIDE_LAUNCH(ANY_, COUNT_, 24); \
            else if (wneed <= 32) RDR_WIDE_LAUNCH(ANY_, COUNT_, 32); \
            else RDR_WIDE_LAUNCH(ANY_, COUNT_, 48);                            \
        } while (0)
        if (st.counting) { if (any) RDR_WIDE_BY_STACK(true, true); else RDR_WIDE_BY_STACK(false, true); }
        else { if (any) RDR_WIDE_BY_STACK(true, false); else RDR_WIDE_BY_STACK(false, false); }
#undef RDR_WIDE_BY_STACK
#undef RDR_WIDE_LAUNCH
        check(hipGetLastError(), "trace launch");
        if (st.timing) check(hipEventRecord(p.b, s), "hipEventRecord");
        std::lock_guard<std::mutex> lk(g_stats_lock);
        if (st.timing) g_pending.push_back(p);
        (any ? st.any_launches : st.closest_launches)++;
        if (!st.counting) (any ? st.any_rays : st.closest_rays) += (uint64_t)n;
        return;
    }
    // lanes refilled from the wave's own chunk of the queue (see trace_refill_kernel): queues sized for >= 2^22 lanes that the
    // caller does not mark coherent.  (The queue's host-side bound decides: a launch sized for 2^22 lanes -- four samples of a
    // 1024 x 1024 frame, the edge sub-paths' two lanes per slot -- still holds 1.5-3.3 M rays after the compactions; choosing the
    // rays per lane in the kernel from the actual count was measured too and is slower, 61.8 vs 62.5 Msamples/s.)
    // rdr_tuning: RDR_TUNE_REFILL_OFF never; refill_* those parameters; RDR_TUNE_REFILL_ALL every queue (tools/trace_ab.py).
    const bool refill_off = tune.has(RDR_TUNE_REFILL_OFF), refill_all = tune.has(RDR_TUNE_REFILL_ALL);
    const int refill_k = refill_off ? 0 : ((refill_all || (!coherent && n >= (1 << 22))) ? tune.refill_k : 0);
    if (refill_k >= 1 && !st.counting && bvh.stack_need <= rt::kTraverseStack) {
        const int idle_min = tune.refill_idle, steps = tune.refill_steps;
        const int wg_rays = 4 * 64 * refill_k;
        const int rblocks = (int)(((long long)n + wg_rays - 1) / wg_rays);
        const int k_arg = refill_k;
        const int sort_mode = tune.refill_sort;        // rdr_tuning::refill_order: 0 queue order, 1 octant (default), 2 octant x axis
        const bool small = bvh.num_nodes < 65536 && bvh.stack_need <= 24;
        // big hierarchies (int entries): 32 entries where that covers the tree -- 40 KiB of LDS per workgroup instead of 49: four
        // workgroups per CU instead of three (a hierarchy beyond the L2 is latency-bound: more waves, profiles/r6_notes.md)
        const bool mid32 = !small && bvh.stack_need <= 32;
        // ... and the hybrid stack (16 LDS entries + scratch: six workgroups per CU) for every hierarchy too big for the 16-bit
        // column: 0.92 M triangles 2.39 -> 2.73, 3.7 M 2.02 -> 2.33 G rays/s against the 32-entry tier.  RDR_TRACE_HYBRID=0: the tiers.
        static const bool hybrid_on = [] { const char *e = std::getenv("RDR_TRACE_HYBRID"); return !(e && e[0] == '0'); }();
        const bool hybrid = !small && hybrid_on;
#define RDR_REFILL_LAUNCH(ANY_, STACK_, IDX_, SORT_) \
        hipLaunchKernelGGL((trace_refill_kernel<ANY_, STACK_, IDX_, SORT_>), dim3(rblocks), dim3(256), 0, s, bvh, rays, hits, n, n_dev, k_arg, idle_min, steps, sort_mode)
        const bool sort_on = sort_mode > 0 && k_arg == 4;
        if (sort_on) {
            if (any && small) RDR_REFILL_LAUNCH(true, 24, unsigned short, true);
            else if (any && hybrid) RDR_REFILL_LAUNCH(true, 16, int, true);
            else if (any && mid32) RDR_REFILL_LAUNCH(true, 32, int, true);
            else if (any) RDR_REFILL_LAUNCH(true, rt::kTraverseStack, int, true);
            else if (small) RDR_REFILL_LAUNCH(false, 24, unsigned short, true);
            else if (hybrid) RDR_REFILL_LAUNCH(false, 16, int, true);
            else if (mid32) RDR_REFILL_LAUNCH(false, 32, int, true);
            else RDR_REFILL_LAUNCH(false, rt::kTraverseStack, int, true);
        } else {
            if (any && small) RDR_REFILL_LAUNCH(true, 24, unsigned short, false);
            else if (any && hybrid) RDR_REFILL_LAUNCH(true, 16, int, false);
            else if (any && mid32) RDR_REFILL_LAUNCH(true, 32, int, false);
            else if (any) RDR_REFILL_LAUNCH(true, rt::kTraverseStack, int, false);
            else if (small) RDR_REFILL_LAUNCH(false, 24, unsigned short, false);
            else if (hybrid) RDR_REFILL_LAUNCH(false, 16, int, false);
            else if (mid32) RDR_REFILL_LAUNCH(false, 32, int, false);
            else RDR_REFILL_LAUNCH(false, rt::kTraverseStack, int, false);
        }
#undef RDR_REFILL_LAUNCH
        check(hipGetLastError(), "trace launch");
        if (st.timing) check(hipEventRecord(p.b, s), "hipEventRecord");
        std::lock_guard<std::mutex> lk(g_stats_lock);
        if (st.timing) g_pending.push_back(p);
        (any ? st.any_launches : st.closest_launches)++;
        (any ? st.any_rays : st.closest_rays) += (uint64_t)n;
        return;
    }
#define RDR_TRACE_LAUNCH(ANY_, COUNT_, STACK_, ctr)                                                                          \
    do {                                                                                                                     \
        if (bvh.num_nodes < 65536 && stage_top)                                                                              \
            hipLaunchKernelGGL((trace_kernel<ANY_, COUNT_, STACK_, unsigned short, true>), dim3(blocks), dim3(256), 0, s, bvh, rays, hits, n, n_dev, ctr); \
        else if (bvh.num_nodes < 65536)                                                                                      \
            hipLaunchKernelGGL((trace_kernel<ANY_, COUNT_, STACK_, unsigned short, false>), dim3(blocks), dim3(256), 0, s, bvh, rays, hits, n, n_dev, ctr); \
        else if (stage_top)                                                                                                  \
            hipLaunchKernelGGL((trace_kernel<ANY_, COUNT_, STACK_, int, true>), dim3(blocks), dim3(256), 0, s, bvh, rays, hits, n, n_dev, ctr);            \
        else                                                                                                                 \
            hipLaunchKernelGGL((trace_kernel<ANY_, COUNT_, STACK_, int, false>), dim3(blocks), dim3(256), 0, s, bvh, rays, hits, n, n_dev, ctr);            \
    } while (0)
#define RDR_TRACE_BY_STACK(ANY_, COUNT_, ctr)                                  \
    do {                                                                       \
        if (bvh.stack_need <= 16) RDR_TRACE_LAUNCH(ANY_, COUNT_, 16, ctr);      \
        else if (bvh.stack_need <= 24) RDR_TRACE_LAUNCH(ANY_, COUNT_, 24, ctr); \
        else if (bvh.stack_need <= 32) RDR_TRACE_LAUNCH(ANY_, COUNT_, 32, ctr); \
        else RDR_TRACE_LAUNCH(ANY_, COUNT_, rt::kTraverseStack, ctr);          \
    } while (0)
    if (st.counting) {
        {
            std::lock_guard<std::mutex> lk(g_stats_lock);
            if (!g_counters) {
                g_counters = (unsigned long long *)dmalloc(64);
                check(hipMemset(g_counters, 0, 64), "hipMemset");        // synchronous: another worker's launch may be next
            }
        }
        // counters: {nodes, tris} of this query type at [0, 1], its ray tally at [4] (closest) / [3] relative to base + 2 (any)
        if (any) RDR_TRACE_BY_STACK(true, true, g_counters + 2);
        else RDR_TRACE_BY_STACK(false, true, g_counters);
    } else {
        if (any) RDR_TRACE_BY_STACK(true, false, (unsigned long long *)nullptr);
        else RDR_TRACE_BY_STACK(false, false, (unsigned long long *)nullptr);
    }
#undef RDR_TRACE_BY_STACK
#undef RDR_TRACE_LAUNCH
    check(hipGetLastError(), "trace launch");
    if (st.timing) {
        check(hipEventRecord(p.b, s), "hipEventRecord");
    }
    std::lock_guard<std::mutex> lk(g_stats_lock);
    if (st.timing) g_pending.push_back(p);
    (any ? st.any_launches : st.closest_launches)++;
    if (!st.counting) (any ? st.any_rays : st.closest_rays) += (uint64_t)n;     // an upper bound; exact tallies come from the counting kernels
}

} // namespace exec
