// exec.h (host debugging harness) -- TEST INFRASTRUCTURE, not a product path.
//
// Single-threaded stand-in for redner_amd/csrc/hip/exec.h so that the *same* stage bodies and
// host driver can be stepped through with gdb/ASan and compared against the oracle in a container
// without a GPU.  It is compiled only into tests/hostsim/_build/, is never imported by
// redner_amd, and the product raises if the HIP library or a GPU is missing.
#pragma once
#include <typeinfo>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <string>

#define RDR_FN inline
#define RDR_INLINE_CALL
#define RDR_DEV_FN inline
#define RDR_STACK_DECL(T, name, N) T name[N]
#define RDR_STACK_AT(name, k) name[k]
#define RDR_WALK_STACK_MEMBER(T, name, N) T name[N];
#define RDR_WALK_STACK(st, T, name, N, TAG) ((st).name)
#define RDR_WALK_AT(stk, k) stk[k]
#define RDR_HOSTSIM 1

namespace rdr {
// RDR_HOSTSIM_REF_ORDER=1 (tests/test_accumulation_order.py): next to the fp64 accumulator of every SMALL tensor the harness
// keeps what the REFERENCE keeps -- a float that takes every addend as `float += (float)term` (src/atomic.h:43-141) -- in the
// order the harness issues the adds: sample by sample, stage by stage, lane by lane, which for the camera tensors is the order of
// the reference run on ONE thread (oracle/one_core.c; one add per lane in d_primary_intersection, then one per slot in
// compute_primary_edge_derivatives, src/camera.h:255-257,824-826).  GradStore::flush hands out the floats instead of the sums.
struct F32Shadow { double *base = nullptr; size_t count = 0; std::vector<float> acc; };
inline F32Shadow &f32_shadow() { static F32Shadow s; return s; }
inline void shadow_add(double *p, double v) {
    F32Shadow &s = f32_shadow();
    if (s.base && p >= s.base && p < s.base + s.count) s.acc[(size_t)(p - s.base)] += (float)v;
}
// The three places GradStore (render.cpp) tells its accumulator backend about: the small tier has been laid out; a small
// tensor is about to be folded into the caller's floats; the fold is done.  (hip/exec.h: empty bodies.)
inline void accumulators_laid_out(double *small_base, size_t small_stride) {
    f32_shadow() = F32Shadow();
    if (std::getenv("RDR_HOSTSIM_REF_ORDER") && small_stride) {
        f32_shadow().base = small_base; f32_shadow().count = small_stride;
        f32_shadow().acc.assign(small_stride, 0.f);
    }
}
inline void accumulator_before_fold(double *small_base, double *acc, size_t count) {     // reference-order mode: the floats, not the fp64 sums
    if (f32_shadow().base && f32_shadow().base == small_base && count <= 16)
        for (size_t i = 0; i < count; ++i) acc[i] = (double)f32_shadow().acc[(size_t)(acc - small_base) + i];
}
inline void accumulators_folded() { f32_shadow() = F32Shadow(); }
inline void accum(double *p, double v) { *p += v; shadow_add(p, v); }
inline void accum_plain(double *p, double v) { *p += v; shadow_add(p, v); }
inline void accum_triple(double *p, double x, double y, double z) { p[0] += x; shadow_add(p, x); p[1] += y; shadow_add(p + 1, y); p[2] += z; shadow_add(p + 2, z); }
inline void accum_texel_triple(double *p, double x, double y, double z) { accum_triple(p, x, y, z); }
inline void accum_texel(double *p, double v) { *p += v; shadow_add(p, v); }
inline void atomic_add_f64(double *p, double v) { *p += v; }
inline int atomic_fetch_add(int *p, int v) { int o = *p; *p += v; return o; }
}

namespace exec {
inline void *dmalloc(size_t bytes) { void *p = malloc(bytes ? bytes : 16); if (!p) throw std::bad_alloc(); return p; }
inline void dfree(void *p) { free(p); }
inline void *pool_alloc(size_t bytes) { return dmalloc(bytes); }
inline void pool_free(void *p) { free(p); }
inline void pool_trim() {}
inline size_t pool_device_mallocs() { return 0; }
inline size_t host_count_reads() { return 0; }
inline void zero(void *p, size_t bytes) { memset(p, 0, bytes); }
inline void upload(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
inline void download(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
inline void copy_dev(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
inline void sync() {}
constexpr bool kDeviceBvh = false;            // ... nor a device builder of the triangle hierarchy: bvh.cpp builds it
constexpr bool kDeviceEdgeTrees = false;      // the harness has no kernels: the host builder of edges.cpp builds the edge hierarchies
inline void device_sync() {}
inline size_t pool_cached_bytes() { return 0; }
inline size_t memory_available() { return ~(size_t)0; }
inline double memory_held_by_others() { return 0.0; }
inline void pool_set_cap(long long) {}
inline size_t pool_cap_bytes() { return ~(size_t)0; }
inline int current_device() { return 0; }
inline void upload_async(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
inline void upload_flush() {}
struct DownloadItem { void *dst; const void *src; size_t bytes; };
inline void download_batch(const DownloadItem *items, int n) { for (int i = 0; i < n; ++i) memcpy(items[i].dst, items[i].src, items[i].bytes); }
inline const void *device_constant(const void *host, size_t) { return host; }
}
typedef void *hipStream_t;       // the host driver names streams; the harness has one implicit stream
namespace exec {
struct Context { hipStream_t stream = nullptr; };
inline Context &ctx() { static Context c; return c; }
inline hipStream_t side_stream(int) { return nullptr; }
struct StreamScope { explicit StreamScope(hipStream_t) {} ~StreamScope() {} };
struct Fence { void after(hipStream_t) {} void gate(hipStream_t) {} };
inline int sample_workers(int, int, bool = false) { return 1; }
struct SecondThread {          // never used: sample_workers() == 1
    static SecondThread &get(int = 0) { static SecondThread t; return t; }
    template <class F> void start(F) {}
    void wait() {}
};
struct Count {                 // see hip/exec.h: `dev` is host memory here
    const int *dev; int upper;
    Count(int n) : dev(nullptr), upper(n) {}
    Count(const int *d, int u) : dev(d), upper(u) {}
    int value() const { return dev ? (*dev < upper ? *dev : upper) : upper; }
};
inline int *new_count() { static int ring[8192]; static int at = 0; at = (at + 1) % 8192; return ring + at; }
inline int read_count(Count c) { return c.value(); }
inline Count scaled_count(Count c, int k) { if (!c.dev) return Count(k * c.upper); int *d = new_count(); *d = k * c.value(); return Count(d, k * c.upper); }
// RDR_HOSTSIM_LANES=1: one line per stage launch (stage type, lanes) on stderr -- where the lanes of a frame go
inline void note_launch(const char *stage, int n) {
    static const bool on = std::getenv("RDR_HOSTSIM_LANES") != nullptr;
    if (on) std::fprintf(stderr, "[lanes] %d %s\n", n, stage);
}
template <class F>
inline void launch(Count c, const F &f) { const int n = c.value(); note_launch(typeid(F).name(), n); for (int i = 0; i < n; ++i) f(i); }
inline int *persistent_counter() { static int ring[64]; static int at = 0; at = (at + 1) % 64; ring[at] = 0; return ring + at; }
template <class W>
inline void launch_persistent(Count c, const W &w) {          // see hip/exec.h: begin / step... / finish per item
    const int n = c.value();
    note_launch(typeid(W).name(), n);
    for (int i = 0; i < n; ++i) {
        typename W::State st;
        if (w.begin(i, st)) { while (!w.step(st)) {} }
        w.finish(st);
    }
}
template <class W>
inline void launch_chunked(Count c, const W &w, int = 4, int = 16, int = 8) { launch_persistent(c, w); }     // (hip/exec.h: wave-local refill)
} // namespace exec

// ---- host stand-ins for the hand-written kernels (compact.hip / trace.hip) ----------------------
#include "bvh.h"
#include "tuning.h"
#include "trace_sim.h"
namespace exec {
inline void select_device(int /*use_gpu*/, int /*gpu_index*/) {}
template <class P>
inline int compact(const int *in, int n, int *out, const P &pred) {
    int c = 0;
    for (int i = 0; i < n; ++i) { int p = in ? in[i] : i; if (pred(p)) out[c++] = p; }
    return c;
}
template <class P>
inline Count compact_dev(const int *in, Count n, int *out, const P &pred, const Count *append_at = nullptr, int *dyn = nullptr, int inc = 0,
                         int *pos_out = nullptr, int /*scratch*/ = 0) {
    const int n_in = n.value();
    const int base = append_at ? append_at->value() : 0;
    int *result = new_count();
    if (pos_out) { int c = 0; for (int i = 0; i < n_in; ++i) { const int p = in ? in[i] : i; if (pred(p)) pos_out[base + c++] = i; } }
    *result = base + compact(in, n_in, out + base, pred);
    if (dyn && n_in > 0) *dyn += inc;
    return Count(result, n.upper + (append_at ? append_at->upper : 0));
}
struct TraceStats { double closest_ms = 0, any_ms = 0, closest_union_ms = 0, any_union_ms = 0; uint64_t closest_launches = 0, any_launches = 0, closest_rays = 0, any_rays = 0, nodes[2] = {0, 0}, tris[2] = {0, 0}, wide_nodes[2] = {0, 0}; bool timing = false, counting = false; };
inline TraceStats &trace_stats() { static TraceStats s; return s; }
inline void trace(const rt::BvhD &bvh, const rt::RayRec *rays, rt::HitRec *hits, Count cnt_n, bool any, bool = false) {
    const int n = cnt_n.value();
    TraceStats &st = trace_stats();
    rt::Counters cnt{0, 0}, wcnt{0, 0};
    static const bool sim = std::getenv("RDR_TRACE_SIM") != nullptr;
    if (sim) tracesim::launch(bvh, rays, n, any);
    for (int i = 0; i < n; ++i) {
        const rt::RayRec &r = rays[i];
        rt::Hit h{0.f, -1, -1};
        if (!(r.tmax < 0.f)) {
            float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
            int stack[rt::kTraverseStack];
            const bool binary = rdr::tuning().has(RDR_TUNE_TRACE_BINARY);
            if (bvh.wide && !binary) {            // like the GPU build: the 4-wide records when the hierarchy has them
                int wstack[64];
                if (bvh.wide_stack_need > 64) throw std::runtime_error("wide hierarchy deeper than the harness stack");
                h = any ? rt::traverse_wide<true>(bvh, o, d, r.tmin, r.tmax, wstack, 1, st.counting ? &wcnt : nullptr)
                        : rt::traverse_wide<false>(bvh, o, d, r.tmin, r.tmax, wstack, 1, st.counting ? &wcnt : nullptr);
            } else
            h = any ? rt::traverse<true>(bvh, o, d, r.tmin, r.tmax, stack, 1, st.counting ? &cnt : nullptr)
                    : rt::traverse<false>(bvh, o, d, r.tmin, r.tmax, stack, 1, st.counting ? &cnt : nullptr);
        }
        hits[i] = rt::HitRec{h.shape, h.shape >= 0 ? h.prim : -1};
    }
    (any ? st.any_launches : st.closest_launches)++;
    (any ? st.any_rays : st.closest_rays) += n;
    st.nodes[any ? 1 : 0] += cnt.nodes; st.tris[any ? 1 : 0] += cnt.tris + wcnt.tris; st.wide_nodes[any ? 1 : 0] += wcnt.nodes;
}
} // namespace exec
namespace exec {
inline void trace_stats_collect() {}
inline int choose_replicas(size_t, size_t) { return 1; }
inline size_t replica_budget(size_t) { return 0; }
}
namespace rdr { struct ReplicaLayout { const double *hot_end; unsigned long long hot_stride, stride; unsigned hot_mask, mask; }; }
namespace exec { inline void set_replicas(const rdr::ReplicaLayout &) {} }
namespace rdr { inline void accum_f32(float *p, float v) { *p += v; } }

namespace exec {
// debugging harness only: how much work the gather does per slot (printed at exit when RDR_GATHER_STATS is set)
struct GatherStats {
    long slots = 0, nodes = 0, edges = 0, cands = 0, overflow = 0, hist[12] = {0};
    long wave_max_nodes = 0, wave_max_edges = 0, cur_n = 0, cur_e = 0, in_wave = 0, waves = 0, nhist[16] = {0};
    long surv = 0, cur_s = 0, wave_max_surv = 0, shist[16] = {0};
    ~GatherStats() {
        if (!getenv("RDR_GATHER_STATS") || slots == 0) return;
        fprintf(stderr, "[gather] per-wave(64) max nodes %.1f  max edge tests %.1f ; node-count histogram (log2 buckets):", (double)wave_max_nodes / (waves ? waves : 1), (double)wave_max_edges / (waves ? waves : 1));
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %ld", nhist[i]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[gather] survivors of the cheap tests/slot %.3f, per-wave max %.2f, hist:", (double)surv / slots, (double)wave_max_surv / (waves ? waves : 1));
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %ld", shist[i]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[gather] slots %ld nodes/slot %.1f edge tests/slot %.2f positive leaves/slot %.4f overflow %ld  hist:", slots,
                (double)nodes / slots, (double)edges / slots, (double)cands / slots, overflow);
        for (int i = 0; i < 12; ++i) fprintf(stderr, " %ld", hist[i]);
        fprintf(stderr, "\n");
    }
};
inline void gather_stats_add(long nodes, long edges, int ncand, int surv) {
    static GatherStats st;
    st.surv += surv; if (surv > st.cur_s) st.cur_s = surv; st.shist[surv < 15 ? surv : 15]++;
    st.slots++; st.nodes += nodes; st.edges += edges; st.cands += ncand; st.overflow += ncand > 8; st.hist[ncand < 11 ? ncand : 11]++;
    int b = 0; while ((1L << b) <= nodes && b < 15) b++;
    st.nhist[b]++;
    if (nodes > st.cur_n) st.cur_n = nodes;
    if (edges > st.cur_e) st.cur_e = edges;
    if (++st.in_wave == 64) { st.wave_max_nodes += st.cur_n; st.wave_max_edges += st.cur_e; st.wave_max_surv += st.cur_s; st.waves++; st.cur_n = st.cur_e = st.cur_s = 0; st.in_wave = 0; }
}
}
