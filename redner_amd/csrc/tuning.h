// tuning.h -- the resolved form of rdr_tuning (include/redner_amd.h) and the library-wide build flags.
//
// Rounds 1-3 selected kernels and schedules through ~45 read-once RDR_* environment variables, so a test that wanted another
// path had to start a process.  Now render() resolves `rdr_render_options::tuning` ONCE per call into a Tuning -- a field
// that is 0 takes the environment variable of the same meaning if that is set (A/B shell scripts keep working), else the
// built-in default -- and every piece of the host driver reads the calling thread's current Tuning (`rdr::tuning()`;
// helper threads get a copy with their job).
#pragma once
#include "../../include/redner_amd.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace rdr {

struct Tuning {
    unsigned flags = 0;
    int batch_samples = 16;
    long long batch_lanes = 0;          // 0 = decide from the device's free memory (render.cpp)
    int workers = 0;                    // 0 = by size (exec::sample_workers)
    int refill_k = 4, refill_idle = 24, refill_steps = 4;
    int wide_max = 1 << 19;
    int gather_budget = 256;
    int gather_heavy_cap = -1, gather_work_cap = -1;      // -1 = the buffers' full capacity
    double mem_available_mb = -1.0;     // < 0: ask the driver
    int refill_sort = 1;                // 0 queue order, 1 octant, 2 octant x axis (trace.hip)
    int pickh_k = 1, pickh_idle = 8, pickh_steps = 8;
    bool has(unsigned f) const { return (flags & f) != 0; }
};

namespace detail {
inline const char *env(const char *name) { return std::getenv(name); }
inline bool env_set(const char *name) { return std::getenv(name) != nullptr; }
// the environment's contribution, read once per process
struct EnvDefaults {
    unsigned flags = 0;
    int batch_samples = 0; long long batch_lanes = 0; int workers = 0;
    int refill_k = 0, refill_idle = 0, refill_steps = 0, wide_max = 0, gather_budget = 0, heavy_cap = -1, work_cap = -1;
    double mem_mb = -1.0;
    int refill_order = 0, pickh_k = 0, pickh_idle = 0, pickh_steps = 0;
    EnvDefaults() {
        if (env_set("RDR_NO_OVERLAP") || env_set("RDR_DEBUG_DUMP")) flags |= RDR_TUNE_NO_OVERLAP;
        if (env_set("RDR_FORCE_GENERAL")) flags |= RDR_TUNE_FORCE_GENERAL;
        if (env_set("RDR_PICKN_WALK")) flags |= RDR_TUNE_PICKN_WALK;
        if (env_set("RDR_PICKH_FUSED")) flags |= RDR_TUNE_PICKH_FUSED;
        if (env_set("RDR_PICKH_LAZY")) flags |= RDR_TUNE_PICKH_LAZY;
        if (env_set("RDR_NO_HOIST")) flags |= RDR_TUNE_NO_HOIST;
        if (env_set("RDR_TRACE_REFILL_ALL")) flags |= RDR_TUNE_REFILL_ALL;
        if (env_set("RDR_TRACE_BINARY")) flags |= RDR_TUNE_TRACE_BINARY;
        if (env_set("RDR_TRACE_NO_LDS_TOP")) flags |= RDR_TUNE_TRACE_NO_LDS_TOP;
        if (env_set("RDR_NO_FUSED_BOUNCE")) flags |= RDR_TUNE_NO_FUSED_BOUNCE;
        if (env_set("RDR_PICKH_ONE_LAUNCH")) flags |= RDR_TUNE_PICKH_ONE_LAUNCH;
        if (env_set("RDR_NO_NEE_COMPACT")) flags |= RDR_TUNE_NO_NEE_COMPACT;
        if (env_set("RDR_LARGE_FRAME_FORMS")) flags |= RDR_TUNE_LARGE_FORMS;
        if (env_set("RDR_TRACE_EVERY_CONTINUATION")) flags |= RDR_TUNE_TRACE_EVERY_CONTINUATION;
        if (const char *e = env("RDR_TRACE_REFILL")) {
            int k = 0, idle = 0, steps = 0;
            const int got = std::sscanf(e, "%d,%d,%d", &k, &idle, &steps);
            if (got >= 1 && k <= 0) flags |= RDR_TUNE_REFILL_OFF;
            if (got >= 1 && k > 0) refill_k = k;
            if (got >= 2 && idle > 0) refill_idle = idle;
            if (got >= 3 && steps > 0) refill_steps = steps;
        }
        if (const char *e = env("RDR_BATCH")) batch_samples = std::atoi(e);
        if (const char *e = env("RDR_BATCH_LANES")) batch_lanes = std::atoll(e);
        if (const char *e = env("RDR_WORKERS")) workers = std::atoi(e);
        if (const char *e = env("RDR_WIDE_MAX")) wide_max = std::atoi(e);
        if (const char *e = env("RDR_GATHER_BUDGET")) gather_budget = std::atoi(e);
        if (const char *e = env("RDR_GATHER_CAPS")) { int h = 0, w = 0; if (std::sscanf(e, "%d,%d", &h, &w) == 2) { heavy_cap = h; work_cap = w; } }
        if (const char *e = env("RDR_MEM_AVAILABLE_MB")) mem_mb = std::atof(e);
        if (const char *e = env("RDR_REFILL_SORT")) refill_order = std::atoi(e) + 1;
        if (const char *e = env("RDR_PICKH_REFILL")) {
            int k = 0, idle = 0, steps = 0;
            const int got = std::sscanf(e, "%d,%d,%d", &k, &idle, &steps);
            if (got >= 1 && k > 0) pickh_k = k;
            if (got >= 2 && idle > 0) pickh_idle = idle;
            if (got >= 3 && steps > 0) pickh_steps = steps;
        }
    }
};
inline const EnvDefaults &env_defaults() { static const EnvDefaults d; return d; }
inline int clampi(long long v, long long lo, long long hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }
}

inline Tuning resolve_tuning(const rdr_tuning *t) {
    const detail::EnvDefaults &e = detail::env_defaults();
    static const rdr_tuning zero{};
    const rdr_tuning &u = t ? *t : zero;
    Tuning r;
    r.flags = u.flags | e.flags;
    auto pick = [](long long field, long long env, long long dflt) { return field != 0 ? field : (env != 0 ? env : dflt); };
    r.batch_samples = detail::clampi(pick(u.batch_samples, e.batch_samples, 16), 1, 16);
    r.batch_lanes = pick(u.batch_lanes, e.batch_lanes, 0);
    if (r.batch_lanes < 0) r.batch_lanes = 1;
    r.workers = detail::clampi(pick(u.workers, e.workers, 0), 0, 4);
    r.refill_k = detail::clampi(pick(u.refill_rays_per_lane, e.refill_k, 4), 1, 64);
    r.refill_idle = detail::clampi(pick(u.refill_idle_lanes, e.refill_idle, 24), 1, 64);
    r.refill_steps = detail::clampi(pick(u.refill_steps, e.refill_steps, 4), 1, 1024);
    r.wide_max = (int)pick(u.wide_max_rays, e.wide_max, 1 << 19);
    r.gather_budget = detail::clampi(pick(u.gather_budget, e.gather_budget, 256), 1, 1 << 20);
    r.gather_heavy_cap = u.gather_heavy_cap_plus1 > 0 ? u.gather_heavy_cap_plus1 - 1 : e.heavy_cap;
    r.gather_work_cap = u.gather_work_cap_plus1 > 0 ? u.gather_work_cap_plus1 - 1 : e.work_cap;
    r.mem_available_mb = u.mem_available_mb > 0 ? (double)u.mem_available_mb : e.mem_mb;
    r.refill_sort = detail::clampi(pick(u.refill_order, e.refill_order, 2), 1, 3) - 1;
    r.pickh_k = detail::clampi(pick(u.pickh_slots_per_lane, e.pickh_k, 1), 1, 64);
    r.pickh_idle = detail::clampi(pick(u.pickh_idle_lanes, e.pickh_idle, 8), 1, 64);
    r.pickh_steps = detail::clampi(pick(u.pickh_steps, e.pickh_steps, 8), 1, 1024);
    return r;
}

// how the last gradient render was scheduled {samples per launch set, workers} (rdr_debug_counters: bench.py's single-chain legs repeat the batch size)
inline std::atomic<int> *last_schedule() { static std::atomic<int> s[2]; return s; }

// the calling thread's current tuning (default-constructed = resolve_tuning(nullptr) on first use)
inline Tuning &tuning() {
    static thread_local Tuning t = resolve_tuning(nullptr);
    return t;
}
struct TuningScope {          // render() installs the call's tuning for its duration
    Tuning saved;
    explicit TuningScope(const Tuning &t) : saved(tuning()) { tuning() = t; }
    ~TuningScope() { tuning() = saved; }
    TuningScope(const TuningScope &) = delete;
};

// rdr_set_build_flags | what the environment asks for
inline std::atomic<unsigned> &build_flags_ref() { static std::atomic<unsigned> f{0}; return f; }
inline unsigned build_flags() {
    static const unsigned from_env = (detail::env_set("RDR_NO_REFIT") ? (unsigned)RDR_BUILD_NO_REFIT : 0u) |
                                     (detail::env_set("RDR_NO_EDGE_CACHE") ? (unsigned)RDR_BUILD_NO_EDGE_CACHE : 0u) |
                                     (detail::env_set("RDR_SYNC_EDGES") ? (unsigned)RDR_BUILD_SYNC_EDGES : 0u) |
                                     (detail::env_set("RDR_EDGE_HOST_BUILD") ? (unsigned)RDR_BUILD_EDGE_HOST_BUILD : 0u);
    return from_env | build_flags_ref().load(std::memory_order_relaxed);
}

} // namespace rdr
