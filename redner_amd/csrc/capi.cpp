// capi.cpp -- extern "C" entry points declared in include/redner_amd.h.
#include "../../include/redner_amd.h"
#include "render.h"
#include "tuning.h"
#include "edges.h"
#include <cstdio>
#include "scene.h"
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <vector>
#include <cstdlib>

namespace {
thread_local std::string g_last_error;
// rdr_render / rdr_scene_create / rdr_scene_trace are serialised PER DEVICE: the reference's render() is not re-entrant either
// (global thread pool, src/parallel.cpp:10-14) but runs with the GIL held; ctypes releases the GIL, and two concurrent calls on
// one device would share its replicated-accumulator symbols, its helper threads and the Scene caches.  Calls on DIFFERENT
// devices run side by side (one host thread per device in one process): every piece of state they touch is per device (buffer
// pool free lists, Scene caches, helper threads, streams, compaction scratch) or per host thread (stream, tuning, staging), and
// what is shared (host thread pool, edge-builder thread, trace statistics) has its own lock.
std::recursive_mutex g_device_lock[16];
std::recursive_mutex &device_lock(int gpu_index) { return g_device_lock[(gpu_index < 0 ? 0 : gpu_index) & 15]; }
void set_error(const char *what) { g_last_error = what ? what : "unknown error"; }
// rdr_set_stream: the stream the calling thread's launches are ordered on (null = the null stream)
thread_local void *g_user_stream = nullptr;
void use_caller_stream() { exec::ctx().stream = (hipStream_t)g_user_stream; }
}

// rdr_debug_libm: one argument per lane through the routines every stage calls
struct LibmProbe {
    int fn; const double *x, *y; double *out;
    RDR_FN void operator()(int i) const {
        const double a = x[i], b = y[i];
        double r;
        switch (fn) {
            case 0: r = gm::sin(a); break;
            case 1: r = gm::cos(a); break;
            case 2: r = gm::atan2(a, b); break;
            case 3: r = gm::atan(a); break;
            case 4: r = gm::acos(a); break;
            case 5: r = gm::log(a); break;
            default: r = gm::pow(a, b); break;
        }
        out[i] = r;
    }
};

extern "C" {

const char *rdr_last_error(void) { return g_last_error.c_str(); }

rdr_scene *rdr_scene_create(const rdr_camera_desc *camera, const rdr_shape_desc *shapes, int num_shapes,
                            const rdr_material_desc *materials, int num_materials,
                            const rdr_area_light_desc *area_lights, int num_area_lights,
                            const rdr_envmap_desc *envmap, int use_gpu, int gpu_index,
                            int use_primary_edge_sampling, int use_secondary_edge_sampling) {
    try {
        g_last_error.clear();
        std::lock_guard<std::recursive_mutex> lk(device_lock(gpu_index));
        use_caller_stream();
        return reinterpret_cast<rdr_scene *>(rdr::create_scene(camera, shapes, num_shapes, materials, num_materials,
                                                               area_lights, num_area_lights, envmap, use_gpu, gpu_index,
                                                               use_primary_edge_sampling, use_secondary_edge_sampling));
    } catch (const std::exception &e) {
        set_error(e.what());
        return nullptr;
    }
}

void rdr_scene_destroy(rdr_scene *scene) {
    if (!scene) return;
    rdr::Scene *s = reinterpret_cast<rdr::Scene *>(scene);
    // under the device's lock like every other call that touches its pool / caches (the destructor returns buffers and may join
    // the Scene's edge build)
    std::lock_guard<std::recursive_mutex> lk(device_lock(s->gpu_index));
    delete s;
}

int rdr_scene_max_generic_texture_dimension(const rdr_scene *scene) {
    return scene ? reinterpret_cast<const rdr::Scene *>(scene)->max_generic_texture_dimension : 0;
}

int rdr_render(const rdr_scene *scene, const rdr_render_options *options, float *rendered_image,
               const float *d_rendered_image, const rdr_dscene_desc *d_scene, float *screen_gradient_image,
               float *debug_image) {
    try {
        g_last_error.clear();
        if (!scene || !options) throw std::runtime_error("rdr_render: scene and options are required");
        const rdr::Scene &s = *reinterpret_cast<const rdr::Scene *>(scene);
        std::lock_guard<std::recursive_mutex> lk(device_lock(s.gpu_index));
        exec::select_device(1, s.gpu_index);
        use_caller_stream();
        rdr::render(s, *options, rendered_image, d_rendered_image, d_scene, screen_gradient_image, debug_image);
        return 0;
    } catch (const std::exception &e) {
        set_error(e.what());
        return 1;
    }
}

int rdr_compute_num_channels(const int *channels, int num_channels, int max_generic_texture_dimension) {
    return rdr::compute_num_channels(channels, num_channels, max_generic_texture_dimension);
}

void rdr_trace_stats_enable(int timing, int counting) {
    exec::trace_stats().timing = timing != 0;
    exec::trace_stats().counting = counting != 0;
}
void rdr_trace_stats_reset(void) {
    exec::TraceStats &s = exec::trace_stats();
    bool t = s.timing, c = s.counting;
    exec::trace_stats_collect();
    s = exec::TraceStats();
    s.timing = t; s.counting = c;
}
void rdr_trace_stats_get(rdr_trace_stats *out) {
    exec::trace_stats_collect();
    const exec::TraceStats &s = exec::trace_stats();
    out->closest_ms = s.closest_ms; out->any_ms = s.any_ms;
    out->closest_launches = s.closest_launches; out->any_launches = s.any_launches;
    out->closest_rays = s.closest_rays; out->any_rays = s.any_rays;
    out->closest_nodes = s.nodes[0]; out->closest_tris = s.tris[0];
    out->any_nodes = s.nodes[1]; out->any_tris = s.tris[1];
    out->closest_wide_nodes = s.wide_nodes[0]; out->any_wide_nodes = s.wide_nodes[1];
    out->closest_union_ms = s.closest_union_ms; out->any_union_ms = s.any_union_ms;
}

uint64_t rdr_trim_cache(void) {
    std::unique_lock<std::recursive_mutex> all[16];                 // every device, in index order (a call holds one lock only)
    for (int d = 0; d < 16; ++d) all[d] = std::unique_lock<std::recursive_mutex>(g_device_lock[d]);
    rdr::drop_edge_cache();            // the last Scene's edge structures, kept for the next one (scene.cpp: EdgeCache)
    const uint64_t bytes = exec::pool_cached_bytes();
    exec::pool_trim();
    return bytes;
}

void rdr_set_stream(void *hip_stream) { g_user_stream = hip_stream; }
void rdr_set_pool_cap_mb(int64_t megabytes) { exec::pool_set_cap(megabytes < 0 ? -1 : (long long)megabytes << 20); }
int64_t rdr_get_pool_cap_mb(void) { return (int64_t)(exec::pool_cap_bytes() >> 20); }
void rdr_set_build_flags(unsigned flags) { rdr::build_flags_ref().store(flags); }

void rdr_debug_counters_get(rdr_debug_counters *out) {
    out->device_mallocs = exec::pool_device_mallocs();
    out->host_count_reads = exec::host_count_reads();
    out->last_batch_samples = (uint64_t)rdr::last_schedule()[0].load();
    out->last_workers = (uint64_t)rdr::last_schedule()[1].load();
}

int rdr_debug_dump_edges(const rdr_scene *scene, const char *path) {
    const rdr::Scene &s = *reinterpret_cast<const rdr::Scene *>(scene);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    {
        std::lock_guard<std::recursive_mutex> lk(device_lock(s.gpu_index));
        exec::select_device(1, s.gpu_index);
        try { s.edge_data(); } catch (const std::exception &e) { set_error(e.what()); fclose(f); return 1; }
    }
    if (!s.edges) { fprintf(f, "edges 0\n"); fclose(f); return 0; }
    if (s.edges->device_trees) {
        std::lock_guard<std::recursive_mutex> lk(device_lock(s.gpu_index));
        try { rdr::download_edge_trees(*s.edges); } catch (const std::exception &e) { set_error(e.what()); fclose(f); return 1; }
    }
    const rdr::EdgeData &ed = *s.edges;
    fprintf(f, "edges %d\n", (int)ed.edges.size());
    for (const rdr::EdgeD &e : ed.edges) fprintf(f, "%d %d %d %d %d\n", e.shape_id, e.v0, e.v1, e.f0, e.f1);
    if (!ed.cs_nodes.empty() || !ed.ncs_nodes.empty()) {
        fprintf(f, "expand %.17g\n", ed.edge_bounds_expand);
        for (int t = 0; t < 2; ++t) {
            const std::vector<rdr::EdgeNode> &nodes = t == 0 ? ed.cs_nodes : ed.ncs_nodes;
            int nl = t == 0 ? ed.cs_leaves : ed.ncs_leaves;
            int nn = (int)nodes.size() - nl;
            fprintf(f, "%s %d %d\n", t == 0 ? "cs" : "ncs", nl == 0 ? 0 : nn, nl);
            for (size_t i = 0; i < nodes.size(); ++i) {
                const rdr::EdgeNode &n = nodes[i];
                fprintf(f, "%d %d %d %d %d %.17g %.17g", (int)i, n.parent, n.child0, n.child1, n.edge_id, n.wlen, n.cost);
                fprintf(f, " %.17g %.17g %.17g %.17g %.17g %.17g", n.p_min.x, n.p_min.y, n.p_min.z, n.p_max.x, n.p_max.y, n.p_max.z);
                if (t == 1) fprintf(f, " %.17g %.17g %.17g %.17g %.17g %.17g", n.d_min.x, n.d_min.y, n.d_min.z, n.d_max.x, n.d_max.y, n.d_max.z);
                fprintf(f, "\n");
            }
        }
    }
    fclose(f);
    return 0;
}

/* Test hook: the hierarchy this Scene's kernels built against the host builder's (bvh.cpp) on the same mesh arrays -- node
 * records, leaf order, triangle records, 4-wide records.  Returns the number of records that differ (0: identical), or -1
 * when the Scene's hierarchy is a refit of an earlier build / was not built by kernels. */
int rdr_debug_bvh_check(const rdr_scene *scene) {
    try {
        const rdr::Scene &s = *reinterpret_cast<const rdr::Scene *>(scene);
        if (!s.bvh_dev || s.bvh_dev->parent) return -1;
        std::lock_guard<std::recursive_mutex> lk(device_lock(s.gpu_index));
        exec::select_device(1, s.gpu_index);
        use_caller_stream();
        std::vector<rt::MeshView> meshes(s.shapes.size());
        for (size_t i = 0; i < s.shapes.size(); ++i) meshes[i] = rt::MeshView{s.h_vertices[i].data(), s.h_indices[i].data(), s.shapes[i].num_triangles};
        const rt::BvhHost h = rt::build_bvh(meshes);
        const rt::BvhDev &d = *s.bvh_dev;
        int bad = 0;
        if ((int)h.nodes.size() != d.num_nodes || (int)h.ids.size() != 2 * d.num_slots || (int)h.wide.size() != d.num_wide ||
            h.depth != d.depth || h.wide_stack_need != d.wide_stack_need)
            return 1000000 + std::abs((int)h.nodes.size() - d.num_nodes);
        std::vector<rt::Node> nodes(d.num_nodes);
        std::vector<int> ids((size_t)2 * d.num_slots);
        std::vector<float> tris((size_t)9 * d.num_slots);
        std::vector<rt::Node4> wide(d.num_wide);
        exec::download(nodes.data(), d.nodes, sizeof(rt::Node) * nodes.size());
        exec::download(ids.data(), d.ids, sizeof(int) * ids.size());
        exec::download(tris.data(), d.tris, sizeof(float) * tris.size());
        exec::download(wide.data(), d.wide, sizeof(rt::Node4) * wide.size());
        for (size_t i = 0; i < nodes.size(); ++i) {
            const rt::Node &a = nodes[i], &b = h.nodes[i];
            bool same = a.a == b.a && a.b == b.b;
            for (int k = 0; k < 3; ++k) same = same && a.lo[k] == b.lo[k] && a.hi[k] == b.hi[k];
            bad += !same;
        }
        for (size_t i = 0; i < ids.size(); ++i) bad += ids[i] != h.ids[i];
        for (size_t i = 0; i < tris.size(); ++i) bad += !(tris[i] == h.tris[i]);
        for (size_t i = 0; i < wide.size(); ++i) {
            const rt::Node4 &a = wide[i], &b = h.wide[i];
            bool same = a.aux[0] == b.aux[0];
            for (int k = 0; k < 4; ++k)
                same = same && a.link[k] == b.link[k] && a.lox[k] == b.lox[k] && a.loy[k] == b.loy[k] && a.loz[k] == b.loz[k] &&
                       a.hix[k] == b.hix[k] && a.hiy[k] == b.hiy[k] && a.hiz[k] == b.hiz[k];
            bad += !same;
        }
        return bad;
    } catch (const std::exception &e) {
        set_error(e.what());
        return -2;
    }
}

int rdr_scene_trace(const rdr_scene *scene, const float *rays, int32_t *hits, int num_rays, int any_hit) {
    try {
        g_last_error.clear();
        const rdr::Scene &s = *reinterpret_cast<const rdr::Scene *>(scene);
        std::lock_guard<std::recursive_mutex> lk(device_lock(s.gpu_index));
        exec::select_device(1, s.gpu_index);
        use_caller_stream();
        exec::trace(s.bvh, reinterpret_cast<const rt::RayRec *>(rays), reinterpret_cast<rt::HitRec *>(hits), num_rays, any_hit != 0);
        exec::sync();
        return 0;
    } catch (const std::exception &e) {
        set_error(e.what());
        return 1;
    }
}

/* Test hook: the library's transcendental routines (libm_exact.h) evaluated by a kernel on `n` arguments (HOST pointers;
 * `y` only for the two-argument functions).  fn: 0 sin, 1 cos, 2 atan2(x, y), 3 atan, 4 acos, 5 log, 6 pow(x, y). */
int rdr_debug_libm(int fn, const double *x, const double *y, double *out, int n) {
    try {
        g_last_error.clear();
        if (fn < 0 || fn > 6 || n < 0) throw std::runtime_error("rdr_debug_libm: bad arguments");
        std::lock_guard<std::recursive_mutex> lk(device_lock(exec::current_device()));
        exec::select_device(1, exec::current_device());        // the calling thread's device: checked, not changed
        use_caller_stream();
        const size_t bytes = sizeof(double) * (size_t)n;
        struct Held {                      // released on every path out, also when the launch or a copy throws
            double *p[3] = {nullptr, nullptr, nullptr};
            ~Held() { exec::device_sync(); for (double *q : p) exec::dfree(q); }
        } held;
        for (double *&q : held.p) q = (double *)exec::dmalloc(bytes);
        double *dx = held.p[0], *dy = held.p[1], *dout = held.p[2];
        exec::upload(dx, x, bytes);
        if (y) exec::upload(dy, y, bytes); else exec::zero(dy, bytes);
        exec::launch(exec::Count(n), LibmProbe{fn, dx, dy, dout});
        exec::download(out, dout, bytes);
        exec::sync();
        return 0;
    } catch (const std::exception &e) {
        set_error(e.what());
        return 1;
    }
}

} // extern "C"
