// scene.cpp -- Scene construction (see scene.h).
#include "scene.h"
#include "tuning.h"
#include "bvh_gpu.h"
#include "hostpool.h"
#include "surface.h"
#include "sobol.h"
#include <memory>
#include "edges.h"
#include <thread>
#include <mutex>
#include <functional>
#include <deque>
#include <condition_variable>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>

extern "C" const uint64_t rdr_sobol_table[];
extern "C" const float rdr_ltc_table[];

namespace rdr {

namespace {

template <class T>
T *to_device(Scene &s, const T *src, size_t count) {
    T *p = (T *)exec::pool_alloc(sizeof(T) * (count ? count : 1));
    s.owned.push_back(p);
    if (count) exec::upload_async(p, src, sizeof(T) * count);      // create_scene() ends the batch with upload_flush()
    return p;
}

template <class T>
std::vector<T> from_device(const T *src, size_t count) {
    std::vector<T> v(count);
    if (src && count) exec::download(v.data(), src, sizeof(T) * count);
    return v;
}

TexD convert_tex(const rdr_texture_desc &t) {
    TexD r;
    std::memset(&r, 0, sizeof(r));
    r.num_levels = std::min(t.num_levels, (int)kMaxMip);
    r.channels = t.channels;
    r.uv_scale = t.uv_scale;
    for (int i = 0; i < r.num_levels; ++i) { r.texels[i] = t.texels[i]; r.width[i] = t.width[i]; r.height[i] = t.height[i]; }
    return r;
}

M4 m4_from(const float *p) { M4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = p[4 * i + j]; return m; }
M3 m3_from(const float *p) { M3 m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.m[i][j] = p[3 * i + j]; return m; }

} // namespace

namespace {
// One long-lived thread builds the edge structures of every Scene, one after the other (the topology caches of edges.cpp are
// process-wide): the host part on the pool, then the device copies and the hierarchy kernels on a stream of its own.  It keeps
// its pinned staging buffer and its stream from Scene to Scene (a thread per Scene would allocate them each time).
class EdgeBuilder {
public:
    static EdgeBuilder &get() { static EdgeBuilder *b = new EdgeBuilder(); return *b; }      // never destroyed
    typedef std::shared_ptr<EdgeData> Result;
    std::future<Result> submit(std::function<Result()> fn) {
        std::packaged_task<Result()> task(std::move(fn));
        std::future<Result> fut = task.get_future();
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(task));
        }
        cv_.notify_all();
        return fut;
    }
    // Process exit: a build that is still running (a Scene that was never destroyed) must not be inside the HIP runtime
    // when the runtime's own exit handlers tear it down.  Registered after the runtime's (the first Scene initialised it),
    // so it runs before them.
    void drain() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.empty() && !running_; });
    }
private:
    EdgeBuilder() {
        std::thread([this] { loop(); }).detach();
        std::atexit([] { EdgeBuilder::get().drain(); });
    }
    void loop() {
        for (;;) {
            std::packaged_task<Result()> task;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty(); });
                task = std::move(q_.front());
                q_.pop_front();
                running_ = true;
            }
            task();
            {
                std::lock_guard<std::mutex> lk(m_);
                running_ = false;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<Result()>> q_;
    bool running_ = false;
};
}

const EdgeData *Scene::edge_data() const {
    std::lock_guard<std::mutex> lk(edge_join);
    if (edge_build.valid()) {
        edges_ref = edge_build.get();             // rethrows what the build threw
        edge_build = std::shared_future<std::shared_ptr<EdgeData>>();
        edges = edges_ref.get();
    }
    return edges;
}

Scene::~Scene() {
    // a build that is still running reads this Scene's host arrays: wait for it (the structures themselves may live on in
    // the cache / in other Scenes and are released with their last owner)
    if (edge_build.valid()) { try { edge_build.wait(); } catch (...) {} }
    for (void *p : owned) exec::pool_free(p);
}

int compute_num_channels(const int *channels, int n, int max_generic) {
    int total = 0;
    for (int i = 0; i < n; ++i) {
        switch (channels[i]) {
            case RDR_CH_RADIANCE: case RDR_CH_POSITION: case RDR_CH_GEOMETRY_NORMAL: case RDR_CH_SHADING_NORMAL:
            case RDR_CH_DIFFUSE_REFLECTANCE: case RDR_CH_SPECULAR_REFLECTANCE: case RDR_CH_VERTEX_COLOR:
                total += 3; break;
            case RDR_CH_UV: case RDR_CH_BARYCENTRIC_COORDINATES:
                total += 2; break;
            case RDR_CH_ALPHA: case RDR_CH_DEPTH: case RDR_CH_ROUGHNESS: case RDR_CH_SHAPE_ID:
            case RDR_CH_TRIANGLE_ID: case RDR_CH_MATERIAL_ID:
                total += 1; break;
            case RDR_CH_GENERIC_TEXTURE:
                total += max_generic; break;
            default: return -1;
        }
    }
    return total;
}

// The inputs and the result of the last edge build (see create_scene): host mirrors of every shape + a reference to the
// device structures.  It outlives the Scene it was made for on purpose (the next Scene of an optimisation loop takes it), and
// rdr_trim_cache() -- "give back what the library keeps for later" -- drops it along with the parked buffers.
struct EdgeCache {
    int gpu_index = -1; bool primary = false, secondary = false;
    CameraD cam;
    std::vector<std::vector<float>> vertices, normals;
    std::vector<std::vector<int>> indices;
    std::shared_future<std::shared_ptr<EdgeData>> result;
};
// (one per device: calls on one device are serialised by the API lock of that device, capi.cpp)
static EdgeCache *edge_cache(int device) { static EdgeCache *c = new EdgeCache[16]; return c + ((device < 0 ? 0 : device) & 15); }
// The last triangle hierarchy that was BUILT (index buffers + the build): a Scene with the same connectivity refits a copy of it.
struct TopologyCache { std::vector<std::vector<int>> indices; rt::BvhHost bvh; std::shared_ptr<rt::BvhDev> dev; int gpu_index = -1; };
static TopologyCache *topology_cache(int device) { static TopologyCache *c = new TopologyCache[16]; return c + ((device < 0 ? 0 : device) & 15); }
void drop_edge_cache() {
    for (int d = 0; d < 16; ++d) {       // (rdr_trim_cache holds every device's lock)
        *edge_cache(d) = EdgeCache();    // the device structures go when the last Scene that shares them does
        *topology_cache(d) = TopologyCache();
    }
    drop_gather_cache();
}

Scene *create_scene(const rdr_camera_desc *cam, const rdr_shape_desc *shapes, int num_shapes,
                    const rdr_material_desc *materials, int num_materials,
                    const rdr_area_light_desc *area_lights, int num_area_lights,
                    const rdr_envmap_desc *envmap, int use_gpu, int gpu_index,
                    int primary_edges, int secondary_edges) {
    exec::select_device(use_gpu, gpu_index);   // throws when no gfx950 device / use_gpu == 0
    if (!cam) throw std::runtime_error("Scene: camera is required");
    if (cam->camera_type < 0 || cam->camera_type > 3) throw std::runtime_error("Scene: unknown camera type");

    std::unique_ptr<Scene> sp(new Scene());
    Scene &s = *sp;
    PhaseTimer timer("scene build");
    s.gpu_index = gpu_index;
    s.build_flags = build_flags();
    s.use_primary_edges = primary_edges != 0;
    s.use_secondary_edges = secondary_edges != 0;

    // ---- camera (src/camera.h:22-65) ----
    CameraD &c = s.d.cam;
    std::memset(&c, 0, sizeof(c));
    c.width = cam->width; c.height = cam->height;
    c.clip_near = cam->clip_near; c.kind = cam->camera_type;
    c.vp_x0 = cam->viewport_beg[0]; c.vp_y0 = cam->viewport_beg[1];
    c.vp_x1 = cam->viewport_end[0]; c.vp_y1 = cam->viewport_end[1];
    if (cam->distortion_params) {
        c.distortion.defined = 1;
        for (int i = 0; i < 6; ++i) c.distortion.k[i] = cam->distortion_params[i];
        c.distortion.p[0] = cam->distortion_params[6]; c.distortion.p[1] = cam->distortion_params[7];
    }
    c.intrinsic_mat_inv = m3_from(cam->intrinsic_mat_inv);
    c.intrinsic_mat = m3_from(cam->intrinsic_mat);
    if (cam->cam_to_world) {
        c.cam_to_world = m4_from(cam->cam_to_world);
        c.world_to_cam = m4_from(cam->world_to_cam);
        c.use_look_at = 0;
    } else {
        c.position = v3f(cam->position); c.look = v3f(cam->look); c.up = v3f(cam->up);
        c.cam_to_world = look_at(c.position, c.look, c.up);
        c.world_to_cam = inverse(c.cam_to_world);
        c.use_look_at = 1;
    }

    // ---- shapes / materials / lights ----
    s.shapes.resize(num_shapes);
    s.h_vertices.resize(num_shapes); s.h_indices.resize(num_shapes);
    s.h_uvs.resize(num_shapes); s.h_normals.resize(num_shapes);
    s.h_uv_indices.resize(num_shapes); s.h_normal_indices.resize(num_shapes);
    for (int i = 0; i < num_shapes; ++i) {
        const rdr_shape_desc &in = shapes[i];
        ShapeD &o = s.shapes[i];
        o.vertices = in.vertices; o.indices = in.indices; o.uvs = in.uvs; o.normals = in.normals;
        o.geom = nullptr;              // set when the gathered triangle records have been uploaded (below)
        o.uv_indices = in.uv_indices; o.normal_indices = in.normal_indices; o.colors = in.colors;
        if (in.colors) s.has_vertex_colors = true;
        o.num_vertices = in.num_vertices; o.num_uv_vertices = in.num_uv_vertices;
        o.num_normal_vertices = in.num_normal_vertices; o.num_triangles = in.num_triangles;
        o.material_id = in.material_id; o.light_id = in.light_id;
    }
    {   // host mirrors of every mesh array: one batch, one synchronisation
        std::vector<exec::DownloadItem> items;
        auto want = [&](auto &vec, const auto *src, size_t count) {
            vec.resize(count);
            if (src && count) items.push_back(exec::DownloadItem{vec.data(), src, sizeof(vec[0]) * count});
        };
        for (int i = 0; i < num_shapes; ++i) {
            const rdr_shape_desc &in = shapes[i];
            want(s.h_vertices[i], in.vertices, (size_t)3 * in.num_vertices);
            want(s.h_indices[i], in.indices, (size_t)3 * in.num_triangles);
            if (in.normals) want(s.h_normals[i], in.normals, (size_t)3 * (in.num_normal_vertices > 0 ? in.num_normal_vertices : in.num_vertices));
            if (in.uvs) want(s.h_uvs[i], in.uvs, (size_t)2 * (in.num_uv_vertices > 0 ? in.num_uv_vertices : in.num_vertices));
            if (in.uv_indices) want(s.h_uv_indices[i], in.uv_indices, (size_t)3 * in.num_triangles);
            if (in.normal_indices) want(s.h_normal_indices[i], in.normal_indices, (size_t)3 * in.num_triangles);
        }
        exec::download_batch(items.data(), (int)items.size());
    }
    for (int i = 0; i < num_shapes; ++i) {
        const rdr_shape_desc &in = shapes[i];
        for (int k = 0; k < 3 * in.num_triangles; ++k)
            if (s.h_indices[i][k] < 0 || s.h_indices[i][k] >= in.num_vertices)
                throw std::runtime_error("Scene: triangle index out of range in shape " + std::to_string(i));
        if (in.material_id < 0 || in.material_id >= num_materials)
            throw std::runtime_error("Scene: material_id out of range in shape " + std::to_string(i));
    }
    s.materials.resize(num_materials);
    for (int i = 0; i < num_materials; ++i) {
        const rdr_material_desc &in = materials[i];
        MaterialD &o = s.materials[i];
        o.diffuse = convert_tex(in.diffuse_reflectance); o.specular = convert_tex(in.specular_reflectance);
        o.roughness = convert_tex(in.roughness); o.generic = convert_tex(in.generic_texture);
        o.normal_map = convert_tex(in.normal_map);
        o.diffuse.channels = 3; o.specular.channels = 3; o.roughness.channels = 1; o.normal_map.channels = 3;
        o.compute_specular_lighting = in.compute_specular_lighting; o.two_sided = in.two_sided;
        o.use_vertex_color = in.use_vertex_color;
        for (const TexD *t : {&o.diffuse, &o.specular, &o.roughness, &o.generic, &o.normal_map})
            if (t->num_levels > 1) s.has_mipmaps = true;
        for (const TexD *t : {&o.diffuse, &o.specular, &o.roughness})
            if (t->width[0] > 0 || t->height[0] > 0) s.has_textures = true;
        if (o.normal_map.num_levels > 0) s.has_textures = true;
        if (o.generic.num_levels > 0)
            s.max_generic_texture_dimension = std::max(s.max_generic_texture_dimension, o.generic.channels);
    }
    {   // Can a path that has bounced once still be specular enough for secondary edge sampling (min_roughness <= 0.01,
        // src/edge.cpp:1396-1401)?  Not if every material has a constant specular reflectance of exactly zero (the sampler then
        // always takes the Lambert lobe, which sets min_roughness to 1) and a constant roughness well above the threshold (for the
        // black-material corner where it takes the other branch, src/material.h:704-811).
        std::vector<float> vals((size_t)4 * num_materials, 1.f);
        std::vector<exec::DownloadItem> items;
        bool candidate = num_materials > 0;
        for (int i = 0; i < num_materials && candidate; ++i) {
            const MaterialD &m = s.materials[i];
            auto constant = [](const TexD &t) { return t.num_levels >= 1 && t.width[0] <= 0 && t.height[0] <= 0 && t.texels[0] != nullptr; };
            if (!constant(m.roughness) || !(m.use_vertex_color || constant(m.specular))) { candidate = false; break; }
            if (!m.use_vertex_color) items.push_back(exec::DownloadItem{&vals[4 * (size_t)i], m.specular.texels[0], 3 * sizeof(float)});
            else vals[4 * (size_t)i] = vals[4 * (size_t)i + 1] = vals[4 * (size_t)i + 2] = 0.f;
            items.push_back(exec::DownloadItem{&vals[4 * (size_t)i + 3], m.roughness.texels[0], sizeof(float)});
        }
        if (candidate) {
            exec::download_batch(items.data(), (int)items.size());
            for (int i = 0; i < num_materials; ++i)
                candidate = candidate && vals[4 * (size_t)i] == 0.f && vals[4 * (size_t)i + 1] == 0.f && vals[4 * (size_t)i + 2] == 0.f &&
                            vals[4 * (size_t)i + 3] > 0.02f;
        }
        s.diffuse_only = candidate;
    }
    s.lights.resize(num_area_lights);
    for (int i = 0; i < num_area_lights; ++i) {
        const rdr_area_light_desc &in = area_lights[i];
        if (in.shape_id < 0 || in.shape_id >= num_shapes) throw std::runtime_error("Scene: area light shape_id out of range");
        LightD &o = s.lights[i];
        o.shape_id = in.shape_id; o.two_sided = in.two_sided; o.directly_visible = in.directly_visible;
        for (int k = 0; k < 3; ++k) o.intensity[k] = in.intensity[k];
    }

    timer.lap("tables, host mirrors");
    // ---- light PMF / CDF, per-light area CDF (src/scene.cpp:38-61, 197-253) ----
    int num_lights = num_area_lights + (envmap ? 1 : 0);       // the environment light is the last entry (:197-202)
    s.light_pmf.assign(num_lights, 0); s.light_cdf.assign(num_lights, 0); s.light_areas.assign(num_area_lights, 0);
    s.area_cdf_offset.assign(num_area_lights, 0);
    {
        int total_tris = 0;
        for (int l = 0; l < num_area_lights; ++l) { s.area_cdf_offset[l] = total_tris; total_tris += s.shapes[s.lights[l].shape_id].num_triangles; }
        s.area_cdf_pool.assign(total_tris, 0);
        s.emitter_triangles = total_tris;
        double total_importance = 0;
        for (int l = 0; l < num_area_lights; ++l) {
            int sid = s.lights[l].shape_id;
            ShapeD hs = s.shapes[sid];     // host view of the same shape
            hs.vertices = s.h_vertices[sid].data(); hs.indices = s.h_indices[sid].data(); hs.geom = nullptr;
            double *cdf = s.area_cdf_pool.data() + s.area_cdf_offset[l];
            double total = 0;
            for (int t = 0; t < hs.num_triangles; ++t) { cdf[t] = tri_area(hs, t); total += cdf[t]; }
            double run = 0;
            for (int t = 0; t < hs.num_triangles; ++t) { double a = cdf[t]; cdf[t] = run; run += a; }
            for (int t = 0; t < hs.num_triangles; ++t) cdf[t] = cdf[t] / total;
            s.light_areas[l] = total;
            const float *I = s.lights[l].intensity;
            float lum = 0.212671f * I[0] + 0.715160f * I[1] + 0.072169f * I[2];   // fp32, as luminance<float>
            s.light_pmf[l] = total * lum * double(M_PI);
            total_importance += s.light_pmf[l];
        }
        if (envmap) {
            // bounding sphere of all vertices in fp32 (:161-195).  [quirk] the z extent is taken from y (:184,187)
            float inf = std::numeric_limits<float>::infinity();
            float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
            for (int i = 0; i < num_shapes; ++i) {
                float smn[3] = {inf, inf, inf}, smx[3] = {-inf, -inf, -inf};
                const std::vector<float> &v = s.h_vertices[i];
                for (size_t k = 0; k + 2 < v.size(); k += 3)
                    for (int c = 0; c < 3; ++c) { smn[c] = std::min(smn[c], v[k + c]); smx[c] = std::max(smx[c], v[k + c]); }
                mn[0] = std::min(smn[0], mn[0]); mn[1] = std::min(smn[1], mn[1]); mn[2] = std::min(smn[1], mn[2]);
                mx[0] = std::max(smx[0], mx[0]); mx[1] = std::max(smx[1], mx[1]); mx[2] = std::max(smx[1], mx[2]);
            }
            float radius = 0;
            if (num_shapes > 0) {
                float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
                radius = 0.5f * std::sqrt(dx * dx + dy * dy + dz * dz);
            }
            double r = radius;                 // Sphere::radius is a Real holding the fp32 result
            double surface_area = 4 * double(M_PI) * (r * r);
            int envmap_id = num_area_lights;
            s.light_pmf[envmap_id] = surface_area > 0 ? surface_area / (double)envmap->pdf_norm : 1.0;
            total_importance += s.light_pmf[envmap_id];
        }
        if (num_lights > 0) {
            if (!(total_importance > 0)) throw std::runtime_error("Scene: total light importance must be positive");
            for (int l = 0; l < num_lights; ++l) s.light_pmf[l] /= total_importance;
            s.light_cdf[0] = 0;
            for (int l = 1; l < num_lights; ++l) s.light_cdf[l] = s.light_cdf[l - 1] + s.light_pmf[l - 1];
        }
    }

    timer.lap("light tables");
    // ---- triangle hierarchy: built on other threads while this one prepares the tables and the edge list ----
    std::vector<rt::MeshView> meshes(num_shapes);
    for (int i = 0; i < num_shapes; ++i)
        meshes[i] = rt::MeshView{s.h_vertices[i].data(), s.h_indices[i].data(), s.shapes[i].num_triangles};
    // A Scene with the index buffers of the previous one (an optimisation loop moves vertices, pyredner builds a Scene per
    // forward call, render_pytorch.py:609) keeps that hierarchy's topology and refits its boxes; hits do not depend on the
    // hierarchy (raytri.h), and it is rebuilt once its inner surface area has grown by more than 30 %.  RDR_BUILD_NO_REFIT: always build.
    // On the GPU build the hierarchy never exists on the host: bvh_gpu.cpp builds it from the caller's device arrays, or --
    // same connectivity as the last build -- refits a copy of that build's records (below, once the shape table is uploaded).
    TopologyCache *topo_cache = topology_cache(gpu_index);           // guarded by the device's API lock (capi.cpp)
    const bool refit_allowed = !(s.build_flags & RDR_BUILD_NO_REFIT);
    rt::BvhHost bvh_built;
    auto bvh_job = hostpool::run([&meshes, &bvh_built, &s, refit_allowed, topo_cache] {      // (joins in its destructor on an early exit)
        if (exec::kDeviceBvh) return;
        bool same = refit_allowed && topo_cache->indices.size() == s.h_indices.size() && !topo_cache->bvh.nodes.empty();
        for (size_t i = 0; same && i < s.h_indices.size(); ++i) same = topo_cache->indices[i] == s.h_indices[i];
        if (same) {
            bvh_built = topo_cache->bvh;
            if (rt::refit_bvh(bvh_built, meshes) <= 1.3) return;
        }
        bvh_built = rt::build_bvh(meshes);
        topo_cache->indices = s.h_indices;
        topo_cache->bvh = bvh_built;
    });

    // ---- device copies of the flat tables ----
    // ---- gathered per-triangle records (scene_data.h: TriGeomD) ----
    {
        size_t total = 0;
        for (int i = 0; i < num_shapes; ++i) total += (size_t)s.shapes[i].num_triangles;
        std::vector<TriGeomD> geom(total);
        std::vector<size_t> first(num_shapes, 0);
        size_t at = 0;
        for (int i = 0; i < num_shapes; ++i) {
            first[i] = at;
            const ShapeD &sh = s.shapes[i];
            const std::vector<float> &vtx = s.h_vertices[i], &uvs = s.h_uvs[i], &nrm = s.h_normals[i];
            const std::vector<int> &idx = s.h_indices[i], &uidx = s.h_uv_indices[i], &nidx = s.h_normal_indices[i];
            const int nuv = (int)uvs.size() / 2, nn = (int)nrm.size() / 3;
            for (int t = 0; t < sh.num_triangles; ++t, ++at) {
                TriGeomD &g = geom[at];
                std::memset(&g, 0, sizeof(g));
                for (int k = 0; k < 3; ++k) {
                    const int vi = idx[3 * t + k];
                    const int ui = sh.uv_indices ? uidx[3 * t + k] : vi;
                    const int ni = sh.normal_indices ? nidx[3 * t + k] : vi;
                    g.vi[k] = vi; g.ui[k] = ui; g.ni[k] = ni;
                    for (int c = 0; c < 3; ++c) g.p[3 * k + c] = vtx[3 * (size_t)vi + c];
                    if (sh.uvs) {
                        if (ui < 0 || ui >= nuv) throw std::runtime_error("Scene: uv index out of range in shape " + std::to_string(i));
                        g.uv[2 * k] = uvs[2 * (size_t)ui]; g.uv[2 * k + 1] = uvs[2 * (size_t)ui + 1];
                    }
                    if (sh.normals) {
                        if (ni < 0 || ni >= nn) throw std::runtime_error("Scene: normal index out of range in shape " + std::to_string(i));
                        for (int c = 0; c < 3; ++c) g.n[3 * k + c] = nrm[3 * (size_t)ni + c];
                    }
                }
            }
        }
        const TriGeomD *d_geom = to_device(s, geom.data(), geom.size());
        for (int i = 0; i < num_shapes; ++i) s.shapes[i].geom = s.shapes[i].num_triangles > 0 ? d_geom + first[i] : nullptr;
    }
    s.d.shapes = to_device(s, s.shapes.data(), s.shapes.size());
    s.d.materials = to_device(s, s.materials.data(), s.materials.size());
    s.d.lights = to_device(s, s.lights.data(), s.lights.size());
    s.d.envmap = nullptr;
    s.d.no_diffs = 0;
    s.d.plain_materials = 0;
    if (envmap) {
        EnvmapD e;
        std::memset(&e, 0, sizeof(e));
        e.values = convert_tex(envmap->values);
        e.values.channels = 3;
        if (e.values.num_levels < 1 || e.values.width[0] <= 0 || e.values.height[0] <= 0)
            throw std::runtime_error("Scene: the environment map needs a 2-D RGB texture");
        if (!envmap->env_to_world || !envmap->world_to_env || !envmap->sample_cdf_ys || !envmap->sample_cdf_xs)
            throw std::runtime_error("Scene: environment map transforms and sampling tables are required");
        e.env_to_world = m4_from(envmap->env_to_world);
        e.world_to_env = m4_from(envmap->world_to_env);
        e.sample_cdf_ys = envmap->sample_cdf_ys; e.sample_cdf_xs = envmap->sample_cdf_xs;
        e.pdf_norm = (double)envmap->pdf_norm;
        e.directly_visible = envmap->directly_visible;
        if (e.values.num_levels > 1) s.has_mipmaps = true;
        s.h_envmap = e;
        s.d.envmap = to_device(s, &s.h_envmap, 1);
    }
    s.d.num_shapes = num_shapes; s.d.num_materials = num_materials;
    s.d.num_area_lights = num_area_lights; s.d.num_lights = num_lights;
    s.d.light_pmf = to_device(s, s.light_pmf.data(), s.light_pmf.size());
    s.d.light_cdf = to_device(s, s.light_cdf.data(), s.light_cdf.size());
    s.d.light_areas = to_device(s, s.light_areas.data(), s.light_areas.size());
    s.d.area_cdf_pool = to_device(s, s.area_cdf_pool.data(), s.area_cdf_pool.size());
    s.d.area_cdf_offset = to_device(s, s.area_cdf_offset.data(), s.area_cdf_offset.size());
    s.sobol_table = (const uint64_t *)exec::device_constant(rdr_sobol_table, sizeof(uint64_t) * (size_t)kSobolTableWords);
    s.ltc_table = (const float *)exec::device_constant(rdr_ltc_table, sizeof(float) * (size_t)128 * 128 * 9);

    timer.lap("device copies");
    // ---- edge sampling structures ----
    if (s.use_primary_edges || s.use_secondary_edges) {
        const bool sync_edges = (s.build_flags & RDR_BUILD_SYNC_EDGES) != 0;
        // Everything the edge structures are computed from (edges.cpp: compute_edge_data): if it equals what the previous
        // Scene's were computed from, that result -- finished or still in the builder's hands -- is this Scene's too.
        EdgeCache *cache = edge_cache(gpu_index);             // guarded by the device's API lock (capi.cpp)
        const bool cache_allowed = !(s.build_flags & (RDR_BUILD_NO_EDGE_CACHE | RDR_BUILD_NO_REFIT));
        const bool hit = cache_allowed && cache->result.valid() && cache->gpu_index == s.gpu_index &&
                         cache->primary == s.use_primary_edges && cache->secondary == s.use_secondary_edges &&
                         std::memcmp(&cache->cam, &s.d.cam, sizeof(CameraD)) == 0 && cache->indices == s.h_indices &&
                         cache->vertices == s.h_vertices && cache->normals == s.h_normals;
        if (hit) {
            s.edge_build = cache->result;
        } else {
            const Scene *sc = &s;
            s.edge_build = EdgeBuilder::get().submit([sc]() -> std::shared_ptr<EdgeData> {
                EdgeData *ed = compute_edge_data(*sc);
                try {
                    exec::select_device(1, sc->gpu_index);
                    exec::StreamScope on(exec::side_stream(0));             // this thread's own stream
                    publish_edge_data(*ed);
                    exec::upload_flush();
                } catch (...) {
                    exec::device_sync();          // the builder's kernels may still be running: no block returns to the pool before
                    delete_edge_data(ed);
                    throw;
                }
                return std::shared_ptr<EdgeData>(ed, [](EdgeData *e) { delete_edge_data(e); });
            }).share();
            if (cache_allowed) {
                cache->gpu_index = s.gpu_index; cache->primary = s.use_primary_edges; cache->secondary = s.use_secondary_edges;
                std::memcpy(&cache->cam, &s.d.cam, sizeof(CameraD));      // (padding too: the comparison above is a memcmp)
                cache->vertices = s.h_vertices; cache->normals = s.h_normals; cache->indices = s.h_indices;
                cache->result = s.edge_build;
            }
        }
        if (sync_edges || timer.on) s.edge_data();
    }
    timer.lap("edge structures");
    if (exec::kDeviceBvh) {
        bvh_job.wait();
        // {vertices, indices} per shape, for the kernels that read the caller's arrays
        std::vector<const void *> refs((size_t)2 * std::max(num_shapes, 1), nullptr);
        size_t total = 0;
        for (int i = 0; i < num_shapes; ++i) { refs[2 * i] = s.shapes[i].vertices; refs[2 * i + 1] = s.shapes[i].indices; total += (size_t)s.shapes[i].num_triangles; }
        const void *d_refs = to_device(s, refs.data(), refs.size());
        // (a cached EMPTY hierarchy -- no shapes / no triangles -- is never refitted: there is nothing to gather and a zero-size
        //  launch is an error; the build below returns at once for it)
        bool same = refit_allowed && total > 0 && topo_cache->dev && topo_cache->dev->num_slots > 0 && topo_cache->gpu_index == s.gpu_index &&
                    topo_cache->indices.size() == s.h_indices.size();
        for (size_t i = 0; same && i < s.h_indices.size(); ++i) same = topo_cache->indices[i] == s.h_indices[i];
        auto dev = std::make_shared<rt::BvhDev>();
        bool built = false;
        if (same) {
            rt::refit_tri_bvh_device(*topo_cache->dev, d_refs, *dev);
            dev->parent = topo_cache->dev;
            double area = 0;
            exec::upload_flush();                    // (the shape table above travels through the staging buffer)
            exec::download(&area, dev->area, sizeof(double));
            if (!(dev->inner_area > 0) || area / dev->inner_area > 1.3) dev = std::make_shared<rt::BvhDev>();      // gone stale: build
            else built = true;
        }
        if (!built) {
            std::vector<int> prim_ids;
            prim_ids.reserve(2 * total);
            for (int i = 0; i < num_shapes; ++i)
                for (int t = 0; t < s.shapes[i].num_triangles; ++t) { prim_ids.push_back(i); prim_ids.push_back(t); }
            rt::BvhBuildParams prm;
            if (const char *e = std::getenv("RDR_BVH_BINS")) prm.bins = std::min(64, std::max(2, std::atoi(e)));
            if (const char *e = std::getenv("RDR_BVH_TCOST")) prm.trav_cost = (float)std::atof(e);
            if (const char *e = std::getenv("RDR_BVH_LEAF")) prm.leaf_max = std::min(4, std::max(1, std::atoi(e)));
            rt::build_tri_bvh_device(d_refs, prim_ids.data(), (int)total, prm, *dev);
            topo_cache->indices = s.h_indices;
            topo_cache->dev = dev;
            topo_cache->gpu_index = s.gpu_index;
        }
        s.bvh_dev = dev;
        s.bvh = dev->view();
        timer.lap("triangle hierarchy (device)");
    } else {
        bvh_job.wait();
        s.bvh_host = std::move(bvh_built);
        s.bvh.num_nodes = (int)s.bvh_host.nodes.size();
        s.bvh.num_tris = (int)s.bvh_host.ids.size() / 2;
        s.bvh.stack_need = s.bvh_host.depth + 2;
        s.bvh.nodes = to_device(s, s.bvh_host.nodes.data(), s.bvh_host.nodes.size());
        s.bvh.tris = to_device(s, s.bvh_host.tris.data(), s.bvh_host.tris.size());
        s.bvh.ids = to_device(s, s.bvh_host.ids.data(), s.bvh_host.ids.size());
        s.bvh.num_wide = (int)s.bvh_host.wide.size();
        s.bvh.wide_stack_need = s.bvh_host.wide_stack_need;
        s.bvh.wide = s.bvh.num_wide > 0 ? to_device(s, s.bvh_host.wide.data(), s.bvh_host.wide.size()) : nullptr;
        timer.lap("triangle hierarchy (wait)");
    }
    exec::upload_flush();
    timer.lap("upload flush");
    return sp.release();
}

} // namespace rdr
