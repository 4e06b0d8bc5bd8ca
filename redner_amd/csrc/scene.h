// scene.h -- host-side Scene object behind rdr_scene_create().
//
// Counterpart of the reference's Scene constructor (src/scene.cpp:63-307): copies the PODs, keeps
// the caller's data pointers, and builds the acceleration/sampling structures once:
//   * the triangle hierarchy of bvh.h (replaces the Embree / OptiX Prime scene),
//   * light PMF/CDF and per-light area CDFs (src/scene.cpp:38-61, 197-253),
//   * the edge list, its PMF/CDF and the two edge hierarchies (edges.h; src/edge.cpp:233-383,
//     src/edge_tree.cpp:724-882).
// Geometry is read back from HBM once for those builds (GPU-side build/refit is section 8f row 1).
#pragma once
#include "../../include/redner_amd.h"
#include "bvh.h"
#include "bvh_gpu.h"
#include "scene_data.h"
#include <string>
#include <vector>
#include <future>
#include <memory>
#include <mutex>

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace rdr {
struct PhaseTimer {         // RDR_DEBUG_DUMP: wall time of the scene-build phases on stderr
    const char *group;
    const bool on = std::getenv("RDR_DEBUG_DUMP") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit PhaseTimer(const char *g) : group(g) {}
    void lap(const char *what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[redner_amd] %s: %-24s %7.2f ms\n", group, what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}

namespace rdr {

struct EdgeData;   // edges.h

struct Scene {
    int gpu_index = 0;
    unsigned build_flags = 0;      // rdr_build_flags in force when the Scene was created (tuning.h)
    bool use_primary_edges = false, use_secondary_edges = false;
    int max_generic_texture_dimension = 0;
    bool has_textures = false;     // some reflectance / roughness is an image, or a normal map is present
    bool has_vertex_colors = false;
    bool has_mipmaps = false;
    int emitter_triangles = 0;     // triangles of all area-light shapes together (render.cpp: last-bounce emitter test)
    bool diffuse_only = false;     // every material: constant specular reflectance (0, 0, 0) -- see render.cpp: run_sample
    EnvmapD h_envmap;              // valid when d.envmap != nullptr      // some texture has > 1 level, i.e. ray differentials influence results

    // host mirrors
    std::vector<ShapeD> shapes;
    std::vector<MaterialD> materials;
    std::vector<LightD> lights;
    std::vector<std::vector<float>> h_vertices, h_uvs, h_normals;
    std::vector<std::vector<int>> h_indices, h_uv_indices, h_normal_indices;
    std::vector<double> light_pmf, light_cdf, light_areas, area_cdf_pool;
    std::vector<int> area_cdf_offset;
    rt::BvhHost bvh_host;                        // CPU debugging harness only: the GPU build keeps the hierarchy on the device
    std::shared_ptr<rt::BvhDev> bvh_dev;         // (bvh_gpu.cpp)

    // device view
    SceneD d;
    rt::BvhD bvh;
    const uint64_t *sobol_table = nullptr;   // 1024 x 52 u64
    const float *ltc_table = nullptr;        // 128 x 128 x 9 f32
    // Edge-sampling structures.  Only a gradient render reads them, so create_scene() hands their build to the edge-builder
    // thread (scene.cpp) and returns: the host part (edge list, PMF, billboard hierarchy refit, per-edge records) on the
    // pool, then the device copies and the kernels that build the two order-exact hierarchies (edges_gpu.cpp) on that
    // thread's own stream.  The forward render and whatever the caller does before its backward call run beside all of it;
    // edge_data() joins (the first gradient render calls it), and so does the destructor.  (The reference builds them inside
    // the Scene constructor, src/scene.cpp:63-307, i.e. in front of every forward render, pyredner/render_pytorch.py:609.)
    // RDR_SYNC_EDGES=1 / RDR_DEBUG_DUMP: joined inside create_scene().  RDR_EDGE_HOST_BUILD=1: hierarchies by the host builder.
    // The structures depend on positions, connectivity, shading normals and the camera only: a Scene whose inputs equal the
    // previous Scene's in all of those (a loop that moves materials, lights or textures) SHARES that Scene's structures --
    // or its build, if that is still running -- instead of building them again (scene.cpp: EdgeCache; RDR_NO_EDGE_CACHE=1 /
    // RDR_NO_REFIT=1: always build).
    const EdgeData *edge_data() const;
    mutable EdgeData *edges = nullptr;            // valid after edge_data()
    mutable std::shared_ptr<EdgeData> edges_ref;  // keeps `edges` alive (shared with the cache and with other Scenes)
    mutable std::shared_future<std::shared_ptr<EdgeData>> edge_build;   // pending (possibly shared) build, if any
    mutable std::mutex edge_join;                 // edge_data() may be called by several sample workers at once

    std::vector<void *> owned;   // device allocations released in the destructor
    ~Scene();
};

void drop_edge_cache();          // releases the edge structures kept for the next Scene (rdr_trim_cache)
Scene *create_scene(const rdr_camera_desc *camera,
                    const rdr_shape_desc *shapes, int num_shapes,
                    const rdr_material_desc *materials, int num_materials,
                    const rdr_area_light_desc *area_lights, int num_area_lights,
                    const rdr_envmap_desc *envmap,
                    int use_gpu, int gpu_index, int primary_edges, int secondary_edges);

// Channel layout (src/channels.cpp:54-113).  -1 on an unknown channel.
int compute_num_channels(const int *channels, int n, int max_generic_texture_dimension);

} // namespace rdr
