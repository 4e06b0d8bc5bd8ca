// scene_data.h -- flat, pointer-based scene description shared by the host driver and the kernels.
//
// These PODs mirror what crosses the reference's binding boundary (src/redner.cpp:34-181):
// caller-owned fp32/int32 arrays plus a few scalars.  The fields keep the reference's names so a
// reader can map them 1:1 (Shape: src/shape.h:9-61; Texture: src/texture.h:14-47; Material:
// src/material.h:12-99; AreaLight: src/area_light.h:8-38; Camera: src/camera.h:19-84).
// Pointers in ShapeD/TexD/EnvmapD address the memory space the kernels run in (HBM for the gfx950
// build); CameraD and light intensities are copied by value at Scene construction, as in the
// reference (src/camera.h:44-65, src/area_light.h:18-20).
//
// Gradient targets (G*) are fp64 accumulators owned by the renderer; they are folded into the
// caller's fp32 gradient tensors once at the end of render() (see render.cpp: flush_gradients).
#pragma once
#include "vecmath.h"

namespace rdr {

constexpr int kMaxMip = 8;   // src/texture.h:11

// Everything the shading stages read about one triangle, gathered at Scene construction (scene.cpp) so that a lane
// reaches it with one dependent fetch of nine aligned 16-byte loads instead of the index -> vertex / uv / normal chains
// through five arrays.  The values are copies of the mesh arrays' floats, so results are unchanged.
struct alignas(16) TriGeomD {
    float p[9]; int vi[3];          // corners, vertex indices
    float n[9]; int ni[3];          // shading normals (zero when the shape has none), their indices
    float uv[6]; int ui[3]; int pad[3];
};
static_assert(sizeof(TriGeomD) == 144, "triangle record size");

struct ShapeD {
    const TriGeomD *geom;           // per triangle; nullptr in host-side views of a shape (they read the mesh arrays)
    const float *vertices;
    const int *indices;
    const float *uvs;
    const float *normals;
    const int *uv_indices;
    const int *normal_indices;
    const float *colors;
    int num_vertices, num_uv_vertices, num_normal_vertices, num_triangles;
    int material_id, light_id;
};

struct TexD {
    const float *texels[kMaxMip];
    int width[kMaxMip], height[kMaxMip];
    int channels;     // 1, 3 or N (generic)
    int num_levels;   // 0 = absent
    const float *uv_scale;
};

struct MaterialD {
    TexD diffuse, specular, roughness, generic, normal_map;
    int compute_specular_lighting, two_sided, use_vertex_color;
};

struct LightD {
    int shape_id;
    float intensity[3];
    int two_sided, directly_visible;
};

enum CameraKind { CAM_PERSPECTIVE = 0, CAM_ORTHOGRAPHIC = 1, CAM_FISHEYE = 2, CAM_PANORAMA = 3 };

struct DistortD { int defined; double k[6], p[2]; };    // Brown-Conrady lens distortion (src/camera_distortion.h:7-11)
struct CameraD {
    int width, height;
    int use_look_at;
    V3 position, look, up;
    M4 cam_to_world, world_to_cam;
    M3 intrinsic_mat_inv, intrinsic_mat;
    float clip_near;
    int kind;
    int vp_x0, vp_y0, vp_x1, vp_y1;   // viewport_beg / viewport_end
    DistortD distortion;
    // Several samples of a small frame rendered as ONE set of lanes (render.cpp "sample batches"): lane v is pixel
    // v % (pixels of the viewport) of the batch's sample v / pixels; > 0: the viewport's row count.  0: one sample per launch.
    int batch_rows = 0;
};

struct EnvmapD {
    TexD values;
    M4 env_to_world, world_to_env;
    const float *sample_cdf_ys, *sample_cdf_xs;
    double pdf_norm;
    int directly_visible;
};

// Everything a stage kernel needs to shade: passed by value inside the stage functors.
struct SceneD {
    CameraD cam;
    const ShapeD *shapes;
    const MaterialD *materials;
    const LightD *lights;
    const EnvmapD *envmap;         // null when absent
    int num_shapes, num_materials, num_area_lights, num_lights;   // num_lights counts the envmap
    const double *light_pmf, *light_cdf, *light_areas;
    const double *area_cdf_pool;   // concatenated per-light triangle CDFs
    const int *area_cdf_offset;    // per area light: start inside the pool
    int no_diffs;                  // ray differentials cannot influence the result (set by the lean stages only)
    int plain_materials;           // every texture is a constant, no normal maps (set by the lean stages only)
};

// ---- gradient accumulators --------------------------------------------------------------------
struct GShape { double *vertices, *uvs, *normals, *colors; };
struct GTex { double *texels[kMaxMip]; double *uv_scale; };
struct GMaterial { GTex diffuse, specular, roughness, generic, normal_map; };
struct GCamera {
    double *position, *look, *up;          // 3 each (look-at parameterisation)
    double *cam_to_world, *world_to_cam;   // 16 each
    double *intrinsic_mat_inv, *intrinsic_mat;   // 9 each
    double *distortion;                          // 8: k0..k5, p0, p1
};
struct GEnvmap { GTex values; double *world_to_env; };
struct GScene {
    GShape *shapes;
    GMaterial *materials;
    double *light_intensity;   // 3 per area light
    GCamera cam;
    GEnvmap *envmap;
};

} // namespace rdr
