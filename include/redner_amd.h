/* redner_amd.h -- C ABI of the MI355X-native differentiable path tracer.
 *
 * This is the drop-in boundary for the one hot path of BachiLi/redner: everything the reference's
 * pybind11 module `redner` (src/redner.cpp:20-272) exposes for rendering, expressed as plain C
 * structs, raw pointers and sizes -- no torch / pybind types.  The Python module
 * redner_amd/redner.py re-creates the reference's class surface (redner.Camera, redner.Shape,
 * redner.Scene, redner.render, ...) on top of these entry points via ctypes, so the unmodified
 * pyredner/render_pytorch.py runs against it.
 *
 * Pointer conventions are the reference's own (src/ptr.h:10-24: raw addresses, no ownership, no
 * size, 0 = absent): `dev` pointers address GPU memory of device `gpu_index` (torch CUDA/HIP
 * tensors' data_ptr()), `host` pointers address CPU memory and are read during the call that
 * receives them.  All outputs are caller-owned and ACCUMULATED into (+=), never overwritten
 * (src/primary_contribution.cpp:39-43, src/atomic.h:43-141).
 *
 * Error handling: the reference aborts (assert/exit(1), src/cuda_utils.h:12-16).  Here every
 * entry point that can fail returns NULL / non-zero and writes a message retrievable with
 * rdr_last_error(); the Python layer raises RuntimeError.  There is NO CPU fallback: creating a
 * scene with use_gpu == 0, or without a usable gfx950 device, is an error.
 */
#ifndef REDNER_AMD_H
#define REDNER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDR_MAX_MIP_LEVELS 8 /* src/texture.h:11 max_num_texels */

/* enum values follow the declaration order of the reference's enums */
enum rdr_camera_type { RDR_CAMERA_PERSPECTIVE = 0, RDR_CAMERA_ORTHOGRAPHIC = 1, RDR_CAMERA_FISHEYE = 2,
                       RDR_CAMERA_PANORAMA = 3 };                 /* src/camera.h:12-17   */
enum rdr_sampler_type { RDR_SAMPLER_INDEPENDENT = 0, RDR_SAMPLER_SOBOL = 1 }; /* src/pathtracer.h:11-14 */
enum rdr_channel {                                                /* src/channels.h:6-23  */
    RDR_CH_RADIANCE = 0, RDR_CH_ALPHA, RDR_CH_DEPTH, RDR_CH_POSITION, RDR_CH_GEOMETRY_NORMAL,
    RDR_CH_SHADING_NORMAL, RDR_CH_UV, RDR_CH_BARYCENTRIC_COORDINATES, RDR_CH_DIFFUSE_REFLECTANCE,
    RDR_CH_SPECULAR_REFLECTANCE, RDR_CH_ROUGHNESS, RDR_CH_GENERIC_TEXTURE, RDR_CH_VERTEX_COLOR,
    RDR_CH_SHAPE_ID, RDR_CH_TRIANGLE_ID, RDR_CH_MATERIAL_ID
};

/* redner.Camera(...)  src/redner.cpp:34-50, src/camera.h:19-84.  All pointers HOST, read at
 * scene creation.  cam_to_world != NULL selects the matrix parameterisation (use_look_at = 0). */
typedef struct rdr_camera_desc {
    int width, height;
    const float *position, *look, *up;          /* 3 floats each, or NULL */
    const float *cam_to_world, *world_to_cam;   /* 16 floats row-major, or NULL */
    const float *intrinsic_mat_inv, *intrinsic_mat; /* 9 floats row-major */
    const float *distortion_params;             /* 8 floats or NULL */
    float clip_near;
    int camera_type;                            /* rdr_camera_type */
    int viewport_beg[2], viewport_end[2];       /* (x, y) */
} rdr_camera_desc;

/* redner.DCamera(...)  src/redner.cpp:52-60.  DEV pointers to fp32 gradient buffers (or NULL). */
typedef struct rdr_dcamera_desc {
    float *position, *look, *up, *cam_to_world, *world_to_cam, *intrinsic_mat_inv, *intrinsic_mat,
          *distortion_params;
} rdr_dcamera_desc;

/* redner.Shape(...)  src/redner.cpp:84-104, src/shape.h:9-61.  DEV pointers. */
typedef struct rdr_shape_desc {
    const float *vertices;      /* [num_vertices, 3] */
    const int32_t *indices;     /* [num_triangles, 3] */
    const float *uvs;           /* [num_uv_vertices, 2] or NULL */
    const float *normals;       /* [num_normal_vertices, 3] or NULL */
    const int32_t *uv_indices;  /* or NULL */
    const int32_t *normal_indices; /* or NULL */
    const float *colors;        /* [num_vertices, 3] or NULL */
    int num_vertices, num_uv_vertices, num_normal_vertices, num_triangles;
    int material_id, light_id;
} rdr_shape_desc;

/* redner.DShape(vertices, uvs, normals, colors)  src/redner.cpp:106-110.  DEV fp32. */
typedef struct rdr_dshape_desc { float *vertices, *uvs, *normals, *colors; } rdr_dshape_desc;

/* redner.Texture1/3/N(...)  src/redner.cpp:112-131, src/texture.h:14-47.  DEV pointers.
 * A constant texture has num_levels = 1 and width[0] = height[0] = 0. */
typedef struct rdr_texture_desc {
    const float *texels[RDR_MAX_MIP_LEVELS];
    int width[RDR_MAX_MIP_LEVELS], height[RDR_MAX_MIP_LEVELS];
    int channels;       /* 1, 3, or N for the generic texture */
    int num_levels;     /* 0 = texture absent */
    const float *uv_scale; /* 2 floats */
} rdr_texture_desc;

/* redner.Material(...)  src/redner.cpp:133-151 */
typedef struct rdr_material_desc {
    rdr_texture_desc diffuse_reflectance, specular_reflectance, roughness, generic_texture, normal_map;
    int compute_specular_lighting, two_sided, use_vertex_color;
} rdr_material_desc;

/* redner.DMaterial(...)  src/redner.cpp:153-158: same shape, texel pointers are fp32 gradients */
typedef struct rdr_dtexture_desc {
    float *texels[RDR_MAX_MIP_LEVELS];
    int num_levels;
    float *uv_scale;
} rdr_dtexture_desc;
typedef struct rdr_dmaterial_desc {
    rdr_dtexture_desc diffuse_reflectance, specular_reflectance, roughness, generic_texture, normal_map;
} rdr_dmaterial_desc;

/* redner.AreaLight(shape_id, intensity, two_sided, directly_visible)  src/redner.cpp:160-164.
 * intensity is copied (the reference reads its HOST pointer in the constructor). */
typedef struct rdr_area_light_desc {
    int shape_id;
    float intensity[3];
    int two_sided, directly_visible;
} rdr_area_light_desc;
/* redner.DAreaLight(intensity)  src/redner.cpp:166-167.  DEV fp32[3]. */
typedef struct rdr_darea_light_desc { float *intensity; } rdr_darea_light_desc;

/* redner.EnvironmentMap(...)  src/redner.cpp:169-177.  values/cdfs DEV, matrices HOST. */
typedef struct rdr_envmap_desc {
    rdr_texture_desc values;
    const float *env_to_world, *world_to_env;   /* 16 floats, HOST */
    const float *sample_cdf_ys, *sample_cdf_xs; /* DEV */
    float pdf_norm;
    int directly_visible;
} rdr_envmap_desc;
typedef struct rdr_denvmap_desc { rdr_dtexture_desc values; float *world_to_env; } rdr_denvmap_desc;

/* redner.RenderOptions(seed, num_samples, max_bounces, channels, sampler_type, sample_pixel_center)
 * src/redner.cpp:207-216, src/pathtracer.h:16-23.
 * Extension for multi-GPU sample sharding (no reference counterpart): this
 * call renders Sobol' samples [sample_offset, sample_offset + num_samples) of a total_samples-spp
 * estimate, i.e. with weight 1/total_samples.  total_samples == 0 means "num_samples". */
typedef struct rdr_tuning rdr_tuning;
typedef struct rdr_render_options {
    uint64_t seed;
    int num_samples, max_bounces;
    const int *channels; int num_channels;   /* rdr_channel values */
    int sampler_type;                        /* rdr_sampler_type */
    int sample_pixel_center;
    int sample_offset, total_samples;
    const rdr_tuning *tuning;                /* or NULL = every default (below) */
} rdr_render_options;

/* How a render() call is scheduled on the GPU -- which kernels, how many samples per launch, how many host threads.  No
 * reference counterpart (its RenderOptions stop at sample_pixel_center, src/pathtracer.h:16-23); results do not depend on any
 * of this beyond the order of floating-point atomics.  EVERY field: 0 = the library's default, so a zeroed struct (or a NULL
 * pointer) is the shipped configuration.  These fields replace the RDR_* environment switches of earlier rounds for
 * everything that selects a kernel or a schedule; a variable that is still set supplies the default of a field that is 0
 * (A/B scripts), the field wins. */
enum rdr_tune_flags {
    RDR_TUNE_NO_OVERLAP       = 1 << 0,   /* every stage on the calling stream (no side streams)              RDR_NO_OVERLAP */
    RDR_TUNE_FORCE_GENERAL    = 1 << 1,   /* no stage specialisation (lean / mid kernels)                     RDR_FORCE_GENERAL */
    RDR_TUNE_PICKN_WALK       = 1 << 2,   /* NEE-mode edge pick: reference-order walk for every slot          RDR_PICKN_WALK */
    RDR_TUNE_PICKH_FUSED      = 1 << 3,   /* hierarchical edge pick: the one-loop form                        RDR_PICKH_FUSED */
    RDR_TUNE_PICKH_LAZY       = 1 << 4,   /* ... per-field node loads                                         RDR_PICKH_LAZY */
    RDR_TUNE_NO_HOIST         = 1 << 5,   /* first-vertex edge picks inside the backward sweep                RDR_NO_HOIST */
    RDR_TUNE_REFILL_OFF       = 1 << 6,   /* never the refilling traversal kernel                             RDR_TRACE_REFILL=0 */
    RDR_TUNE_REFILL_ALL       = 1 << 7,   /* the refilling traversal kernel on every queue                    RDR_TRACE_REFILL_ALL */
    RDR_TUNE_TRACE_BINARY     = 1 << 8,   /* never the 4-wide node records                                    RDR_TRACE_BINARY */
    RDR_TUNE_TRACE_NO_LDS_TOP = 1 << 9,   /* hierarchy top not staged in LDS                                  RDR_TRACE_NO_LDS_TOP */
    RDR_TUNE_NO_FUSED_BOUNCE  = 1 << 10,  /* BounceContrib(d) and BounceSample(d+1) as two launches          RDR_NO_FUSED_BOUNCE */
    RDR_TUNE_PICKH_ONE_LAUNCH = 1 << 11,  /* hierarchical edge pick: one slot per lane, one launch (r1-r5)   RDR_PICKH_ONE_LAUNCH */
    RDR_TUNE_NO_NEE_COMPACT   = 1 << 12,  /* bounce adjoints without list compactions (next-event half over
                                           * the whole live-lane list, continuation half over the next
                                           * depth's list as it is)                                            RDR_NO_NEE_COMPACT */
    RDR_TUNE_LARGE_FORMS      = 1 << 13,  /* the stage forms of large frames (split pick, compacted adjoint
                                           * lists) at every size; default: from 2^19 lanes per launch set     RDR_LARGE_FRAME_FORMS */
    RDR_TUNE_TRACE_EVERY_CONTINUATION = 1 << 14   /* the last bounce's continuation rays are all traced (default in a
                                           * plain scene with <= 8 emitter triangles: those that meet no
                                           * emitter triangle are answered "no hit" untraced)                  RDR_TRACE_EVERY_CONTINUATION */
};
struct rdr_tuning {
    unsigned flags;                 /* rdr_tune_flags */
    int batch_samples;              /* most samples rendered as one set of lanes; 1 = one sample per launch (default 16)   RDR_BATCH */
    int64_t batch_lanes;            /* most lanes of such a set (default 2^24; 2^22 when another allocator holds > 10 % of
                                     * the device's memory)                                                                  RDR_BATCH_LANES */
    int workers;                    /* host threads that drive the batches of a gradient render (default: by size)         RDR_WORKERS */
    int refill_rays_per_lane, refill_idle_lanes, refill_steps;   /* trace_refill_kernel (4, 24, 4)                          RDR_TRACE_REFILL=k,idle,steps */
    int wide_max_rays;              /* queues of up to this many rays walk the 4-wide records (2^19)                       RDR_WIDE_MAX */
    int gather_budget;              /* pops per lane of SecEdgeGatherN before it hands over (256)                          RDR_GATHER_BUDGET */
    int gather_heavy_cap_plus1, gather_work_cap_plus1;   /* list capacities of the gather's hand-over paths, + 1 (tests: 1 = capacity 0)   RDR_GATHER_CAPS */
    int mem_available_mb;           /* size the batches as if this much device memory were free (tests)                    RDR_MEM_AVAILABLE_MB */
    int refill_order;               /* order in which trace_refill_kernel hands a wave's 256 rays out: 1 = queue order (rounds
                                     * 3-5), 2 = by direction octant (default), 3 = octant x dominant axis               RDR_REFILL_SORT=0|1|2 */
    int pickh_slots_per_lane, pickh_idle_lanes, pickh_steps;     /* the hierarchical pick's descent walk (1, 8, 8)           RDR_PICKH_REFILL=k,idle,steps */
};

/* Library-wide settings (no reference counterpart).
 * rdr_set_stream: launches of later rdr_scene_create / rdr_render / rdr_scene_trace calls made by THIS host thread are
 *   ordered on `hip_stream` (a hipStream_t; NULL = the null stream, the default) -- a caller whose tensors are produced on
 *   a non-default stream (torch.cuda.stream(...)) passes that stream and needs no device-wide synchronisation of its own.
 *   The calls still return synchronised (like the reference, src/pathtracer.cpp:947-949).
 * rdr_set_pool_cap_mb: bound of the buffer cache per device (see rdr_trim_cache), default min(a quarter of the device, 8 GiB)
 *   or RDR_POOL_CAP_MB; a dedicated render process may raise it so that the ~48 GB of a 2^24-lane sample batch stay parked
 *   between calls.  Negative = back to the default.
 * rdr_set_build_flags: rdr_build_flags for later rdr_scene_create calls (debugging / tests). */
void rdr_set_stream(void *hip_stream);
void rdr_set_pool_cap_mb(int64_t megabytes);
int64_t rdr_get_pool_cap_mb(void);      /* the bound in effect on the calling thread's current device */
enum rdr_build_flags {
    RDR_BUILD_NO_REFIT        = 1 << 0,   /* no topology caches: hierarchies built from scratch every Scene   RDR_NO_REFIT */
    RDR_BUILD_NO_EDGE_CACHE   = 1 << 1,   /* edge structures never shared between Scenes                      RDR_NO_EDGE_CACHE */
    RDR_BUILD_SYNC_EDGES      = 1 << 2,   /* edge structures built inside rdr_scene_create                    RDR_SYNC_EDGES */
    RDR_BUILD_EDGE_HOST_BUILD = 1 << 3    /* edge hierarchies by the host builder                             RDR_EDGE_HOST_BUILD */
};
void rdr_set_build_flags(unsigned flags);

/* redner.DScene(...)  src/redner.cpp:75-82 */
typedef struct rdr_dscene_desc {
    rdr_dcamera_desc camera;
    const rdr_dshape_desc *shapes; int num_shapes;
    const rdr_dmaterial_desc *materials; int num_materials;
    const rdr_darea_light_desc *area_lights; int num_area_lights;
    const rdr_denvmap_desc *envmap;   /* or NULL */
} rdr_dscene_desc;

typedef struct rdr_scene rdr_scene;

/* redner.Scene(camera, shapes, materials, area_lights, envmap, use_gpu, gpu_index,
 *              use_primary_edge_sampling, use_secondary_edge_sampling)
 * src/redner.cpp:62-73, src/scene.cpp:63-307.  Copies the descriptors, keeps the data pointers
 * (the caller keeps the tensors alive), builds the triangle hierarchy, the light CDFs and the
 * edge-sampling structures.  Returns NULL on error. */
rdr_scene *rdr_scene_create(const rdr_camera_desc *camera,
                            const rdr_shape_desc *shapes, int num_shapes,
                            const rdr_material_desc *materials, int num_materials,
                            const rdr_area_light_desc *area_lights, int num_area_lights,
                            const rdr_envmap_desc *envmap,
                            int use_gpu, int gpu_index,
                            int use_primary_edge_sampling, int use_secondary_edge_sampling);
void rdr_scene_destroy(rdr_scene *scene);
/* Scene.max_generic_texture_dimension  src/redner.cpp:73 */
int rdr_scene_max_generic_texture_dimension(const rdr_scene *scene);

/* redner.render(scene, options, rendered_image, d_rendered_image, d_scene, screen_gradient_image,
 *               debug_image)   src/redner.cpp:257, src/pathtracer.cpp:177-958.
 * Forward iff rendered_image != NULL ([H_vp, W_vp, C] fp32 DEV, accumulated); backward iff
 * d_rendered_image != NULL (then d_scene must be given).  Synchronises the device before
 * returning, like the reference (src/pathtracer.cpp:947-949).  Returns 0 on success. */
int rdr_render(const rdr_scene *scene, const rdr_render_options *options,
               float *rendered_image, const float *d_rendered_image,
               const rdr_dscene_desc *d_scene,
               float *screen_gradient_image, float *debug_image);

/* redner.compute_num_channels(channels, max_generic_texture_dimension)  src/redner.cpp:201.
 * Returns -1 for an unknown channel id. */
int rdr_compute_num_channels(const int *channels, int num_channels, int max_generic_texture_dimension);

/* Message of the last failure on the calling thread ("" if none). */
const char *rdr_last_error(void);

/* Measurement hooks (no reference counterpart; used by bench.py and the roofline report). */
typedef struct rdr_trace_stats {
    double closest_ms, any_ms;           /* accumulated device time of the two traversal kernels */
    uint64_t closest_launches, any_launches;
    uint64_t closest_rays, any_rays;
    /* 32-byte node records loaded / 36-byte triangle records tested, per query kind; only
     * counted when counting is enabled (instrumented kernel variant) */
    uint64_t closest_nodes, closest_tris, any_nodes, any_tris;
    /* 128-byte records of the 4-wide form of the hierarchy loaded (the kernels walk one form or the other per launch) */
    uint64_t closest_wide_nodes, any_wide_nodes;
    /* time during which at least one launch of the kind was in flight (union of the launches' intervals): equals closest_ms /
     * any_ms when launches of a kind never overlap, less when two sample workers trace side by side */
    double closest_union_ms, any_union_ms;
} rdr_trace_stats;
void rdr_trace_stats_enable(int timing, int counting);
void rdr_trace_stats_reset(void);
void rdr_trace_stats_get(rdr_trace_stats *out);

/* Closest-hit / any-hit queries on a batch of rays (DEV pointers; 32-byte ray records
 * {org.xyz, tmin, dir.xyz, tmax}, 8-byte hit records {shape, prim}).  This is the boundary the
 * reference crosses into Embree/OptiX (src/scene.cpp:503-597, 629-690); exposed for the
 * traversal parity tests and micro-benchmarks. */
int rdr_scene_trace(const rdr_scene *scene, const float *rays, int32_t *hits, int num_rays, int any_hit);

/* Per-call device buffers (the reference's PathBuffer, src/pathtracer.cpp:36-152, allocated and freed by every render())
 * come from a caching allocator: blocks are parked for the next call of the same shape (bounded by RDR_POOL_CAP_MB per device,
 * default a quarter of the device's memory).  rdr_trim_cache() returns every parked block to the driver -- for processes that share the device with
 * another allocator (torch) and change resolution.  Returns the number of bytes released. */
uint64_t rdr_trim_cache(void);

/* Test hook: how often the library has gone to the runtime for device memory (hipMalloc calls made by its caching allocator)
 * and how often the host has read a live-lane count back from the device, since the library was loaded.  A steady-state
 * rdr_render() adds nothing to either (the reference allocates its PathBuffer per call, src/pathtracer.cpp:36-152, and
 * reads a count after every stage, :292,590,833). */
typedef struct rdr_debug_counters {
    uint64_t device_mallocs, host_count_reads;
    uint64_t last_batch_samples, last_workers;      /* how the last gradient render of this process was scheduled: samples per launch set, host threads */
} rdr_debug_counters;
void rdr_debug_counters_get(rdr_debug_counters *out);

/* Test hook: the triangle hierarchy the kernels of this Scene built (bvh_gpu.cpp) against the host builder's on the same
 * arrays: the number of records that differ (0: identical), -1 when this Scene's hierarchy is a refit or was not built by
 * kernels, -2 on error. */
int rdr_debug_bvh_check(const rdr_scene *scene);

/* Test hook: writes the edge list and both edge hierarchies (links, edge ids, weights, costs) as
 * text, for the build-order parity test against the reference (tests/test_edge_build.py). */
int rdr_debug_dump_edges(const rdr_scene *scene, const char *path);

/* The library is built twice from the same sources (__graft_entry__.build_native):
 *   libredner_amd.so        the stage kernels call the DEVICE's own sin / cos / atan2 / atan / acos / log / pow (ocml): the
 *                           default, +1 ... 3 % throughput; every result is an equally valid sample of the same estimator, and
 *                           sample-for-sample equal to the reference wherever no transcendental feeds a chaotic decision
 *                           (perspective / orthographic cameras: all BASELINE configs);
 *   libredner_amd_exact.so  they call restatements of glibc 2.35's routines (csrc/libm_exact.h, see NOTICE), bit for bit what
 *                           the reference's CPU path computes: fisheye / panorama cameras with secondary edge sampling are then
 *                           sample-exact too.  The parity tests load this one (REDNER_AMD_LIBM=exact, redner_amd/_capi.py).
 * rdr_libm_exact(): 1 in the second, 0 in the first. */
int rdr_libm_exact(void);

/* Test hook: sin / cos / atan2 / atan / acos / log / pow as the EXACT routines evaluate them (in either build) (csrc/libm_exact.h: glibc 2.35's
 * results bit for bit -- the reference's CPU path calls glibc, src/camera.h:142-191, src/material.h, src/envmap.h), one
 * argument per lane; HOST pointers, `y` may be NULL for the one-argument functions.
 * fn: 0 sin(x), 1 cos(x), 2 atan2(x, y), 3 atan(x), 4 acos(x), 5 log(x), 6 pow(x, y). */
int rdr_debug_libm(int fn, const double *x, const double *y, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* REDNER_AMD_H */
