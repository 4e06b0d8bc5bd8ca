"""ctypes binding of the C ABI in include/redner_amd.h.

The shared library is the product: `redner_amd/lib/libredner_amd.so`, built by
`__graft_entry__.build()` (hipcc, gfx950).  If it is missing the import fails loudly -- there is
no CPU fallback.  `load(path)` exists so the test-suite can point the binding at the
single-threaded debugging harness under tests/hostsim/ (test infrastructure); product code never
calls it with an argument.
"""
import ctypes as C
import os

MAX_MIP = 8
_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, 'lib', 'libredner_amd.so')
# the same library with glibc-exact transcendental functions in its kernels (include/redner_amd.h: rdr_libm_exact):
# REDNER_AMD_LIBM=exact selects it (the parity tests do, tests/conftest.py)
EXACT_LIBRARY = os.path.join(_HERE, 'lib', 'libredner_amd_exact.so')

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class CameraDesc(C.Structure):
    _fields_ = [('width', C.c_int), ('height', C.c_int),
                ('position', C.c_void_p), ('look', C.c_void_p), ('up', C.c_void_p),
                ('cam_to_world', C.c_void_p), ('world_to_cam', C.c_void_p),
                ('intrinsic_mat_inv', C.c_void_p), ('intrinsic_mat', C.c_void_p),
                ('distortion_params', C.c_void_p),
                ('clip_near', C.c_float), ('camera_type', C.c_int),
                ('viewport_beg', C.c_int * 2), ('viewport_end', C.c_int * 2)]


class DCameraDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('position', 'look', 'up', 'cam_to_world', 'world_to_cam',
                                          'intrinsic_mat_inv', 'intrinsic_mat', 'distortion_params')]


class ShapeDesc(C.Structure):
    _fields_ = [('vertices', C.c_void_p), ('indices', C.c_void_p), ('uvs', C.c_void_p),
                ('normals', C.c_void_p), ('uv_indices', C.c_void_p), ('normal_indices', C.c_void_p),
                ('colors', C.c_void_p),
                ('num_vertices', C.c_int), ('num_uv_vertices', C.c_int), ('num_normal_vertices', C.c_int),
                ('num_triangles', C.c_int), ('material_id', C.c_int), ('light_id', C.c_int)]


class DShapeDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('vertices', 'uvs', 'normals', 'colors')]


class TextureDesc(C.Structure):
    _fields_ = [('texels', C.c_void_p * MAX_MIP), ('width', C.c_int * MAX_MIP), ('height', C.c_int * MAX_MIP),
                ('channels', C.c_int), ('num_levels', C.c_int), ('uv_scale', C.c_void_p)]


class MaterialDesc(C.Structure):
    _fields_ = [('diffuse_reflectance', TextureDesc), ('specular_reflectance', TextureDesc),
                ('roughness', TextureDesc), ('generic_texture', TextureDesc), ('normal_map', TextureDesc),
                ('compute_specular_lighting', C.c_int), ('two_sided', C.c_int), ('use_vertex_color', C.c_int)]


class DTextureDesc(C.Structure):
    _fields_ = [('texels', C.c_void_p * MAX_MIP), ('num_levels', C.c_int), ('uv_scale', C.c_void_p)]


class DMaterialDesc(C.Structure):
    _fields_ = [(n, DTextureDesc) for n in ('diffuse_reflectance', 'specular_reflectance', 'roughness',
                                            'generic_texture', 'normal_map')]


class AreaLightDesc(C.Structure):
    _fields_ = [('shape_id', C.c_int), ('intensity', C.c_float * 3), ('two_sided', C.c_int),
                ('directly_visible', C.c_int)]


class DAreaLightDesc(C.Structure):
    _fields_ = [('intensity', C.c_void_p)]


class EnvmapDesc(C.Structure):
    _fields_ = [('values', TextureDesc), ('env_to_world', C.c_void_p), ('world_to_env', C.c_void_p),
                ('sample_cdf_ys', C.c_void_p), ('sample_cdf_xs', C.c_void_p),
                ('pdf_norm', C.c_float), ('directly_visible', C.c_int)]


class DEnvmapDesc(C.Structure):
    _fields_ = [('values', DTextureDesc), ('world_to_env', C.c_void_p)]


class Tuning(C.Structure):
    """rdr_tuning: every field 0 = the library's default (include/redner_amd.h)."""
    _fields_ = [('flags', C.c_uint), ('batch_samples', C.c_int), ('batch_lanes', C.c_int64), ('workers', C.c_int),
                ('refill_rays_per_lane', C.c_int), ('refill_idle_lanes', C.c_int), ('refill_steps', C.c_int),
                ('wide_max_rays', C.c_int), ('gather_budget', C.c_int),
                ('gather_heavy_cap_plus1', C.c_int), ('gather_work_cap_plus1', C.c_int), ('mem_available_mb', C.c_int),
                ('refill_order', C.c_int), ('pickh_slots_per_lane', C.c_int), ('pickh_idle_lanes', C.c_int), ('pickh_steps', C.c_int)]


# rdr_tune_flags / rdr_build_flags
TUNE_NO_OVERLAP, TUNE_FORCE_GENERAL, TUNE_PICKN_WALK, TUNE_PICKH_FUSED, TUNE_PICKH_LAZY, TUNE_NO_HOIST, TUNE_REFILL_OFF, \
    TUNE_REFILL_ALL, TUNE_TRACE_BINARY, TUNE_TRACE_NO_LDS_TOP, TUNE_NO_FUSED_BOUNCE, TUNE_PICKH_ONE_LAUNCH, TUNE_NO_NEE_COMPACT, \
    TUNE_LARGE_FORMS, TUNE_TRACE_EVERY_CONTINUATION = [1 << k for k in range(15)]
BUILD_NO_REFIT, BUILD_NO_EDGE_CACHE, BUILD_SYNC_EDGES, BUILD_EDGE_HOST_BUILD = [1 << k for k in range(4)]


class RenderOptionsDesc(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('num_samples', C.c_int), ('max_bounces', C.c_int),
                ('channels', C.POINTER(C.c_int)), ('num_channels', C.c_int),
                ('sampler_type', C.c_int), ('sample_pixel_center', C.c_int),
                ('sample_offset', C.c_int), ('total_samples', C.c_int),
                ('tuning', C.POINTER(Tuning))]


class DSceneDesc(C.Structure):
    _fields_ = [('camera', DCameraDesc),
                ('shapes', C.POINTER(DShapeDesc)), ('num_shapes', C.c_int),
                ('materials', C.POINTER(DMaterialDesc)), ('num_materials', C.c_int),
                ('area_lights', C.POINTER(DAreaLightDesc)), ('num_area_lights', C.c_int),
                ('envmap', C.POINTER(DEnvmapDesc))]


class TraceStats(C.Structure):
    _fields_ = [('closest_ms', C.c_double), ('any_ms', C.c_double),
                ('closest_launches', C.c_uint64), ('any_launches', C.c_uint64),
                ('closest_rays', C.c_uint64), ('any_rays', C.c_uint64),
                ('closest_nodes', C.c_uint64), ('closest_tris', C.c_uint64),
                ('any_nodes', C.c_uint64), ('any_tris', C.c_uint64),
                ('closest_wide_nodes', C.c_uint64), ('any_wide_nodes', C.c_uint64),
                ('closest_union_ms', C.c_double), ('any_union_ms', C.c_double)]


class DebugCounters(C.Structure):
    _fields_ = [('device_mallocs', C.c_uint64), ('host_count_reads', C.c_uint64),
                ('last_batch_samples', C.c_uint64), ('last_workers', C.c_uint64)]


EXPORTS = ('rdr_scene_create', 'rdr_scene_destroy', 'rdr_scene_max_generic_texture_dimension',
           'rdr_render', 'rdr_compute_num_channels', 'rdr_last_error',
           'rdr_trace_stats_enable', 'rdr_trace_stats_reset', 'rdr_trace_stats_get', 'rdr_scene_trace',
           'rdr_debug_counters_get', 'rdr_trim_cache', 'rdr_debug_dump_edges', 'rdr_debug_bvh_check',
           'rdr_set_stream', 'rdr_set_pool_cap_mb', 'rdr_get_pool_cap_mb', 'rdr_set_build_flags', 'rdr_debug_libm', 'rdr_libm_exact')

_lib = None
_lib_path = None


def load(path=None):
    """Load the C-ABI library (default: the HIP build).  Raises RuntimeError when it is missing."""
    global _lib, _lib_path
    # REDNER_AMD_LIB: another build of the same library (A/B of two builds inside one GPU session, tools/gpu_ab_builds.sh)
    path = path or os.environ.get('REDNER_AMD_LIB') or \
        (EXACT_LIBRARY if os.environ.get('REDNER_AMD_LIBM', '').lower() == 'exact' else DEFAULT_LIBRARY)
    if not os.path.exists(path):
        raise RuntimeError(
            "redner_amd: native library %s not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError("redner_amd: %s does not export %s" % (path, name))
    lib.rdr_scene_create.restype = C.c_void_p
    lib.rdr_scene_create.argtypes = [C.POINTER(CameraDesc), C.POINTER(ShapeDesc), C.c_int,
                                     C.POINTER(MaterialDesc), C.c_int, C.POINTER(AreaLightDesc), C.c_int,
                                     C.POINTER(EnvmapDesc), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rdr_scene_destroy.restype = None
    lib.rdr_scene_destroy.argtypes = [C.c_void_p]
    lib.rdr_scene_max_generic_texture_dimension.restype = C.c_int
    lib.rdr_scene_max_generic_texture_dimension.argtypes = [C.c_void_p]
    lib.rdr_render.restype = C.c_int
    lib.rdr_render.argtypes = [C.c_void_p, C.POINTER(RenderOptionsDesc), C.c_void_p, C.c_void_p,
                               C.POINTER(DSceneDesc), C.c_void_p, C.c_void_p]
    lib.rdr_compute_num_channels.restype = C.c_int
    lib.rdr_compute_num_channels.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.rdr_last_error.restype = C.c_char_p
    lib.rdr_last_error.argtypes = []
    lib.rdr_trace_stats_enable.restype = None
    lib.rdr_trace_stats_enable.argtypes = [C.c_int, C.c_int]
    lib.rdr_trace_stats_reset.restype = None
    lib.rdr_debug_counters_get.restype = None
    lib.rdr_debug_counters_get.argtypes = [C.POINTER(DebugCounters)]
    lib.rdr_trim_cache.restype = C.c_uint64
    lib.rdr_trim_cache.argtypes = []
    lib.rdr_debug_bvh_check.restype = C.c_int
    lib.rdr_debug_bvh_check.argtypes = [C.c_void_p]
    lib.rdr_set_stream.restype = None
    lib.rdr_set_stream.argtypes = [C.c_void_p]
    lib.rdr_set_pool_cap_mb.restype = None
    lib.rdr_set_pool_cap_mb.argtypes = [C.c_int64]
    lib.rdr_get_pool_cap_mb.restype = C.c_int64
    lib.rdr_get_pool_cap_mb.argtypes = []
    lib.rdr_set_build_flags.restype = None
    lib.rdr_set_build_flags.argtypes = [C.c_uint]
    lib.rdr_trace_stats_get.restype = None
    lib.rdr_trace_stats_get.argtypes = [C.POINTER(TraceStats)]
    lib.rdr_scene_trace.restype = C.c_int
    lib.rdr_scene_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.rdr_libm_exact.restype = C.c_int
    lib.rdr_libm_exact.argtypes = []
    lib.rdr_debug_libm.restype = C.c_int
    lib.rdr_debug_libm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    _lib, _lib_path = lib, path
    return lib


def lib():
    if _lib is None:
        load()
    return _lib


def library_path():
    return _lib_path


def is_product_library():
    """The loaded library is one of the two HIP builds (not the CPU debugging harness, not an A/B variant)."""
    return _lib_path in (DEFAULT_LIBRARY, EXACT_LIBRARY)


def last_error():
    return lib().rdr_last_error().decode('utf-8', 'replace')
