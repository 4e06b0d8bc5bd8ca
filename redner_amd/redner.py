"""`redner` -- drop-in mirror of the reference's pybind11 module (src/redner.cpp:20-272) on top of
the MI355X-native C ABI (include/redner_amd.h).

Same class names, constructor argument order and attributes as the reference, so that the
unmodified `pyredner/render_pytorch.py` runs against it:

    import redner_amd; redner_amd.install()     # registers this module as `redner`
    import pyredner                              # the reference's Python package, unchanged

What `render_pytorch.py` touches (rendering) is mirrored in full; of the asset helpers only `load_serialized` is
provided (pure Python: zlib + numpy), so that `pyredner.load_mitsuba('scenes/bunny_box.xml')` -- the reference's own
tests/test_bunny_box.py -- runs on this module; `automatic_uv_map` / `rebuild_topology` (xatlas, host-side mesh
processing) raise NotImplementedError (out of scope).
Errors raise RuntimeError instead of aborting the process (the reference: assert / exit(1)).
"""
import ctypes as C
import enum
import sys
import struct
import zlib

from . import _capi


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, (float_ptr, int_ptr)):
        return p.addr
    return int(p)


class float_ptr:                       # src/redner.cpp:23-24, src/ptr.h:10-24
    def __init__(self, addr):
        self.addr = int(addr)


class int_ptr:                         # src/redner.cpp:25-26
    def __init__(self, addr):
        self.addr = int(addr)


class CameraType(enum.IntEnum):        # src/redner.cpp:28-32
    perspective = 0
    orthographic = 1
    fisheye = 2
    panorama = 3


class channels(enum.IntEnum):          # src/redner.cpp:183-199
    radiance = 0
    alpha = 1
    depth = 2
    position = 3
    geometry_normal = 4
    shading_normal = 5
    uv = 6
    barycentric_coordinates = 7
    diffuse_reflectance = 8
    specular_reflectance = 9
    roughness = 10
    generic_texture = 11
    vertex_color = 12
    shape_id = 13
    triangle_id = 14
    material_id = 15


class SamplerType(enum.IntEnum):       # src/redner.cpp:203-205
    independent = 0
    sobol = 1


class Vector2i:                        # src/redner.cpp:218-221
    def __init__(self, x, y):
        self.x, self.y = int(x), int(y)


class Vector2f:
    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)


class Vector3f:
    def __init__(self, x, y, z):
        self.x, self.y, self.z = float(x), float(y), float(z)


def _host_floats(addr, n):
    return list((C.c_float * n).from_address(addr)) if addr else None


class Camera:                          # src/redner.cpp:34-50
    def __init__(self, width, height, position, look, up, cam_to_world, world_to_cam,
                 intrinsic_mat_inv, intrinsic_mat, distortion_params, clip_near, camera_type,
                 viewport_beg, viewport_end):
        d = _capi.CameraDesc()
        d.width, d.height = int(width), int(height)
        # camera pointers are HOST memory read now (src/camera.h:44-65): snapshot them
        self._keep = []

        def snap(p, n):
            vals = _host_floats(_addr(p), n)
            if vals is None:
                return None
            arr = (C.c_float * n)(*vals)
            self._keep.append(arr)
            return C.cast(arr, C.c_void_p)

        d.position, d.look, d.up = snap(position, 3), snap(look, 3), snap(up, 3)
        d.cam_to_world, d.world_to_cam = snap(cam_to_world, 16), snap(world_to_cam, 16)
        d.intrinsic_mat_inv, d.intrinsic_mat = snap(intrinsic_mat_inv, 9), snap(intrinsic_mat, 9)
        d.distortion_params = snap(distortion_params, 8)
        d.clip_near = float(clip_near)
        d.camera_type = int(camera_type)
        d.viewport_beg[0], d.viewport_beg[1] = viewport_beg.x, viewport_beg.y
        d.viewport_end[0], d.viewport_end[1] = viewport_end.x, viewport_end.y
        self._desc = d
        self.use_look_at = not bool(d.cam_to_world)
        self._has_distortion = bool(d.distortion_params)

    def has_distortion_params(self):
        return self._has_distortion


class DCamera:                         # src/redner.cpp:52-60
    def __init__(self, position, look, up, cam_to_world, world_to_cam, intrinsic_mat_inv, intrinsic_mat,
                 distortion_params):
        d = _capi.DCameraDesc()
        d.position, d.look, d.up = _addr(position), _addr(look), _addr(up)
        d.cam_to_world, d.world_to_cam = _addr(cam_to_world), _addr(world_to_cam)
        d.intrinsic_mat_inv, d.intrinsic_mat = _addr(intrinsic_mat_inv), _addr(intrinsic_mat)
        d.distortion_params = _addr(distortion_params)
        self._desc = d


class Shape:                           # src/redner.cpp:84-104
    def __init__(self, vertices, indices, uvs, normals, uv_indices, normal_indices, colors,
                 num_vertices, num_uv_vertices, num_normal_vertices, num_triangles, material_id, light_id):
        d = _capi.ShapeDesc()
        d.vertices, d.indices = _addr(vertices), _addr(indices)
        d.uvs, d.normals = _addr(uvs), _addr(normals)
        d.uv_indices, d.normal_indices = _addr(uv_indices), _addr(normal_indices)
        d.colors = _addr(colors)
        d.num_vertices, d.num_uv_vertices = int(num_vertices), int(num_uv_vertices)
        d.num_normal_vertices, d.num_triangles = int(num_normal_vertices), int(num_triangles)
        d.material_id, d.light_id = int(material_id), int(light_id)
        self._desc = d
        self.num_vertices = d.num_vertices
        self.num_uv_vertices = d.num_uv_vertices
        self.num_normal_vertices = d.num_normal_vertices
        self.num_triangles = d.num_triangles
        self.material_id, self.light_id = d.material_id, d.light_id

    def has_uvs(self):
        return bool(self._desc.uvs)

    def has_normals(self):
        return bool(self._desc.normals)

    def has_colors(self):
        return bool(self._desc.colors)


class DShape:                          # src/redner.cpp:106-110
    def __init__(self, vertices, uvs, normals, colors):
        d = _capi.DShapeDesc()
        d.vertices, d.uvs, d.normals, d.colors = _addr(vertices), _addr(uvs), _addr(normals), _addr(colors)
        self._desc = d


class _Texture:                        # src/redner.cpp:112-131, src/texture.h:14-47
    _fixed_channels = None

    def __init__(self, texels, width, height, channels, uv_scale):
        assert len(texels) == len(width) == len(height)
        n = min(len(texels), _capi.MAX_MIP)
        self.texels = [_addr(t) for t in texels[:n]]
        self.width = [int(w) for w in width[:n]]
        self.height = [int(h) for h in height[:n]]
        self.channels = int(channels) if self._fixed_channels is None else self._fixed_channels
        self.num_levels = n
        self.uv_scale = _addr(uv_scale)

    def _to_desc(self):
        d = _capi.TextureDesc()
        for i in range(self.num_levels):
            d.texels[i], d.width[i], d.height[i] = self.texels[i], self.width[i], self.height[i]
        d.channels, d.num_levels, d.uv_scale = self.channels, self.num_levels, self.uv_scale
        return d

    def _to_ddesc(self):
        d = _capi.DTextureDesc()
        for i in range(self.num_levels):
            d.texels[i] = self.texels[i]
        d.num_levels, d.uv_scale = self.num_levels, self.uv_scale
        return d

    def _size(self, i):
        if i < self.num_levels:
            return self.width[i], self.height[i]
        return 0, 0


class Texture1(_Texture):
    _fixed_channels = 1


class Texture3(_Texture):
    _fixed_channels = 3


class TextureN(_Texture):
    _fixed_channels = None


class Material:                        # src/redner.cpp:133-151
    def __init__(self, diffuse_reflectance, specular_reflectance, roughness, generic_texture, normal_map,
                 compute_specular_lighting, two_sided, use_vertex_color):
        self.diffuse_reflectance, self.specular_reflectance = diffuse_reflectance, specular_reflectance
        self.roughness, self.generic_texture, self.normal_map = roughness, generic_texture, normal_map
        self.compute_specular_lighting = bool(compute_specular_lighting)
        self.two_sided, self.use_vertex_color = bool(two_sided), bool(use_vertex_color)

    def _to_desc(self):
        d = _capi.MaterialDesc()
        d.diffuse_reflectance = self.diffuse_reflectance._to_desc()
        d.specular_reflectance = self.specular_reflectance._to_desc()
        d.roughness = self.roughness._to_desc()
        d.generic_texture = self.generic_texture._to_desc()
        d.normal_map = self.normal_map._to_desc()
        d.compute_specular_lighting = int(self.compute_specular_lighting)
        d.two_sided, d.use_vertex_color = int(self.two_sided), int(self.use_vertex_color)
        return d

    def get_diffuse_levels(self):
        return self.diffuse_reflectance.num_levels

    def get_diffuse_size(self, i):
        return self.diffuse_reflectance._size(i)

    def get_specular_levels(self):
        return self.specular_reflectance.num_levels

    def get_specular_size(self, i):
        return self.specular_reflectance._size(i)

    def get_roughness_levels(self):
        return self.roughness.num_levels

    def get_roughness_size(self, i):
        return self.roughness._size(i)

    def get_generic_levels(self):
        return self.generic_texture.num_levels

    def get_generic_size(self, i):
        w, h = self.generic_texture._size(i)
        return self.generic_texture.channels, w, h

    def get_normal_map_levels(self):
        return self.normal_map.num_levels

    def get_normal_map_size(self, i):
        return self.normal_map._size(i)


class DMaterial:                       # src/redner.cpp:153-158
    def __init__(self, diffuse_reflectance, specular_reflectance, roughness, generic_texture, normal_map):
        self._parts = (diffuse_reflectance, specular_reflectance, roughness, generic_texture, normal_map)

    def _to_desc(self):
        d = _capi.DMaterialDesc()
        (d.diffuse_reflectance, d.specular_reflectance, d.roughness, d.generic_texture,
         d.normal_map) = [p._to_ddesc() for p in self._parts]
        return d


class AreaLight:                       # src/redner.cpp:160-164 (intensity: HOST pointer, read now)
    def __init__(self, shape_id, intensity, two_sided, directly_visible):
        d = _capi.AreaLightDesc()
        d.shape_id = int(shape_id)
        vals = _host_floats(_addr(intensity), 3)
        for k in range(3):
            d.intensity[k] = vals[k]
        d.two_sided, d.directly_visible = int(bool(two_sided)), int(bool(directly_visible))
        self._desc = d


class DAreaLight:                      # src/redner.cpp:166-167
    def __init__(self, intensity):
        d = _capi.DAreaLightDesc()
        d.intensity = _addr(intensity)
        self._desc = d


class EnvironmentMap:                  # src/redner.cpp:169-177
    def __init__(self, values, env_to_world, world_to_env, sample_cdf_ys, sample_cdf_xs, pdf_norm,
                 directly_visible):
        self.values = values
        self._keep = []
        d = _capi.EnvmapDesc()
        d.values = values._to_desc()
        for name, p in (('env_to_world', env_to_world), ('world_to_env', world_to_env)):
            arr = (C.c_float * 16)(*_host_floats(_addr(p), 16))
            self._keep.append(arr)
            setattr(d, name, C.cast(arr, C.c_void_p))
        d.sample_cdf_ys, d.sample_cdf_xs = _addr(sample_cdf_ys), _addr(sample_cdf_xs)
        d.pdf_norm, d.directly_visible = float(pdf_norm), int(bool(directly_visible))
        self._desc = d

    def get_levels(self):
        return self.values.num_levels

    def get_size(self, i):
        return self.values._size(i)


class DEnvironmentMap:                 # src/redner.cpp:179-181
    def __init__(self, values, world_to_env):
        d = _capi.DEnvmapDesc()
        d.values = values._to_ddesc()
        d.world_to_env = _addr(world_to_env)
        self._desc = d


class Scene:                           # src/redner.cpp:62-73
    def __init__(self, camera, shapes, materials, area_lights, envmap, use_gpu, gpu_index,
                 use_primary_edge_sampling, use_secondary_edge_sampling):
        lib = _capi.lib()
        self._lib = lib
        _use_torch_stream(lib, use_gpu, gpu_index)
        self.camera = camera
        sh = (_capi.ShapeDesc * max(len(shapes), 1))(*[s._desc for s in shapes])
        mt = (_capi.MaterialDesc * max(len(materials), 1))(*[m._to_desc() for m in materials])
        al = (_capi.AreaLightDesc * max(len(area_lights), 1))(*[l._desc for l in area_lights])
        env = C.byref(envmap._desc) if envmap is not None else None
        self._refs = (camera, shapes, materials, area_lights, envmap)
        self._handle = lib.rdr_scene_create(C.byref(camera._desc), sh, len(shapes), mt, len(materials),
                                            al, len(area_lights), env, int(bool(use_gpu)), int(gpu_index),
                                            int(bool(use_primary_edge_sampling)),
                                            int(bool(use_secondary_edge_sampling)))
        if not self._handle:
            raise RuntimeError('redner.Scene: ' + _capi.last_error())
        self.max_generic_texture_dimension = lib.rdr_scene_max_generic_texture_dimension(self._handle)
        self.use_gpu, self.gpu_index = bool(use_gpu), int(gpu_index)

    def __del__(self):
        h, self._handle = getattr(self, '_handle', None), None
        if h:
            self._lib.rdr_scene_destroy(h)


class DScene:                          # src/redner.cpp:75-82
    def __init__(self, camera, shapes, materials, area_lights, envmap, use_gpu, gpu_index):
        self._refs = (camera, shapes, materials, area_lights, envmap)
        d = _capi.DSceneDesc()
        d.camera = camera._desc
        self._sh = (_capi.DShapeDesc * max(len(shapes), 1))(*[s._desc for s in shapes])
        self._mt = (_capi.DMaterialDesc * max(len(materials), 1))(*[m._to_desc() for m in materials])
        self._al = (_capi.DAreaLightDesc * max(len(area_lights), 1))(*[l._desc for l in area_lights])
        d.shapes, d.num_shapes = self._sh, len(shapes)
        d.materials, d.num_materials = self._mt, len(materials)
        d.area_lights, d.num_area_lights = self._al, len(area_lights)
        d.envmap = C.pointer(envmap._desc) if envmap is not None else None
        self._desc = d


class RenderOptions:                   # src/redner.cpp:207-216 (+ sample_offset/total_samples extension)
    def __init__(self, seed, num_samples, max_bounces, channels, sampler_type, sample_pixel_center):
        self.seed = int(seed)
        self.num_samples = int(num_samples)
        self.max_bounces = int(max_bounces)
        self.channels = [int(c) for c in channels]
        self.sampler_type = int(sampler_type)
        self.sample_pixel_center = bool(sample_pixel_center)
        self.sample_offset = 0
        self.total_samples = 0
        # rdr_tuning (redner_amd extension; every field 0 = the library's default): e.g. options.tuning.batch_samples = 1
        self.tuning = _capi.Tuning()

    def _to_desc(self):
        d = _capi.RenderOptionsDesc()
        self._ch = (C.c_int * max(len(self.channels), 1))(*self.channels)
        d.seed, d.num_samples, d.max_bounces = self.seed & 0xFFFFFFFFFFFFFFFF, self.num_samples, self.max_bounces
        d.channels, d.num_channels = self._ch, len(self.channels)
        d.sampler_type, d.sample_pixel_center = self.sampler_type, int(self.sample_pixel_center)
        d.sample_offset, d.total_samples = int(self.sample_offset), int(self.total_samples)
        d.tuning = C.pointer(self.tuning)
        return d


def compute_num_channels(channel_list, max_generic_texture_dimension):    # src/redner.cpp:201
    ch = (C.c_int * max(len(channel_list), 1))(*[int(c) for c in channel_list])
    n = _capi.lib().rdr_compute_num_channels(ch, len(channel_list), int(max_generic_texture_dimension))
    if n < 0:
        raise RuntimeError('redner.compute_num_channels: unknown channel')
    return n


def render(scene, options, rendered_image, d_rendered_image, d_scene, screen_gradient_image, debug_image):
    """redner.render(...)  src/redner.cpp:257 -- forward iff rendered_image != 0, backward iff
    d_rendered_image != 0."""
    od = options._to_desc()
    _use_torch_stream(scene._lib, scene.use_gpu, scene.gpu_index)
    ds = C.byref(d_scene._desc) if d_scene is not None else None
    rc = scene._lib.rdr_render(scene._handle, C.byref(od), _addr(rendered_image), _addr(d_rendered_image), ds,
                               _addr(screen_gradient_image), _addr(debug_image))
    if rc != 0:
        raise RuntimeError('redner.render: ' + _capi.last_error())


def _use_torch_stream(lib, use_gpu, gpu_index=None):
    """The library orders its launches on the calling thread's CURRENT torch stream OF THE SCENE'S DEVICE (rdr_set_stream):
    tensors produced under `with torch.cuda.stream(s):` are read after their producers without a device-wide
    synchronisation.  (The reference's kernels run on the null stream, which torch's default stream is.)"""
    if not use_gpu:
        return
    torch = sys.modules.get('torch')
    stream = 0
    if torch is not None and torch.cuda.is_available():
        # a Scene may live on another device than torch's current one (pyredner.set_device without torch.cuda.set_device):
        # a stream handle is only valid on the device it was created on
        stream = int(torch.cuda.current_stream(gpu_index).cuda_stream)
    lib.rdr_set_stream(stream or None)


def set_pool_cap_mb(megabytes):
    """Not in the reference: bound of the library's buffer cache per device (default min(a quarter of the device, 8 GiB)).
    A dedicated render process may raise it (e.g. 65536) so that the buffers of a 2^24-lane sample batch stay parked
    between calls; negative = back to the default."""
    _capi.lib().rdr_set_pool_cap_mb(int(megabytes))


def get_pool_cap_mb():
    """Not in the reference: the bound of the buffer cache in effect (MiB)."""
    return int(_capi.lib().rdr_get_pool_cap_mb())


def set_build_flags(flags):
    """Not in the reference: rdr_build_flags (_capi.BUILD_*) for the Scenes created from now on (debugging / tests)."""
    _capi.lib().rdr_set_build_flags(int(flags))


def trim_cache():
    """Not in the reference: return the device buffers parked by the caching allocator to the driver (rdr_trim_cache) --
    for processes that share the GPU with torch's allocator and change resolution.  Returns the bytes released."""
    return int(_capi.lib().rdr_trim_cache())


# ---- Mitsuba .serialized meshes (src/redner.cpp:232-238, src/load_serialized.cpp:124-288) -----------------------------
class MitsubaTriMesh:
    """vertices [V,3] f32, indices [T,3] i32, uvs [V,2] f32 or [0], normals [V,3] f32 or [0] (numpy)."""

    def __init__(self, vertices, indices, uvs, normals):
        self.vertices, self.indices, self.uvs, self.normals = vertices, indices, uvs, normals


def load_serialized(filename, idx):
    """Sub-mesh `idx` of a Mitsuba 0.5 serialized file: u16 magic, u16 version (3 / 4), a zlib stream per mesh, and
    at the end of the file the offset table (u64 entries for v4, u32 for v3) followed by the mesh count (u32)."""
    import numpy as np
    E_NORMALS, E_TEXCOORDS, E_COLORS, E_DOUBLE = 0x0001, 0x0002, 0x0008, 0x2000
    with open(filename, 'rb') as f:
        data = f.read()
    if len(data) < 8:
        raise RuntimeError('load_serialized: %s is not a serialized mesh file' % filename)
    version = struct.unpack_from('<H', data, 2)[0]
    start = 4
    if idx > 0:
        count = struct.unpack_from('<I', data, len(data) - 4)[0]
        if idx >= count:
            raise RuntimeError('load_serialized: shape index %d out of range (%d meshes)' % (idx, count))
        if version == 4:
            off = struct.unpack_from('<Q', data, len(data) - 4 - 8 * (count - idx))[0]
        else:
            off = struct.unpack_from('<I', data, len(data) - 4 * (count - idx + 1))[0]
        start = off + 4                                   # skip that mesh's own magic + version
    try:
        raw = zlib.decompressobj().decompress(data[start:])
    except zlib.error as e:
        raise RuntimeError('load_serialized: inflate(): %s' % e)
    pos = 0
    flags = struct.unpack_from('<I', raw, pos)[0]
    pos += 4
    if version == 4:
        pos = raw.index(b'\0', pos) + 1                   # null-terminated mesh name
    n_vert, n_tri = struct.unpack_from('<QQ', raw, pos)
    pos += 16
    real = np.dtype('<f8') if flags & E_DOUBLE else np.dtype('<f4')

    def block(count, width, dtype):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=count * width, offset=pos).reshape(count, width)
        pos += a.nbytes
        return a

    vertices = block(n_vert, 3, real).astype(np.float32)
    normals = block(n_vert, 3, real).astype(np.float32) if flags & E_NORMALS else np.zeros((0,), np.float32)
    uvs = block(n_vert, 2, real).astype(np.float32) if flags & E_TEXCOORDS else np.zeros((0,), np.float32)
    if flags & E_COLORS:
        block(n_vert, 3, real)                             # read and dropped, like the reference
    indices = block(n_tri, 3, np.dtype('<i4')).astype(np.int32)
    return MitsubaTriMesh(np.ascontiguousarray(vertices), np.ascontiguousarray(indices), uvs, normals)


def _host_mesh_tool(name):
    def stub(*_a, **_k):
        raise NotImplementedError('redner.%s is host-side mesh processing outside the renderer hot path '
                                  '(SURVEY.md section 2.1); use the reference build for it' % name)
    stub.__name__ = name
    return stub


automatic_uv_map = _host_mesh_tool('automatic_uv_map')
copy_texture_atlas = _host_mesh_tool('copy_texture_atlas')
rebuild_topology = _host_mesh_tool('rebuild_topology')
