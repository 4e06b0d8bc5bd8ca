"""PyTorch autograd surface of the MI355X-native renderer.

Host-side mirror of the reference's `pyredner/render_pytorch.py` (RenderFunction.serialize_scene /
apply(forward, backward), lines 62-1177) for the hot path, written against `redner_amd.redner`.
The same names and argument meanings are kept so tests read like the reference's:

    args = RenderFunction.serialize_scene(scene, num_samples=4, max_bounces=1,
                                          sampler_type=redner.SamplerType.sobol)
    img = RenderFunction.apply(seed, *args)
    img.sum().backward()

The scene classes below (Camera, Shape, Material, AreaLight, Scene) carry only what crosses the
boundary; loaders, mip-map generation, image IO etc. remain the reference's pure-Python package
(out of scope).  On a machine that has the reference checkout, the
unmodified pyredner package can be used instead: `redner_amd.install()`.

`backend` (default: redner_amd.redner) is the module providing the `redner` API; the parity
tests pass the oracle build of the reference here to render the same scene with both.
"""
import math
import os
import weakref
from typing import List, Optional

import torch

from . import redner as _default_backend


class Camera:
    """Pinhole camera (pyredner/camera.py): look-at or cam_to_world parameterisation."""

    def __init__(self, position=None, look_at=None, up=None, fov=None, clip_near=1e-4,
                 resolution=(256, 256), viewport=None, cam_to_world=None, intrinsic_mat=None,
                 camera_type=0, distortion_params=None):
        self.position, self.look_at, self.up = position, look_at, up
        self.cam_to_world = cam_to_world
        self.world_to_cam = torch.inverse(cam_to_world).contiguous() if cam_to_world is not None else None
        if intrinsic_mat is None:
            # pyredner/camera.py: fov (degrees) -> diag(1/tan(fov/2), 1/tan(fov/2), 1)
            fov = fov if isinstance(fov, torch.Tensor) else torch.tensor([float(fov)])
            fov_factor = 1.0 / torch.tan((math.pi / 180.0) * (0.5 * fov))     # fp32, like pyredner
            intrinsic_mat = torch.diag(torch.cat([fov_factor, fov_factor, torch.ones(1)], 0)).contiguous()
        self.intrinsic_mat = intrinsic_mat
        self.intrinsic_mat_inv = torch.inverse(intrinsic_mat).contiguous()
        self.clip_near = clip_near
        self.resolution = tuple(resolution)      # (height, width)
        self.viewport = viewport                  # (y0, x0, y1, x1) or None
        self.camera_type = camera_type            # redner.CameraType: perspective / orthographic / fisheye / panorama
        self.distortion_params = distortion_params   # 8 floats (k0..k5, p0, p1) or None


class Shape:
    def __init__(self, vertices, indices, material_id, uvs=None, normals=None, uv_indices=None,
                 normal_indices=None, colors=None):
        self.vertices, self.indices = vertices, indices
        self.uvs, self.normals = uvs, normals
        self.uv_indices, self.normal_indices = uv_indices, normal_indices
        self.colors = colors
        self.material_id = material_id
        self.light_id = -1


class Texture:
    """A constant colour (1-D tensor) or a list of mip levels [H, W, C] (pyredner/texture.py)."""

    def __init__(self, texels, uv_scale=None):
        if isinstance(texels, torch.Tensor) and texels.dim() == 1:
            self.mipmap, self.constant = [texels], True
        elif isinstance(texels, torch.Tensor):
            self.mipmap, self.constant = [texels], False
        else:
            self.mipmap, self.constant = list(texels), False
        self.uv_scale = uv_scale if uv_scale is not None else torch.tensor([1.0, 1.0])


class Material:
    def __init__(self, diffuse_reflectance=None, specular_reflectance=None, roughness=None,
                 generic_texture=None, normal_map=None, two_sided=False, use_vertex_color=False):
        def tex(t, default):
            if t is None:
                t = torch.tensor(default)
            return t if isinstance(t, Texture) else Texture(t)
        self.diffuse_reflectance = tex(diffuse_reflectance, [0.0, 0.0, 0.0])
        # pyredner/material.py: specular lighting is computed only when a specular term was given
        self.compute_specular_lighting = specular_reflectance is not None
        self.specular_reflectance = tex(specular_reflectance, [0.0, 0.0, 0.0])
        self.roughness = tex(roughness, [1.0])
        # optional (pyredner/material.py): N-channel texture for the generic_texture channel, RGB normal map
        self.generic_texture = tex(generic_texture, None) if generic_texture is not None else None
        self.normal_map = tex(normal_map, None) if normal_map is not None else None
        self.two_sided = two_sided
        self.use_vertex_color = use_vertex_color


class AreaLight:
    def __init__(self, shape_id, intensity, two_sided=False, directly_visible=True):
        self.shape_id, self.intensity = shape_id, intensity
        self.two_sided, self.directly_visible = two_sided, directly_visible


class EnvironmentMap:
    """Latitude-longitude environment light (pyredner/envmap.py): `values` is a Texture (or an [H, W, 3] tensor),
    the sampling tables are rebuilt from level 0 exactly as pyredner.EnvironmentMap.generate_envmap_pdf does."""

    def __init__(self, values, env_to_world=None, directly_visible=True):
        self.values = values if isinstance(values, Texture) else Texture(values)
        self.env_to_world = env_to_world if env_to_world is not None else torch.eye(4, 4)
        self.world_to_env = torch.inverse(self.env_to_world).contiguous()
        self.directly_visible = directly_visible
        t = self.values.mipmap[0].detach()
        lum = 0.212671 * t[:, :, 0] + 0.715160 * t[:, :, 1] + 0.072169 * t[:, :, 2]
        cdf_xs_ = torch.cumsum(lum, dim=1)
        y_weight = torch.sin(math.pi * (torch.arange(lum.shape[0], dtype=torch.float32, device=lum.device) + 0.5)
                             / float(lum.shape[0]))
        cdf_ys_ = torch.cumsum(cdf_xs_[:, -1] * y_weight, dim=0)
        self.pdf_norm = (lum.shape[0] * lum.shape[1]) / (cdf_ys_[-1].item() * (2 * math.pi * math.pi))
        cdf_xs = (cdf_xs_ - cdf_xs_[:, 0:1]) / torch.max(cdf_xs_[:, (lum.shape[1] - 1):lum.shape[1]],
                                                        1e-8 * torch.ones(cdf_xs_.shape[0], 1, device=lum.device))
        cdf_ys = (cdf_ys_ - cdf_ys_[0]) / torch.max(cdf_ys_[-1], torch.tensor([1e-8], device=lum.device))
        self.sample_cdf_ys, self.sample_cdf_xs = cdf_ys.contiguous(), cdf_xs.contiguous()


class Scene:
    def __init__(self, camera, shapes, materials, area_lights, envmap=None):
        self.camera, self.shapes, self.materials, self.area_lights = camera, shapes, materials, area_lights
        self.envmap = envmap


def _data_ptr(t):
    return t.data_ptr() if t is not None else 0


class _Unpacked:
    pass


# serialize_scene() asserts that every floating-point scene tensor is finite, like pyredner (render_pytorch.py:194-270: one
# torch.isfinite(...).all() -- two kernels and a synchronisation -- per tensor and call).  Here the device tensors of a call
# are checked by ONE multi-tensor kernel and one read-back, every call.  The only tensors that are not looked at again are
# LARGE device tensors that do not require a gradient (a static mesh, a fixed image texture: not what a loop writes to) and
# were finite at their current `_version`.  `_version` does not see writes through `.data`, through a numpy alias
# (torch.from_numpy) or by a foreign kernel through data_ptr() -- `p.data.clamp_()` on a PARAMETER is common in pyredner
# scripts -- which is why parameters (requires_grad) and small tensors are never skipped; REDNER_AMD_FINITE_CACHE=0 makes
# every call check everything.
_finite_seen = {}            # id(tensor) -> (weak reference to it, version at which it was found finite)
_FINITE_CACHE_MIN_ELEMS = 1 << 16
_finite_cache_on = os.environ.get('REDNER_AMD_FINITE_CACHE', '1') != '0'


def _known_finite(t):
    if not _finite_cache_on or t.requires_grad or t.numel() < _FINITE_CACHE_MIN_ELEMS or t.device.type == 'cpu':
        return False
    e = _finite_seen.get(id(t))
    return e is not None and e[0]() is t and e[1] == t._version


def _remember_finite(t):
    key = id(t)
    try:
        _finite_seen[key] = (weakref.ref(t, lambda _r, k=key: _finite_seen.pop(k, None)), t._version)
    except TypeError:
        pass


class RenderFunction(torch.autograd.Function):
    """torch.autograd.Function around redner.render (pyredner/render_pytorch.py:62-1177)."""

    @staticmethod
    def serialize_scene(scene: Scene, num_samples, max_bounces, channels: Optional[List] = None,
                        sampler_type=None, use_primary_edge_sampling=True, use_secondary_edge_sampling=True,
                        sample_pixel_center=False, device: Optional[torch.device] = None, backend=None, tuning=None):
        """Flatten a scene into [meta, tensor, tensor, ...] for RenderFunction.apply.
        `tuning` (redner_amd extension): dict of rdr_tuning fields (include/redner_amd.h), e.g. {'batch_samples': 1}."""
        backend = backend or _default_backend
        if device is None:
            device = torch.device('cuda:%d' % torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        if channels is None:
            channels = [backend.channels.radiance]
        if sampler_type is None:
            sampler_type = backend.SamplerType.independent
        if isinstance(num_samples, int):
            num_samples = (num_samples, num_samples)
        if max_bounces == 0:
            use_secondary_edge_sampling = False
        for light_id, light in enumerate(scene.area_lights):
            scene.shapes[light.shape_id].light_id = light_id

        tensors = []
        to_check = []           # device tensors are checked together at the end: one multi-tensor kernel, one synchronisation

        def put(t, dev):
            if t is None:
                return -1
            if t.is_floating_point() and not _known_finite(t):
                if t.device.type == 'cpu':
                    assert torch.isfinite(t).all()
                else:
                    to_check.append(t)
            tensors.append(t.to(dev).contiguous())
            return len(tensors) - 1

        cam = scene.camera
        needs_visibility = False
        cpu = torch.device('cpu')
        meta = {'backend': backend, 'device': device, 'num_samples': num_samples, 'max_bounces': max_bounces,
                'channels': list(channels), 'sampler_type': sampler_type,
                'sample_pixel_center': sample_pixel_center}
        if tuning:
            meta['tuning'] = dict(tuning)
        cm = {}
        for name in ('position', 'look_at', 'up', 'cam_to_world', 'world_to_cam', 'intrinsic_mat_inv', 'intrinsic_mat',
                     'distortion_params'):
            t = getattr(cam, name)
            if t is not None and t.requires_grad:
                needs_visibility = True
            cm[name] = put(t, cpu)                       # camera tensors are HOST tensors (render_pytorch.py:168-192)
        cm['clip_near'], cm['resolution'], cm['camera_type'] = cam.clip_near, cam.resolution, cam.camera_type
        vp = cam.viewport if cam.viewport is not None else (0, 0, cam.resolution[0], cam.resolution[1])
        cm['viewport'] = (max(vp[0], 0), max(vp[1], 0), min(vp[2], cam.resolution[0]), min(vp[3], cam.resolution[1]))
        meta['camera'] = cm
        meta['shapes'] = []
        for sh in scene.shapes:
            if sh.vertices.requires_grad:
                needs_visibility = True
            meta['shapes'].append({
                'vertices': put(sh.vertices, device), 'indices': put(sh.indices.to(torch.int32), device),
                'uvs': put(sh.uvs, device), 'normals': put(sh.normals, device),
                'uv_indices': put(sh.uv_indices.to(torch.int32) if sh.uv_indices is not None else None, device),
                'normal_indices': put(sh.normal_indices.to(torch.int32) if sh.normal_indices is not None else None, device),
                'colors': put(sh.colors, device),
                'material_id': sh.material_id, 'light_id': sh.light_id})
        meta['materials'] = []
        for m in scene.materials:
            mm = {'compute_specular_lighting': m.compute_specular_lighting, 'two_sided': m.two_sided,
                  'use_vertex_color': m.use_vertex_color}
            for name in ('diffuse_reflectance', 'specular_reflectance', 'roughness'):
                tex = getattr(m, name)
                mm[name] = {'levels': [put(l, device) for l in tex.mipmap], 'constant': tex.constant,
                            'uv_scale': put(tex.uv_scale, device)}
            for name in ('generic_texture', 'normal_map'):
                tex = getattr(m, name)
                mm[name] = None if tex is None else {
                    'levels': [put(l, device) for l in tex.mipmap], 'constant': tex.constant,
                    'uv_scale': put(tex.uv_scale, device)}
            meta['materials'].append(mm)
        meta['lights'] = [{'shape_id': l.shape_id, 'intensity': put(l.intensity, cpu), 'two_sided': l.two_sided,
                           'directly_visible': l.directly_visible} for l in scene.area_lights]
        meta['envmap'] = None
        if scene.envmap is not None:
            em = scene.envmap
            meta['envmap'] = {'levels': [put(l, device) for l in em.values.mipmap], 'uv_scale': put(em.values.uv_scale, device),
                              'env_to_world': put(em.env_to_world, cpu), 'world_to_env': put(em.world_to_env, cpu),
                              'sample_cdf_ys': put(em.sample_cdf_ys, device), 'sample_cdf_xs': put(em.sample_cdf_xs, device),
                              'pdf_norm': em.pdf_norm, 'directly_visible': em.directly_visible}
        meta['use_primary_edge_sampling'] = bool(use_primary_edge_sampling and needs_visibility)
        meta['use_secondary_edge_sampling'] = bool(use_secondary_edge_sampling and needs_visibility)
        if to_check:
            # max |x| per tensor in one multi-tensor launch (NaN propagates through max, inf stays inf), one read-back
            live = [t.detach() for t in to_check if t.numel() > 0]
            if live:
                peaks = torch.stack(torch._foreach_norm(live, float('inf')))
                assert bool(torch.isfinite(peaks).all()), 'serialize_scene: a scene tensor holds non-finite values'
            for t in to_check:
                if not t.requires_grad and t.numel() >= _FINITE_CACHE_MIN_ELEMS:
                    _remember_finite(t)
        return [meta] + tensors

    @staticmethod
    def unpack_args(seed, meta, tensors):
        """Rebuild the redner.* objects from raw data_ptr()s (render_pytorch.py:272-649)."""
        rd = meta['backend']
        u = _Unpacked()
        u.keep = []                                   # temporaries whose storage the C++ side reads

        def T(i):
            return tensors[i] if i >= 0 else None

        def fp(t):
            return rd.float_ptr(_data_ptr(t))

        def ip(t):
            return rd.int_ptr(_data_ptr(t))

        cm = meta['camera']
        res, vp = cm['resolution'], cm['viewport']
        use_look_at = cm['cam_to_world'] < 0
        u.camera = rd.Camera(res[1], res[0],
                             fp(T(cm['position']) if use_look_at else None),
                             fp(T(cm['look_at']) if use_look_at else None),
                             fp(T(cm['up']) if use_look_at else None),
                             fp(None if use_look_at else T(cm['cam_to_world'])),
                             fp(None if use_look_at else T(cm['world_to_cam'])),
                             fp(T(cm['intrinsic_mat_inv'])), fp(T(cm['intrinsic_mat'])), fp(T(cm['distortion_params'])),
                             cm['clip_near'], rd.CameraType(int(cm['camera_type'])),
                             rd.Vector2i(vp[1], vp[0]), rd.Vector2i(vp[3], vp[2]))
        u.shapes = []
        for sm in meta['shapes']:
            v, idx = T(sm['vertices']), T(sm['indices'])
            uvs, nrm = T(sm['uvs']), T(sm['normals'])
            u.shapes.append(rd.Shape(fp(v), ip(idx), fp(uvs), fp(nrm), ip(T(sm['uv_indices'])),
                                     ip(T(sm['normal_indices'])), fp(T(sm['colors'])),
                                     int(v.shape[0]), int(uvs.shape[0]) if uvs is not None else 0,
                                     int(nrm.shape[0]) if nrm is not None else 0, int(idx.shape[0]),
                                     sm['material_id'], sm['light_id']))
        u.materials = []

        def make_tex(cls, tm):
            if tm is None:
                return cls([], [], [], 0, rd.float_ptr(0))
            levels = [T(i) for i in tm['levels']]
            if tm['constant']:
                return cls([fp(levels[0])], [0], [0], int(levels[0].shape[0]), fp(T(tm['uv_scale'])))
            return cls([fp(l) for l in levels], [int(l.shape[1]) for l in levels], [int(l.shape[0]) for l in levels],
                       int(levels[0].shape[2]), fp(T(tm['uv_scale'])))

        for mm in meta['materials']:
            u.materials.append(rd.Material(make_tex(rd.Texture3, mm['diffuse_reflectance']),
                                           make_tex(rd.Texture3, mm['specular_reflectance']),
                                           make_tex(rd.Texture1, mm['roughness']),
                                           make_tex(rd.TextureN, mm['generic_texture']),
                                           make_tex(rd.Texture3, mm['normal_map']),
                                           mm['compute_specular_lighting'], mm['two_sided'], mm['use_vertex_color']))
        u.area_lights = [rd.AreaLight(lm['shape_id'], fp(T(lm['intensity'])), lm['two_sided'], lm['directly_visible'])
                         for lm in meta['lights']]
        device = meta['device']
        index = device.index if device.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        u.envmap = None
        if meta['envmap'] is not None:
            em = meta['envmap']
            levels = [T(i) for i in em['levels']]
            values = rd.Texture3([fp(l) for l in levels], [int(l.shape[1]) for l in levels], [int(l.shape[0]) for l in levels],
                                 3, fp(T(em['uv_scale'])))
            u.envmap = rd.EnvironmentMap(values, fp(T(em['env_to_world'])), fp(T(em['world_to_env'])),
                                         fp(T(em['sample_cdf_ys'])), fp(T(em['sample_cdf_xs'])), em['pdf_norm'],
                                         em['directly_visible'])
        u.scene = rd.Scene(u.camera, u.shapes, u.materials, u.area_lights, u.envmap, device.type == 'cuda', index,
                           meta['use_primary_edge_sampling'], meta['use_secondary_edge_sampling'])
        u.options = rd.RenderOptions(seed[0], meta['num_samples'][0], meta['max_bounces'], meta['channels'],
                                     meta['sampler_type'], meta['sample_pixel_center'])
        if meta.get('tuning') and hasattr(u.options, 'tuning'):      # rdr_tuning (redner_amd extension; the reference's options have none)
            known = {f[0] for f in type(u.options.tuning)._fields_}
            unknown = sorted(set(meta['tuning']) - known)
            if unknown:                        # a ctypes.Structure takes any attribute name: a typo would silently render with defaults
                raise ValueError('unknown rdr_tuning field(s) %s; known: %s' % (unknown, sorted(known)))
            for k, v in meta['tuning'].items():
                setattr(u.options.tuning, k, v)
        if 'sample_offset' in meta:             # multi-GPU sample sharding (redner_amd extension)
            u.options.sample_offset = meta['sample_offset'][0]
            u.options.total_samples = meta['total_samples'][0]
        return u

    @staticmethod
    def forward(ctx, seed, meta, *tensors):
        rd = meta['backend']
        if isinstance(seed, int):
            seed = (seed, seed + 1000003)          # backward uses an independent stream (render_pytorch.py:659-663)
        u = RenderFunction.unpack_args(seed, meta, tensors)
        vp = meta['camera']['viewport']
        nc = rd.compute_num_channels(meta['channels'], u.scene.max_generic_texture_dimension)
        img = torch.zeros(vp[2] - vp[0], vp[3] - vp[1], nc, device=meta['device'])
        rd.render(u.scene, u.options, rd.float_ptr(img.data_ptr()), rd.float_ptr(0), None, rd.float_ptr(0), rd.float_ptr(0))
        ctx.u, ctx.meta, ctx.tensors, ctx.seed = u, meta, tensors, seed
        return img

    @staticmethod
    def create_gradient_buffers(meta, tensors):
        """Zero-initialised gradient tensors mirroring the inputs + the redner.DScene that points at them
        (pyredner/render_pytorch.py:710-980 create_gradient_buffers).  -> (d_scene, grads)"""
        rd = meta['backend']
        device = meta['device']
        grads = [None] * len(tensors)
        # one zero-filled allocation for all of them (a scene has ~70 tensors: one fill instead of ~70), 64-byte aligned slices
        sizes = [((t.numel() + 15) // 16) * 16 for t in tensors]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        starts = [0] * len(tensors)
        for k in range(1, len(tensors)):
            starts[k] = starts[k - 1] + sizes[k - 1]

        def zeros_like_arg(i):
            if i < 0:
                return None
            g = flat[starts[i]:starts[i] + tensors[i].numel()].view(tensors[i].shape)
            grads[i] = g
            return g

        def fp(t):
            return rd.float_ptr(_data_ptr(t))

        cm = meta['camera']
        use_look_at = cm['cam_to_world'] < 0
        d_camera = rd.DCamera(fp(zeros_like_arg(cm['position']) if use_look_at else None),
                              fp(zeros_like_arg(cm['look_at']) if use_look_at else None),
                              fp(zeros_like_arg(cm['up']) if use_look_at else None),
                              fp(None if use_look_at else zeros_like_arg(cm['cam_to_world'])),
                              fp(None if use_look_at else zeros_like_arg(cm['world_to_cam'])),
                              fp(zeros_like_arg(cm['intrinsic_mat_inv'])), fp(zeros_like_arg(cm['intrinsic_mat'])),
                              fp(zeros_like_arg(cm['distortion_params'])))
        d_shapes = [rd.DShape(fp(zeros_like_arg(sm['vertices'])), fp(zeros_like_arg(sm['uvs'])),
                              fp(zeros_like_arg(sm['normals'])), fp(zeros_like_arg(sm['colors'])))
                    for sm in meta['shapes']]

        def d_tex(cls, tm):
            if tm is None:
                return cls([], [], [], 0, rd.float_ptr(0))
            levels = [zeros_like_arg(i) for i in tm['levels']]
            sc = zeros_like_arg(tm['uv_scale'])
            if tm['constant']:
                return cls([fp(levels[0])], [0], [0], int(levels[0].shape[0]), fp(sc))
            return cls([fp(l) for l in levels], [int(l.shape[1]) for l in levels], [int(l.shape[0]) for l in levels],
                       int(levels[0].shape[2]), fp(sc))

        d_materials = [rd.DMaterial(d_tex(rd.Texture3, mm['diffuse_reflectance']),
                                    d_tex(rd.Texture3, mm['specular_reflectance']),
                                    d_tex(rd.Texture1, mm['roughness']),
                                    d_tex(rd.TextureN, mm['generic_texture']),
                                    d_tex(rd.Texture3, mm['normal_map'])) for mm in meta['materials']]
        d_lights = [rd.DAreaLight(fp(zeros_like_arg(lm['intensity']))) for lm in meta['lights']]
        index = device.index if device.index is not None else 0
        d_envmap = None
        if meta['envmap'] is not None:
            em = meta['envmap']
            levels = [zeros_like_arg(i) for i in em['levels']]
            d_values = rd.Texture3([fp(l) for l in levels], [int(l.shape[1]) for l in levels], [int(l.shape[0]) for l in levels],
                                   3, fp(zeros_like_arg(em['uv_scale'])))
            d_envmap = rd.DEnvironmentMap(d_values, fp(zeros_like_arg(em['world_to_env'])))
        d_scene = rd.DScene(d_camera, d_shapes, d_materials, d_lights, d_envmap, device.type == 'cuda', index)
        return d_scene, grads

    @staticmethod
    def backward(ctx, grad_img):
        meta, tensors, u = ctx.meta, ctx.tensors, ctx.u
        rd = meta['backend']
        grad_img = grad_img.contiguous()
        assert torch.isfinite(grad_img).all()
        d_scene, grads = RenderFunction.create_gradient_buffers(meta, tensors)
        u.options.seed = ctx.seed[1]
        u.options.num_samples = meta['num_samples'][1]
        if 'sample_offset' in meta:
            u.options.sample_offset = meta['sample_offset'][1]
            u.options.total_samples = meta['total_samples'][1]
        rd.render(u.scene, u.options, rd.float_ptr(0), rd.float_ptr(grad_img.data_ptr()), d_scene,
                  rd.float_ptr(0), rd.float_ptr(0))
        out = []
        for t, g in zip(tensors, grads):
            out.append(g.to(t.device) if g is not None and t.is_floating_point() else None)
        return (None, None) + tuple(out)

    @staticmethod
    def visualize_screen_gradient(grad_img, seed, scene, num_samples, max_bounces, use_primary_edge_sampling=True,
                                  use_secondary_edge_sampling=True, **kw):
        """Two-channel image of d(pixel colour)/d(screen position) (pyredner/render_pytorch.py:983-1048): one backward
        render() with a screen_gradient_image; fed by d_primary_intersection (src/primary_intersection.cpp:111-114) and the
        primary-edge estimator (src/edge.cpp:765-773).  grad_img None = all ones.  kw as for serialize_scene."""
        args = RenderFunction.serialize_scene(scene, num_samples, max_bounces, **kw)
        meta, tensors = args[0], args[1:]
        rd = meta['backend']
        # the reference passes the two flags straight to unpack_args here (:1014-1015), whatever requires grad
        meta['use_primary_edge_sampling'] = bool(use_primary_edge_sampling)
        meta['use_secondary_edge_sampling'] = bool(use_secondary_edge_sampling)
        u = RenderFunction.unpack_args((seed, seed), meta, tensors)
        d_scene, _grads = RenderFunction.create_gradient_buffers(meta, tensors)
        vp = meta['camera']['viewport']
        nc = rd.compute_num_channels(meta['channels'], u.scene.max_generic_texture_dimension)
        h, w = vp[2] - vp[0], vp[3] - vp[1]
        screen_gradient_image = torch.zeros(h, w, 2, device=meta['device'])
        if grad_img is None:
            grad_img = torch.ones(h, w, nc, device=meta['device'])
        grad_img = grad_img.to(meta['device'], torch.float32).contiguous()      # what the native side reads: fp32 on that device
        assert grad_img.shape == (h, w, nc)
        if not torch.isfinite(grad_img).all():
            raise ValueError('visualize_screen_gradient: grad_img is not finite')
        rd.render(u.scene, u.options, rd.float_ptr(0), rd.float_ptr(grad_img.data_ptr()), d_scene,
                  rd.float_ptr(screen_gradient_image.data_ptr()), rd.float_ptr(0))
        return screen_gradient_image


def render(scene, seed, num_samples, max_bounces, **kw):
    """Convenience wrapper: serialize + apply."""
    args = RenderFunction.serialize_scene(scene, num_samples, max_bounces, **kw)
    return RenderFunction.apply(seed, *args)
