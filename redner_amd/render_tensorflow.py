"""TensorFlow (2.x, eager) surface of the MI355X-native renderer.

Host-side mirror of the reference's `pyredner_tensorflow/render_tensorflow.py` for the hot path:
`serialize_scene` (:72-260), `forward` (:659-712), `render` -- the `tf.custom_gradient` operator (:998-1152) -- and
`visualize_screen_gradient` (:1154-1224), with the same names, argument meanings and defaults:

    args = serialize_scene(scene, num_samples=4, max_bounces=1, sampler_type=redner.SamplerType.sobol)
    with tf.GradientTape() as tape:
        img = render(seed, *args)
        loss = tf.reduce_sum(img)
    grads = tape.gradient(loss, [shape.vertices, ...])

How tensors cross the boundary.  The reference reads raw addresses through a compiled TensorFlow op
(`pyredner_tensorflow/custom_ops/data_ptr.cc`); here an eager tensor is viewed IN PLACE as a torch tensor through DLPack
(`tf.experimental.dlpack.to_dlpack` -> `torch.utils.dlpack.from_dlpack`: no copy, no custom op to build against a TensorFlow
installation), and from there on the call is the one the PyTorch surface makes (`render_pytorch.RenderFunction.unpack_args /
create_gradient_buffers`, i.e. the same `redner.*` constructors over the C ABI).  The image and the gradient tensors go back
the same way.  What differs from the reference, and why:
 * integer tensors (index buffers) are held on the host by TensorFlow even under a GPU device scope (the reference's
   `tf.bitcast` work-around, :194-199); they are serialized on the CPU device and copied to the GPU by torch;
 * TensorFlow runs its kernels on its own streams: the device is synchronised once before a render reads its inputs (the
   library returns synchronised);
 * whether a TensorFlow tensor "requires a gradient" is unknown at serialization time (the reference's TODO, :152): like the
   reference, both edge estimators are on whenever their flags are.

`import redner_amd.render_tensorflow` needs TensorFlow and fails without it; nothing else in the package imports it.  This
image has no TensorFlow: the module is written against the TF 2.x eager API and exercised here through the torch-backed
stand-in of the ~40 `tf.*` functions it calls (`tests/tf_standin/`, test infrastructure only), on the CPU harness and on
the GPU -- where the stand-in's tensors are HIP device tensors and the render is libredner_amd.so's.  With TensorFlow-ROCm
installed the DLPack capsules carry `kDLROCM`, which torch-ROCm accepts.

The scene classes below carry only what crosses the boundary (like `render_pytorch`'s); loaders, mip-map generation and image
IO stay the reference's pure-Python `pyredner_tensorflow` package (out of scope).
"""
import math
import time
from typing import List, Optional

import tensorflow as tf
import torch
from torch.utils import dlpack as _torch_dlpack

from . import redner as _default_backend
from .render_pytorch import RenderFunction as _Core

# ---- pyredner_tensorflow/device.py: which TensorFlow device the scene tensors live on ----
_use_gpu = None
_gpu_device_id = 0
_cpu_device_id = 0


def set_use_gpu(v: bool):
    global _use_gpu
    _use_gpu = bool(v)


def get_use_gpu():
    global _use_gpu
    if _use_gpu is None:
        _use_gpu = torch.cuda.is_available()
    return _use_gpu


def set_gpu_device_id(did: int):
    global _gpu_device_id
    _gpu_device_id = int(did)


def get_cpu_device_id():
    return _cpu_device_id


def get_device_name():
    return '/device:gpu:%d' % _gpu_device_id if get_use_gpu() else '/device:cpu:%d' % _cpu_device_id


# ---- render_tensorflow.py:13-60 ----
_use_correlated_random_number = False
_print_timing = False


def set_use_correlated_random_number(v: bool):
    """False (default): the backward pass draws from a stream of its own (seed + 1000003, :1026-1028)."""
    global _use_correlated_random_number
    _use_correlated_random_number = bool(v)


def get_use_correlated_random_number():
    return _use_correlated_random_number


def set_print_timing(v: bool):
    global _print_timing
    _print_timing = bool(v)


def get_print_timing():
    return _print_timing


class Context:
    pass


# ---- minimal scene classes over TensorFlow tensors (pyredner_tensorflow/camera.py, shape.py, texture.py, material.py,
#      area_light.py, envmap.py, scene.py: the attributes serialize_scene reads) ----
class Camera:
    def __init__(self, position=None, look_at=None, up=None, fov=None, clip_near=1e-4, resolution=(256, 256),
                 viewport=None, cam_to_world=None, intrinsic_mat=None, camera_type=0, distortion_params=None):
        self.position, self.look_at, self.up = position, look_at, up
        self.cam_to_world = cam_to_world
        self.world_to_cam = tf.linalg.inv(cam_to_world) if cam_to_world is not None else None
        if intrinsic_mat is None:
            fov = tf.reshape(tf.cast(tf.convert_to_tensor(fov), tf.float32), [1])
            fov_factor = 1.0 / tf.tan((math.pi / 180.0) * (0.5 * fov))            # pyredner_tensorflow/camera.py: fp32
            intrinsic_mat = tf.linalg.diag(tf.concat([fov_factor, fov_factor, tf.ones([1], dtype=tf.float32)], 0))
        self.intrinsic_mat = intrinsic_mat
        self.intrinsic_mat_inv = tf.linalg.inv(intrinsic_mat)
        self.clip_near = clip_near
        self.resolution = tuple(resolution)
        self.viewport = viewport
        self.camera_type = camera_type
        self.distortion_params = distortion_params


class Shape:
    def __init__(self, vertices, indices, material_id, uvs=None, normals=None, uv_indices=None, normal_indices=None,
                 colors=None):
        self.vertices, self.indices = vertices, indices
        self.uvs, self.normals = uvs, normals
        self.uv_indices, self.normal_indices = uv_indices, normal_indices
        self.colors = colors
        self.material_id = material_id
        self.light_id = -1


class Texture:
    """A constant (1-D tensor) or mip levels [H, W, C], finest first."""

    def __init__(self, texels, uv_scale=None):
        if tf.is_tensor(texels):
            self.mipmap = [texels]
        else:
            self.mipmap = list(texels)
        self.texels = self.mipmap[0]
        self.uv_scale = uv_scale if uv_scale is not None else tf.constant([1.0, 1.0])


class Material:
    def __init__(self, diffuse_reflectance=None, specular_reflectance=None, roughness=None, generic_texture=None,
                 normal_map=None, two_sided=False, use_vertex_color=False):
        def tex(t, default):
            if t is None:
                t = tf.constant(default)
            return t if isinstance(t, Texture) else Texture(t)
        self.diffuse_reflectance = tex(diffuse_reflectance, [0.0, 0.0, 0.0])
        self.compute_specular_lighting = specular_reflectance is not None        # pyredner_tensorflow/material.py
        self.specular_reflectance = tex(specular_reflectance, [0.0, 0.0, 0.0])
        self.roughness = tex(roughness, [1.0])
        self.generic_texture = tex(generic_texture, None) if generic_texture is not None else None
        self.normal_map = tex(normal_map, None) if normal_map is not None else None
        self.two_sided = two_sided
        self.use_vertex_color = use_vertex_color


class AreaLight:
    def __init__(self, shape_id, intensity, two_sided=False, directly_visible=True):
        self.shape_id, self.intensity = shape_id, intensity
        self.two_sided, self.directly_visible = two_sided, directly_visible


class EnvironmentMap:
    """Latitude-longitude environment light; the sampling tables as pyredner_tensorflow/envmap.py:37-68 builds them."""

    def __init__(self, values, env_to_world=None, directly_visible=True):
        self.values = values if isinstance(values, Texture) else Texture(values)
        self.env_to_world = env_to_world if env_to_world is not None else tf.eye(4, 4)
        self.world_to_env = tf.linalg.inv(self.env_to_world)
        self.directly_visible = directly_visible
        t = self.values.mipmap[0]
        h, w = int(t.shape[0]), int(t.shape[1])
        # on the render device, like the reference (pyredner_tensorflow/envmap.py:37): a float32 sine made on the host differs
        # from the device's in the last bit, and with it the tables
        with tf.device(get_device_name()):
            t = tf.identity(t)                 # the copy on the render device (a texture handed over as a host tensor stays a host tensor)
            lum = 0.212671 * t[:, :, 0] + 0.715160 * t[:, :, 1] + 0.072169 * t[:, :, 2]
            cdf_xs_ = tf.cumsum(lum, axis=1)
            y_weight = tf.sin(math.pi * (tf.cast(tf.range(h), tf.float32) + 0.5) / float(h))
            cdf_ys_ = tf.cumsum(cdf_xs_[:, -1] * y_weight, axis=0)
            self.pdf_norm = (h * w) / (float(cdf_ys_[-1]) * (2 * math.pi * math.pi))
            self.sample_cdf_xs = (cdf_xs_ - cdf_xs_[:, 0:1]) / tf.maximum(cdf_xs_[:, (w - 1):w], 1e-8 * tf.ones([h, 1], dtype=tf.float32))
            self.sample_cdf_ys = (cdf_ys_ - cdf_ys_[0]) / tf.maximum(cdf_ys_[-1], tf.constant([1e-8]))


class Scene:
    def __init__(self, camera, shapes, materials, area_lights, envmap=None):
        self.camera, self.shapes, self.materials, self.area_lights = camera, shapes, materials, area_lights
        self.envmap = envmap


# ---- the boundary: eager tensor <-> torch tensor, in place ----
def _as_torch(x):
    return _torch_dlpack.from_dlpack(tf.experimental.dlpack.to_dlpack(x))


def _as_tf(t):
    return tf.experimental.dlpack.from_dlpack(_torch_dlpack.to_dlpack(t.contiguous()))


def _torch_device(device_name):
    spec = tf.DeviceSpec.from_string(device_name)
    if (spec.device_type or 'CPU').upper() == 'GPU':
        return torch.device('cuda', spec.device_index if spec.device_index is not None else 0)
    return torch.device('cpu')


def serialize_scene(scene, num_samples, max_bounces, channels: Optional[List] = None, sampler_type=None,
                    use_primary_edge_sampling=True, use_secondary_edge_sampling=True, sample_pixel_center: bool = False,
                    device_name: Optional[str] = None, backend=None, tuning=None) -> List:
    """Flatten a scene into [meta, tensor, tensor, ...] for `render(seed, *args)` (render_tensorflow.py:72-260).
    Every tensor of the list is `tf.identity` of the scene's (under the device it must live on), so a GradientTape that
    watches the scene's tensors reaches them through the list.  `backend`, `tuning`: redner_amd extensions, as in
    render_pytorch.serialize_scene."""
    backend = backend or _default_backend
    if device_name is None:
        device_name = get_device_name()
    host_name = '/device:cpu:%d' % get_cpu_device_id()
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

    tensors, on_host = [], []

    def put(t, host, integer=False):
        if t is None:
            return -1
        # index buffers: serialized on the host whatever the target (TensorFlow keeps int32 there), forward() moves them
        with tf.device(host_name if (host or integer) else device_name):
            x = tf.cast(t, tf.int32) if integer else tf.identity(t)
        if not integer:
            assert x.dtype == tf.float32, 'scene tensors are float32 (pyredner_tensorflow asserts the same)'
        tensors.append(x)
        on_host.append(bool(host))
        return len(tensors) - 1

    cam = scene.camera
    meta = {'backend': backend, 'device': _torch_device(device_name), 'device_name': device_name,
            'num_samples': tuple(num_samples), 'max_bounces': max_bounces, 'channels': list(channels),
            'sampler_type': sampler_type, 'sample_pixel_center': sample_pixel_center}
    if tuning:
        meta['tuning'] = dict(tuning)
    cm = {}
    for name in ('position', 'look_at', 'up', 'cam_to_world', 'world_to_cam', 'intrinsic_mat_inv', 'intrinsic_mat',
                 'distortion_params'):
        cm[name] = put(getattr(cam, name), True)                    # camera tensors are host tensors (:165-187)
    cm['clip_near'], cm['resolution'], cm['camera_type'] = cam.clip_near, tuple(cam.resolution), cam.camera_type
    vp = cam.viewport if cam.viewport is not None else (0, 0, cam.resolution[0], cam.resolution[1])
    cm['viewport'] = (max(vp[0], 0), max(vp[1], 0), min(vp[2], cam.resolution[0]), min(vp[3], cam.resolution[1]))
    meta['camera'] = cm
    meta['shapes'] = [{
        'vertices': put(sh.vertices, False), 'indices': put(sh.indices, False, True),
        'uvs': put(sh.uvs, False), 'normals': put(sh.normals, False),
        'uv_indices': put(sh.uv_indices, False, True), 'normal_indices': put(sh.normal_indices, False, True),
        'colors': put(sh.colors, False), 'material_id': sh.material_id, 'light_id': sh.light_id} for sh in scene.shapes]

    def put_texture(tex):
        if tex is None:
            return None
        return {'levels': [put(l, False) for l in tex.mipmap], 'constant': len(tex.mipmap[0].shape) == 1,
                'uv_scale': put(tex.uv_scale, False)}

    meta['materials'] = [{
        'diffuse_reflectance': put_texture(m.diffuse_reflectance), 'specular_reflectance': put_texture(m.specular_reflectance),
        'roughness': put_texture(m.roughness), 'generic_texture': put_texture(m.generic_texture),
        'normal_map': put_texture(m.normal_map), 'compute_specular_lighting': m.compute_specular_lighting,
        'two_sided': m.two_sided, 'use_vertex_color': m.use_vertex_color} for m in scene.materials]
    meta['lights'] = [{'shape_id': l.shape_id, 'intensity': put(l.intensity, True), 'two_sided': l.two_sided,
                       'directly_visible': l.directly_visible} for l in scene.area_lights]
    meta['envmap'] = None
    if scene.envmap is not None:
        em = scene.envmap
        meta['envmap'] = {'levels': [put(l, False) for l in em.values.mipmap], 'uv_scale': put(em.values.uv_scale, False),
                          'env_to_world': put(em.env_to_world, True), 'world_to_env': put(em.world_to_env, True),
                          'sample_cdf_ys': put(em.sample_cdf_ys, False), 'sample_cdf_xs': put(em.sample_cdf_xs, False),
                          'pdf_norm': float(em.pdf_norm), 'directly_visible': em.directly_visible}
    meta['use_primary_edge_sampling'] = bool(use_primary_edge_sampling)
    meta['use_secondary_edge_sampling'] = bool(use_secondary_edge_sampling)
    meta['on_host'] = on_host
    return [meta] + tensors


def _torch_views(meta, xs):
    """The serialized tensors as torch tensors where the native side reads them: host slots on the host, the others on
    meta['device'] (in place when TensorFlow already holds them there); float tensors checked for non-finite values, the device
    ones by one multi-tensor kernel and one read-back (pyredner_tensorflow asserts tf.reduce_all(tf.math.is_finite(...)) per
    tensor)."""
    dev = meta['device']
    out, to_check = [], []
    for x, host in zip(xs, meta['on_host']):
        t = _as_torch(x).to(torch.device('cpu') if host else dev).contiguous()
        if t.is_floating_point():
            assert t.dtype == torch.float32
            if t.device.type == 'cpu':
                assert bool(torch.isfinite(t).all()), 'render: a scene tensor holds non-finite values'
            elif t.numel() > 0:
                to_check.append(t)
        else:
            assert t.dtype == torch.int32
        out.append(t)
    if to_check:
        peaks = torch.stack(torch._foreach_norm(to_check, float('inf')))
        assert bool(torch.isfinite(peaks).all()), 'render: a scene tensor holds non-finite values'
    return out


def _wait_for_tensorflow(meta):
    # TensorFlow's kernels run on TensorFlow's streams; the render reads its inputs on the library's
    if meta['device'].type == 'cuda':
        torch.cuda.synchronize(meta['device'])


def forward(seed: int, *args):
    """Forward pass (render_tensorflow.py:659-712): serialized scene -> (image, ctx)."""
    assert tf.executing_eagerly()
    meta, xs = args[0], args[1:]
    rd = meta['backend']
    seed = int(seed)
    seeds = (seed, seed if get_use_correlated_random_number() else seed + 1000003)
    tensors = _torch_views(meta, xs)
    start = time.time()
    u = _Core.unpack_args(seeds, meta, tensors)
    if get_print_timing():
        print('Scene construction, time: %.5f s' % (time.time() - start))
    vp = meta['camera']['viewport']
    nc = rd.compute_num_channels(meta['channels'], u.scene.max_generic_texture_dimension)
    img = torch.zeros(vp[2] - vp[0], vp[3] - vp[1], nc, device=meta['device'])
    _wait_for_tensorflow(meta)
    start = time.time()
    rd.render(u.scene, u.options, rd.float_ptr(img.data_ptr()), rd.float_ptr(0), None, rd.float_ptr(0), rd.float_ptr(0))
    if get_print_timing():
        print('Forward pass, time: %.5f s' % (time.time() - start))
    ctx = Context()
    ctx.u, ctx.meta, ctx.tensors, ctx.seeds = u, meta, tensors, seeds
    ctx.args = args                    # keeps the TensorFlow tensors (whose memory `tensors` views) alive
    return _as_tf(img), ctx


def _backward(ctx, grad_img):
    meta, tensors, u = ctx.meta, ctx.tensors, ctx.u
    rd = meta['backend']
    g = _as_torch(tf.identity(grad_img)).to(meta['device'], torch.float32).contiguous()
    assert bool(torch.isfinite(g).all())
    d_scene, grads = _Core.create_gradient_buffers(meta, tensors)
    u.options.seed = ctx.seeds[1]
    u.options.num_samples = meta['num_samples'][1]
    _wait_for_tensorflow(meta)
    start = time.time()
    rd.render(u.scene, u.options, rd.float_ptr(0), rd.float_ptr(g.data_ptr()), d_scene, rd.float_ptr(0), rd.float_ptr(0))
    if get_print_timing():
        print('Backward pass, time: %.5f s' % (time.time() - start))
    out = []
    for t, d in zip(tensors, grads):
        # one gradient per serialized tensor, where the tensor lives; None for index buffers and for what the renderer does not
        # differentiate (env_to_world, the sampling tables: render_tensorflow.py:1045-1150 returns None for the same)
        out.append(_as_tf(d.to(t.device)) if d is not None and t.is_floating_point() else None)
    return out


def render(seed, *args):
    """The operator (render_tensorflow.py:998-1152): image = render(seed, *serialize_scene(...)), differentiable with
    respect to every float tensor of the list through tf.custom_gradient."""
    meta, xs = args[0], args[1:]

    @tf.custom_gradient
    def _render(*tensors):
        img, ctx = forward(seed, meta, *tensors)

        def backward(grad_img):
            return _backward(ctx, grad_img)
        return img, backward

    return _render(*xs)


def visualize_screen_gradient(grad_img, seed: int, scene, num_samples, max_bounces: int, channels: Optional[List] = None,
                              sampler_type=None, use_primary_edge_sampling: bool = True,
                              use_secondary_edge_sampling: bool = True, sample_pixel_center: bool = False, **kw):
    """Two-channel image of d(pixel colour)/d(screen position) (render_tensorflow.py:1154-1224).  grad_img None = ones."""
    args = serialize_scene(scene, num_samples, max_bounces, channels=channels, sampler_type=sampler_type,
                           use_primary_edge_sampling=use_primary_edge_sampling,
                           use_secondary_edge_sampling=use_secondary_edge_sampling and max_bounces > 0,
                           sample_pixel_center=sample_pixel_center, **kw)
    meta, xs = args[0], args[1:]
    rd = meta['backend']
    tensors = _torch_views(meta, xs)
    u = _Core.unpack_args((int(seed), int(seed)), meta, tensors)
    d_scene, _grads = _Core.create_gradient_buffers(meta, tensors)
    vp = meta['camera']['viewport']
    nc = rd.compute_num_channels(meta['channels'], u.scene.max_generic_texture_dimension)
    h, w = vp[2] - vp[0], vp[3] - vp[1]
    screen_gradient_image = torch.zeros(h, w, 2, device=meta['device'])
    if grad_img is None:
        g = torch.ones(h, w, nc, device=meta['device'])
    else:
        g = _as_torch(tf.identity(grad_img)).to(meta['device'], torch.float32).contiguous()
    assert tuple(g.shape) == (h, w, nc)
    if not bool(torch.isfinite(g).all()):
        raise ValueError('visualize_screen_gradient: grad_img is not finite')
    _wait_for_tensorflow(meta)
    start = time.time()
    rd.render(u.scene, u.options, rd.float_ptr(0), rd.float_ptr(g.data_ptr()), d_scene,
              rd.float_ptr(screen_gradient_image.data_ptr()), rd.float_ptr(0))
    if get_print_timing():
        print('Visualize gradient, time: %.5f s' % (time.time() - start))
    return _as_tf(screen_gradient_image)
