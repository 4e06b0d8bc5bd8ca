#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the BUILD container only).

Needs /root/reference (the unmodified pyredner package, to parse tests/scenes/bunny_box.xml) and
the oracle build (oracle/_ref, `make -C oracle`).  Two kinds of files are written:

  bunny_box_scene.npz        raw mesh/material/camera arrays of the reference's bunny_box scene,
                             exactly as pyredner.load_mitsuba hands them to redner.Scene
                             (tests/test_bunny_box.py) -- INPUT data for tests and bench.py
  <case>.npz                 oracle outputs (forward image, every gradient tensor) for the parity
                             cases listed in CASES, rendered through tests/scenes.py with the
                             oracle as backend

The GPU box has neither /root/reference nor network; tests read only these files.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import oracle_util  # noqa: E402

REF = '/root/reference'


def export_bunny_box():
    ref = oracle_util.load_oracle()
    sys.modules['redner'] = ref
    sys.path[:0] = [os.path.join(ROOT, 'oracle', 'pystubs'), REF]
    import pyredner
    pyredner.set_use_gpu(False)
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, 'tests'))
    scene = pyredner.load_mitsuba('scenes/bunny_box.xml')
    os.chdir(cwd)
    out = {'num_shapes': len(scene.shapes), 'num_materials': len(scene.materials)}
    cam = scene.camera
    if cam.cam_to_world is not None:
        out['cam_to_world'] = cam.cam_to_world.numpy()
    else:
        out['cam_position'], out['cam_look_at'], out['cam_up'] = cam.position.numpy(), cam.look_at.numpy(), cam.up.numpy()
    out['intrinsic_mat'] = cam.intrinsic_mat.numpy()
    out['clip_near'] = cam.clip_near
    bunny = max(range(len(scene.shapes)), key=lambda i: scene.shapes[i].indices.shape[0])
    out['bunny_shape_id'] = bunny
    for i, sh in enumerate(scene.shapes):
        out['shape%d_vertices' % i] = sh.vertices.numpy()
        out['shape%d_indices' % i] = sh.indices.numpy().astype(np.int32)
        if sh.uvs is not None:
            out['shape%d_uvs' % i] = sh.uvs.numpy()
        if sh.normals is not None:
            out['shape%d_normals' % i] = sh.normals.numpy()
        assert sh.uv_indices is None and sh.normal_indices is None and sh.colors is None
        out['shape%d_material_id' % i] = sh.material_id
    for i, m in enumerate(scene.materials):
        assert m.diffuse_reflectance.texels.dim() == 1 and m.specular_reflectance.texels.dim() == 1
        assert m.roughness.texels.dim() == 1 and m.normal_map is None and m.generic_texture is None
        out['mat%d_diffuse' % i] = m.diffuse_reflectance.texels.numpy()
        out['mat%d_specular' % i] = m.specular_reflectance.texels.numpy()
        out['mat%d_roughness' % i] = m.roughness.texels.numpy()
        out['mat%d_compute_specular_lighting' % i] = m.compute_specular_lighting
        out['mat%d_two_sided' % i] = m.two_sided
    assert len(scene.area_lights) == 1 and scene.envmap is None
    l = scene.area_lights[0]
    out['light0_shape_id'], out['light0_intensity'], out['light0_two_sided'] = l.shape_id, l.intensity.numpy(), l.two_sided
    np.savez_compressed(os.path.join(HERE, 'bunny_box_scene.npz'), **out)
    print('bunny_box_scene.npz: %d shapes, %d triangles' % (len(scene.shapes), sum(s.indices.shape[0] for s in scene.shapes)))


ALL_CHANNELS = ['radiance', 'alpha', 'depth', 'position', 'geometry_normal', 'shading_normal', 'uv',
                'barycentric_coordinates', 'diffuse_reflectance', 'specular_reflectance', 'roughness',
                'generic_texture', 'vertex_color', 'shape_id', 'triangle_id', 'material_id']

EDGE_SAFE_CHANNELS = [c for c in ALL_CHANNELS if c not in ('barycentric_coordinates', 'generic_texture')]

# name -> (scene builder, resolution, spp, max_bounces[, channel names[, serialize_scene options]])
CASES = {
    'single_triangle_64x64x4': ('single_triangle', 64, 4, 1),
    'two_triangles_64x64x16': ('two_triangles', 64, 16, 1),
    # sample indices above 255: the Sobol' XOR takes its loop for the bits beyond the low byte (csrc/sobol.h: sobol_value)
    'single_triangle_hi_index_12x12x320': ('single_triangle', 12, 320, 1),
    'bunny_box_32x32x4': ('bunny_box', 32, 4, 4),
    # big enough that every scheduling feature of the GPU build is on: side streams, the second sample worker (8 spp),
    # wave-summed gradient scatters
    'bunny_box_96x96x8': ('bunny_box', 96, 8, 4),
    # every output channel at once (tests/test_g_buffer.py renders them in groups) + mip-mapped textures
    'textured_sphere_gbuffer_48x48x4': ('textured_sphere', 48, 4, 1, EDGE_SAFE_CHANNELS),
    # The reference itself cannot run these two channels with edge sampling: its generic-texture scratch is
    # sized for num_pixels lanes but the edge pass indexes 2*num_pixels (src/pathtracer.cpp:103 vs :829,
    # heap overflow), and barycentric_coordinates segfaults in its edge pass.  Pinned without edge sampling.
    # the default `independent` (PCG32) sampler; 3 bounces so the edge sampler's per-slot states diverge
    'bunny_box_pcg_32x32x3': ('bunny_box', 32, 3, 3, None, {'sampler': 'independent'}),
    'two_triangles_pcg_64x64x4': ('two_triangles', 64, 4, 1, None, {'sampler': 'independent'}),
    # camera models and lens distortion (src/camera.h, src/camera_distortion.h)
    'two_triangles_ortho_64x64x4': ('two_triangles_ortho', 64, 4, 1),
    'two_triangles_distorted_64x64x4': ('two_triangles_distorted', 64, 4, 1),
    'bunny_box_fisheye_32x32x4': ('bunny_box_fisheye', 32, 4, 2),
    'bunny_box_panorama_32x32x4': ('bunny_box_panorama', 32, 4, 2),
    # the same two without secondary edge sampling: what a GPU run could match sample for sample before its transcendental
    # functions were glibc's (see CHAOTIC_PICK_CASES below)
    'bunny_box_fisheye_nosec_32x32x4': ('bunny_box_fisheye', 32, 4, 2, None, {'use_secondary_edge_sampling': False}),
    'bunny_box_panorama_nosec_32x32x4': ('bunny_box_panorama', 32, 4, 2, None, {'use_secondary_edge_sampling': False}),
    # environment light only: NEE / BSDF-miss lookups, envmap adjoint, edge rays that reach the environment
    'envmap_sphere_48x48x4': ('envmap_sphere', 48, 4, 2),
    # a convex object under the environment: path depths 1 and 2 have no live lanes (Sobol' dimension bookkeeping of the edge
    # sampler across dead depths)
    'envmap_convex_48x48x4': ('envmap_convex', 48, 4, 3),
    # separate uv / normal index buffers, two lights (two-sided; not directly visible), non-square image, viewport,
    # samples at pixel centres
    'misc_features_40x56x4': ('misc_features', (40, 56), 4, 2),
    'misc_features_viewport_40x56x4': ('misc_features_viewport', (40, 56), 4, 2, None, {'sample_pixel_center': True}),
    # radiance after a 3-wide channel: the reference adds path contributions at the channel INDEX (src/channels.cpp:27)
    'textured_sphere_radiance_last_48x48x2': ('textured_sphere', 48, 2, 2, ['position', 'radiance']),
    # ... and after id channels: the bounce contributions land ON the ids (triangle / material id + what the last sample's paths
    # carried), which a batched forward render has to reproduce on top of its assigned ids
    'textured_sphere_ids_radiance_last_48x48x3': ('textured_sphere', 48, 3, 2,
                                                  ['alpha', 'position', 'shape_id', 'triangle_id', 'material_id', 'radiance']),
    'textured_sphere_generic_48x48x4': ('textured_sphere', 48, 4, 1,
                                        ['radiance', 'barycentric_coordinates', 'generic_texture'],
                                        {'use_primary_edge_sampling': False, 'use_secondary_edge_sampling': False}),
    # BASELINE config 5 stand-in (tests/scenes.py: living_room_standin) at parity-test size: textured two-sided bunny_box,
    # max_bounces 6, camera-pose gradients; and its environment-map + SVBRDF-texture variant
    'living_room_standin_40x40x2': ('living_room_standin', 40, 2, 6),
    'living_room_standin_envmap_32x32x2': ('living_room_standin_envmap', 32, 2, 6),
    # a triangle that crosses the near plane, light behind the camera (tests/test_single_triangle_clipped.py)
    'triangle_through_near_plane_64x64x4': ('triangle_through_near_plane', 64, 4, 1),
    # near-mirror floor reflecting a light and a blocker (tests/test_shadow_glossy.py)
    'glossy_floor_blocker_48x48x4': ('glossy_floor_blocker', 48, 4, 2),
    # material / texture / light optimisation: neither edge estimator (what pyredner selects when neither the camera nor a vertex
    # requires a gradient) -- the gradient render of these mip-mapped / environment-lit scenes is then batched like a plain one
    'living_room_standin_envmap_noedges_32x32x4': ('living_room_standin_envmap', 32, 4, 6, None,
                                                   {'use_primary_edge_sampling': False, 'use_secondary_edge_sampling': False}),
    'envmap_sphere_noedges_48x48x4': ('envmap_sphere', 48, 4, 2, None,
                                      {'use_primary_edge_sampling': False, 'use_secondary_edge_sampling': False}),
    'misc_features_noedges_40x56x4': ('misc_features', (40, 56), 4, 2, None,
                                      {'use_primary_edge_sampling': False, 'use_secondary_edge_sampling': False}),
    # render_albedo/render_g_buffer style: no radiance, no bounces (pyredner/render_utils.py)
    'textured_sphere_albedo_48x48x4': ('textured_sphere', 48, 4, 0,
                                       ['depth', 'shading_normal', 'diffuse_reflectance', 'uv']),
}


# BASELINE configs at their real sizes: config 2 in full; config 3 as a full 512 x 512 frame at
# reduced spp plus a full-resolution 128 x 128 viewport tile at the full 128 spp.  GPU tests only (tests/test_config_parity.py):
# the single-threaded CPU harness would need minutes per case.
CONFIG_CASES = {
    'two_triangles_256x256x64': ('two_triangles', 256, 64, 1),
    'bunny_box_512x512x8': ('bunny_box', 512, 8, 4),
    'bunny_box_tile_512x512x128': ('bunny_box_tile', 512, 128, 4),
    # BASELINE config 5's stand-in (tests/scenes.py) at a size where the mid-specialised kernels run with full waves,
    # both sample-independent streams and mip-mapped texture gradients (texel tensors: 512 x 512 x 3 per material)
    'living_room_standin_256x256x4': ('living_room_standin', 256, 4, 6),
    # ... and its environment-map + SVBRDF-texture variant at 8 spp (round 4): two sample batches of four on the GPU -- chain mode
    # (mip levels), the replay of stale hit positions (edge rays that leave the open front of the room), the general kernels
    'living_room_standin_envmap_256x256x8': ('living_room_standin_envmap', 256, 8, 6),
}

# Cases whose backward pass is reproduced sample for sample only by a build whose sin / cos / atan2 return the oracle's
# bits.  The reference's hierarchical edge pick threads ONE random number through the whole tree walk, rescaling it at
# every node (src/edge.cpp:1160-1230): ~100 rescalings amplify a 1-ulp difference in the shading position to O(1), so the
# pick is chaotic in its inputs.  With a perspective / orthographic camera the first-hit positions involve only
# + - * / sqrt; fisheye / panorama primary rays go through sin / cos / atan2, where the device's own libm and glibc differ
# in the last ulp.  Rounds 1-3 checked these two on the GPU only statistically (tests/test_statistical_parity.py); since
# round 4 the kernels compute the seven transcendental functions of the path as glibc does, bit for bit
# (redner_amd/csrc/libm_exact.h, tests/test_libm_exact.py), and the GPU is held to these fixtures like to every other.
CHAOTIC_PICK_CASES = {'bunny_box_fisheye_32x32x4', 'bunny_box_panorama_32x32x4'}


def render_case(backend, builder, res, spp, mb, channels=None, opts=None, device=torch.device('cpu'), stripe=None, blocks=None,
                seed=1):
    """Forward + backward of one case; returns {'image': ..., 'grad_<i>_<name>': ...}.
    stripe = (k, K): the upstream gradient is zeroed except on pixels k, k + K, ... (row-major) -- see
    oracle_self_inconsistency.  blocks = R: the samples are rendered as R contiguous blocks (sample_offset = b * spp / R)
    and summed in block order -- the single-device counterpart of an R-rank run (redner_amd/distributed.py)."""
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    sc = getattr(scenes, builder)(device, resolution=res if isinstance(res, tuple) else (res, res))
    for l in sc.area_lights:
        l.intensity.requires_grad_(True)
    for m in sc.materials:
        m.diffuse_reflectance.mipmap[0].requires_grad_(True)
    if sc.camera.position is not None:
        sc.camera.position.requires_grad_(True)
        if sc.camera.camera_type != 0 or sc.camera.distortion_params is not None:
            for t in (sc.camera.look_at, sc.camera.up, sc.camera.intrinsic_mat, sc.camera.intrinsic_mat_inv):
                t.requires_grad_(True)
    ch = None if channels is None else [getattr(backend.channels, c) for c in channels]
    opts = dict(opts or {})
    sampler = getattr(backend.SamplerType, opts.pop('sampler', 'sobol'))
    args = RenderFunction.serialize_scene(sc, spp, mb, channels=ch, sampler_type=sampler,
                                          device=device, backend=backend, **opts)
    if blocks:
        from redner_amd.distributed import render_blocked
        img = render_blocked(seed, args, blocks)
    else:
        img = RenderFunction.apply(seed, *args)
    out = {'image': img.detach().cpu().numpy()}
    # upstream gradient: a fixed smooth pattern so every pixel/channel has a distinct weight
    h, w, c = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    base = [1.0 + 0.5 * torch.sin(0.37 * xx + 0.11 * yy), 1.0 + 0.5 * torch.cos(0.23 * yy),
            1.0 - 0.3 * torch.sin(0.19 * (xx + yy))]
    up = torch.stack([base[k % 3] * (1.0 + 0.25 * (k // 3)) for k in range(c)], dim=2)
    if stripe is not None:
        keep = torch.zeros(h * w)
        keep[stripe[0]::stripe[1]] = 1
        up = up * keep.reshape(h, w, 1)
    up = up.to(img.device)
    (img * up).sum().backward()

    def grab(key, t):
        if t is not None and t.grad is not None:
            out[key] = t.grad.cpu().numpy()
    for i, sh in enumerate(sc.shapes):
        grab('grad_shape%d_vertices' % i, sh.vertices)
        for name in ('uvs', 'normals', 'colors'):
            grab('grad_shape%d_%s' % (i, name), getattr(sh, name))
    for i, l in enumerate(sc.area_lights):
        grab('grad_light%d_intensity' % i, l.intensity)
    for i, m in enumerate(sc.materials):
        grab('grad_mat%d_diffuse' % i, m.diffuse_reflectance.mipmap[0])
        for name, tex in (('diffuse', m.diffuse_reflectance), ('specular', m.specular_reflectance),
                          ('roughness', m.roughness), ('generic', m.generic_texture), ('normal_map', m.normal_map)):
            if tex is None:
                continue
            for lv, t in enumerate(tex.mipmap):
                if name != 'diffuse' or lv > 0:
                    grab('grad_mat%d_%s_L%d' % (i, name, lv), t)
    if getattr(sc, 'envmap', None) is not None:
        for lv, t in enumerate(sc.envmap.values.mipmap):
            grab('grad_envmap_L%d' % lv, t)
        grab('grad_envmap_env_to_world', sc.envmap.env_to_world)
    if sc.camera.position is not None:
        grab('grad_cam_position', sc.camera.position)
    for name in ('look_at', 'up', 'intrinsic_mat', 'intrinsic_mat_inv', 'distortion_params'):
        t = getattr(sc.camera, name)
        if t is not None and t.requires_grad:
            grab('grad_cam_' + name, t)
    return out


# Screen-space gradient images (RenderFunction.visualize_screen_gradient, tests/test_screen_gradient.py: a G-buffer
# channel at max_bounces 0 with the default sampler); name -> (builder, resolution, spp, max_bounces, channels, options)
SCREEN_GRADIENT_CASES = {
    # the reference's script: one reflectance channel, no bounces, PCG sampler, both edge estimators on
    'screengrad_textured_sphere_albedo_48x48x4': ('textured_sphere', 48, 4, 0, ['diffuse_reflectance'], {'sampler': 'independent'}),
    # radiance with one bounce: camera-vertex adjoint + primary edges both add to the image
    'screengrad_two_triangles_64x64x4': ('two_triangles', 64, 4, 1, None, {}),
    # several channels of different widths, pixel-centre samples
    'screengrad_textured_sphere_gbuffer_48x48x2': ('textured_sphere', 48, 2, 0, ['depth', 'shading_normal', 'uv'],
                                                   {'sample_pixel_center': True}),
}


def screen_gradient_case(backend, builder, res, spp, mb, channels=None, opts=None, device=torch.device('cpu')):
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    sc = getattr(scenes, builder)(device, resolution=res if isinstance(res, tuple) else (res, res))
    ch = None if channels is None else [getattr(backend.channels, c) for c in channels]
    opts = dict(opts or {})
    sampler = getattr(backend.SamplerType, opts.pop('sampler', 'sobol'))
    img = RenderFunction.visualize_screen_gradient(None, 3, sc, spp, mb, channels=ch, sampler_type=sampler, device=device,
                                                   backend=backend, **opts)
    return {'screen_gradient': img.cpu().numpy()}


# ---- statistical fixtures for the cases a GPU cannot reproduce sample for sample (CHAOTIC_PICK_CASES) -------------------
# Per seed, a vector of linear functionals of the gradient: the translation gradient of the bunny (3), 8 fixed random
# projections of its vertex gradient, light intensity (3), camera position (3).  tests/test_statistical_parity.py compares
# the GPU's per-seed vectors with the oracle's: both are draws of the same estimator on the same Sobol' points, differing only
# in the chaotic edge picks, so their means must agree within Monte-Carlo error.
STAT_CASES = {
    'stat_bunny_box_fisheye_32x32x4': ('bunny_box_fisheye', 32, 4, 2),
    'stat_bunny_box_panorama_32x32x4': ('bunny_box_panorama', 32, 4, 2),
}
STAT_SEEDS = list(range(1, 25))


def stat_case(backend, builder, res, spp, mb, device=torch.device('cpu'), seeds=STAT_SEEDS):
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    rows = []
    proj = None
    for seed in seeds:
        sc = getattr(scenes, builder)(device, resolution=(res, res))
        for l in sc.area_lights:
            l.intensity.requires_grad_(True)
        sc.camera.position.requires_grad_(True)
        args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=backend.SamplerType.sobol, device=device, backend=backend)
        img = RenderFunction.apply(seed, *args)
        img.sum().backward()
        verts = [sh.vertices for sh in sc.shapes if sh.vertices.grad is not None]
        g = verts[0].grad.double().cpu().numpy()
        if proj is None:
            proj = np.random.RandomState(12345).standard_normal((8,) + g.shape)
        row = list(g.sum(0)) + [float((p * g).sum()) for p in proj]
        row += list(sc.area_lights[0].intensity.grad.double().cpu().numpy()) + list(sc.camera.position.grad.double().cpu().numpy())
        row.append(float(img.detach().double().mean()))
        rows.append(row)
    return {'stats': np.asarray(rows, np.float64)}


def oracle_self_inconsistency(ref, case, K=4):
    """How far the reference's own value of a gradient tensor moves under an exactly equivalent evaluation order.

    The backward pass is linear in the upstream image gradient and draws the same samples whatever that gradient is, so
    the sum of K backward passes, each seeing the upstream gradient on every K-th pixel only, is the same estimator on
    the same samples.  The reference adds every contribution into the caller's fp32 tensors with fp32 atomics
    (src/atomic.h:43-141); for tensors with a few elements and millions of contributions (light intensity, constant
    reflectances, camera) that accumulation loses 1e-4 ... 1e-2 of the value at the config sizes -- adds below half an ulp
    of the running sum vanish, large cancelling terms leave their rounding behind -- and the two evaluations disagree by
    that much.  Stored with the fixture as selfdiff_<tensor> = rel-L2(one pass, sum of K striped passes); the parity
    tests widen the 1e-4 bar to 3 x selfdiff for those few-element accumulators only (tests/parity_util.py).
    Vertex / texel gradients (few contributions per element) agree to ~1e-7 between the two evaluations."""
    one = render_case(ref, *case)
    acc = None
    for k in range(K):
        part = render_case(ref, *case, stripe=(k, K))
        part = {n: v.astype(np.float64) for n, v in part.items() if n != 'image'}
        acc = part if acc is None else {n: acc[n] + part[n] for n in acc}
    out = {}
    for n, v in acc.items():
        a = one[n].astype(np.float64)
        na = np.linalg.norm(a)
        out['selfdiff_' + n] = np.float64(np.linalg.norm(v - a) / na if na > 0 else 0.0)
    return one, out


SELFDIFF_CASES = set(CONFIG_CASES) | {'bunny_box_96x96x8'}

# BASELINE config 4's own frame (1024 x 1024, max_bounces 4) at 1 spp: the 12.6 MB image is not committed -- its SHA-256 (the
# forward image is bit-identical to the oracle's in every case), 16 x 16 block sums of it, and the bunny's vertex gradient are.
FULL_FRAME_CASE = ('bunny_box_1024x1024x1', ('bunny_box', 1024, 1, 4))
# ... and at 16 spp: 16.8 M lanes = ONE 16-sample batch of the GPU build (2^24 lanes) -- the launch shape bench.py times
# (15.3 M-ray queues, the refilling traversal kernel, per-sample segment tables of the secondary-edge sampler).  ~10 min of
# oracle time on 8 cores.
# ... BASELINE config 3 at its quoted size as a FULL frame (tests/test_bunny_box.py at 512 x 512 x 128 spp: 33.5 M samples, ~20 min
# of oracle time on 8 cores; round 6), and config 4's frame at 64 of its 256 spp (67 M samples, ~40 min): 8 GPU batches of 8
FULL_FRAME_CASES = dict([FULL_FRAME_CASE, ('bunny_box_1024x1024x16', ('bunny_box', 1024, 16, 4)),
                         ('bunny_box_512x512x128', ('bunny_box', 512, 128, 4)),
                         ('bunny_box_1024x1024x64', ('bunny_box', 1024, 64, 4))])


def full_frame_digest(out):
    import hashlib
    img = np.ascontiguousarray(out['image'], dtype=np.float32)
    h, w, c = img.shape
    blocks = img.astype(np.float64).reshape(h // 16, 16, w // 16, 16, c).sum(axis=(1, 3))
    return {'image_sha256': np.frombuffer(hashlib.sha256(img.tobytes()).digest(), dtype=np.uint8).copy(), 'image_block_sums': blocks,
            'grad_shape6_vertices': out['grad_shape6_vertices']}

def full_frame_check(backend, name, device):
    """-> (image equal bit for bit?, largest block-sum difference, rel-L2 of the bunny's vertex gradient); also used by bench.py"""
    gold = np.load(os.path.join(HERE, name + '.npz'))
    mine = full_frame_digest(render_case(backend, *FULL_FRAME_CASES[name], device=device))
    blocks_err = float(np.abs(mine['image_block_sums'] - gold['image_block_sums']).max())
    g, m = gold['grad_shape6_vertices'].astype(np.float64), mine['grad_shape6_vertices'].astype(np.float64)
    return bool(np.array_equal(mine['image_sha256'], gold['image_sha256'])), blocks_err, float(np.linalg.norm(m - g) / np.linalg.norm(g))


# ---- ref64: the oracle's estimator with the fp32 accumulation error taken out ------------------------------------------------
# Few-element gradient tensors (light intensity, constant reflectances, camera, the 3 + 4 vertices of two_triangles) collect
# millions of fp32 atomic adds per element in ONE backward pass of the reference (src/atomic.h:43-141), and the value it returns
# moves by 1e-4 ... 1e-2 with the order of those adds (selfdiff above).  The backward pass is linear in the upstream gradient and
# draws the same samples whatever it is, so the SAME estimator on the SAME samples is also the sum of K passes that each see the
# upstream gradient on every K-th pixel only; each pass's fp32 accumulators then take K times fewer non-zero adds (adding an
# exact 0 is exact) and the K results are summed in fp64.  ref64_<tensor> = that sum for K = 64, ref64conv_<tensor> = rel-L2
# between the K = 16 and the K = 64 sums (what is left of the accumulation error: it shrinks with K).  The parity tests hold
# every tensor that has a ref64_ entry to 1e-4 against it (tests/parity_util.py) -- no widened bar.
# bunny_box_512x512x8 also carries ref256_<tensor> / ref256conv_<tensor> (K = 256, and rel-L2 between the K = 64 and K = 256 sums):
# its camera gradient -- three numbers of 5e5, sums of 1.7e7 cancelling terms -- is the one tensor whose K = 64 sum has not
# settled to 1e-4; `python make_golden.py --ref256 bunny_box_512x512x8` (4 h of oracle time on 8 cores).
REF64_K = (16, 64)
REF64_MAX_ELEMS = 4096        # larger tensors (per-vertex / per-texel data) take few adds per element and meet 1e-4 as they are


def oracle_striped_sum(ref, case, K):
    acc = None
    for k in range(K):
        part = render_case(ref, *case, stripe=(k, K))
        part = {n: v.astype(np.float64) for n, v in part.items() if n != 'image' and v.size <= REF64_MAX_ELEMS}
        acc = part if acc is None else {n: acc[n] + part[n] for n in acc}
    return acc


def add_ref64(ref, name, case):
    """Adds ref64_* / ref64conv_* to an existing fixture (its single-pass tensors stay as they are)."""
    path = os.path.join(HERE, name + '.npz')
    z = np.load(path)
    out = {k: z[k] for k in z.files if not k.startswith('ref64')}
    lo, hi = (oracle_striped_sum(ref, case, K) for K in REF64_K)
    for n, v in hi.items():
        nv = np.linalg.norm(v)
        out['ref64_' + n] = v
        out['ref64conv_' + n] = np.float64(np.linalg.norm(lo[n] - v) / nv if nv > 0 else 0.0)
        one = out[n].astype(np.float64)
        print('  %-28s single pass vs ref64 %.2e   K=%d vs K=%d %.2e' % (n, np.linalg.norm(one - v) / nv if nv > 0 else 0.0,
                                                                         REF64_K[0], REF64_K[1], float(out['ref64conv_' + n])), flush=True)
    np.savez_compressed(path, **out)


def main():
    # The reference's primary-edge pass reads ray differentials from a scratch buffer at indices it never
    # wrote (slot- vs lane-indexed, src/edge.cpp:608 vs src/scene.cpp:585), i.e. whatever malloc returned.
    # Force every large allocation onto fresh zero pages so the fixtures do not depend on heap history;
    # this only matters for scenes with mip-mapped textures.  glibc reads the variable at start-up.
    # (MALLOC_PERTURB_=255, round 4: glibc zero-fills every chunk it hands out, also the ones below the threshold)
    if os.environ.get('MALLOC_MMAP_THRESHOLD_') != '65536' or os.environ.get('MALLOC_PERTURB_') != '255':
        os.environ['MALLOC_MMAP_THRESHOLD_'] = '65536'
        os.environ['MALLOC_PERTURB_'] = '255'
        os.execv(sys.executable, [sys.executable] + sys.argv)
    if not os.path.exists(os.path.join(HERE, 'bunny_box_scene.npz')) or '--scene' in sys.argv:
        export_bunny_box()
    ref = oracle_util.load_oracle()
    only = [a for a in sys.argv[1:] if not a.startswith('--')]
    if '--full-frame' in sys.argv:                     # python make_golden.py --full-frame [case ...]
        for name, case in FULL_FRAME_CASES.items():
            if only and name not in only:
                continue
            np.savez_compressed(os.path.join(HERE, name + '.npz'), **full_frame_digest(render_case(ref, *case)))
            print(name, 'written', flush=True)
        return
    if '--ref256' in sys.argv:
        for name in only:
            path = os.path.join(HERE, name + '.npz')
            z = np.load(path)
            out = {k: z[k] for k in z.files}
            hi = oracle_striped_sum(ref, (CASES.get(name) or CONFIG_CASES[name]), 256)
            for n, v in hi.items():
                nv = np.linalg.norm(v)
                out['ref256_' + n] = v
                out['ref256conv_' + n] = np.float64(np.linalg.norm(out['ref64_' + n] - v) / nv if nv > 0 else 0.0)
            np.savez_compressed(path, **out)
        return
    if '--ref64' in sys.argv:                          # python make_golden.py --ref64 [case ...]: tens of minutes per config case
        for name in sorted(SELFDIFF_CASES):
            if only and name not in only:
                continue
            print(name, flush=True)
            add_ref64(ref, name, (CASES.get(name) or CONFIG_CASES[name]))
        return
    for name, case in list(CASES.items()) + list(CONFIG_CASES.items()):
        if (only and name not in only) or (not only and name in CONFIG_CASES and os.path.exists(os.path.join(HERE, name + '.npz'))):
            continue                                   # the config-size fixtures take minutes: made once, or on request
        if name in SELFDIFF_CASES:
            out, sd = oracle_self_inconsistency(ref, case)
            print(name, {k: '%.2e' % float(v) for k, v in sd.items()})
            out.update(sd)
        else:
            out = render_case(ref, *case)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, {k: v.shape for k, v in out.items()})
    for name, case in STAT_CASES.items():
        if only and name not in only:
            continue
        out = stat_case(ref, *case)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, out['stats'].shape, out['stats'].mean(0)[:4], out['stats'].std(0)[:4])
    for name, case in SCREEN_GRADIENT_CASES.items():
        if only and name not in only:
            continue
        out = screen_gradient_case(ref, *case)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, {k: (v.shape, float(np.abs(v).sum())) for k, v in out.items()})


if __name__ == '__main__':
    main()
