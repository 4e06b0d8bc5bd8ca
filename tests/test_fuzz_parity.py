"""Seeded random scenes through the CPU harness (the stage bodies and host driver of the GPU build) against the LIVE oracle
(the reference's C++ core, oracle/_ref): image bit for bit, every gradient tensor to 1e-4.  Triangle soups with random
diffuse / glossy / two-sided materials, one or two area lights, a jittered camera -- interpenetrating, partly back-facing,
partly behind each other, which is what the fixed fixtures do not have; a second family adds shading normals, mip-mapped
textures, environment maps and orthographic cameras.  Needs the built oracle (build container only).

The comparisons run in a subprocess started with MALLOC_MMAP_THRESHOLD_=1024: the reference's primary-edge pass reads ray
differentials at indices it never wrote (src/edge.cpp:608 vs src/scene.cpp:585), i.e. whatever malloc returned -- at these
frame sizes recycled heap memory, and then the texture level of a few edge samples depends on the heap's history (found by this
test: 28 of 60 textured scenes off by up to 10 % on single pixels, none with fresh zero pages; tests/golden/make_golden.py does
the same for the fixtures)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_util
import scenes
from redner_amd.render_pytorch import Camera, Shape, Material, AreaLight, Scene, RenderFunction, Texture, EnvironmentMap

pytestmark = pytest.mark.skipif(not oracle_util.oracle_available(), reason='oracle not built')

# Where `redner_amd` renders: the CPU harness (build container) or cuda:0 (the GPU leg, `_main(..., gpu=True)`).  Scenes are
# always BUILT on the CPU -- leaf tensors and their gradients live there -- and serialize_scene() moves the data to the device
# of the backend that renders it; the oracle always renders on the CPU.
MINE_DEVICE = torch.device('cpu')
ON_GPU = False


def _sdev(backend):
    return MINE_DEVICE if getattr(backend, '__name__', '').startswith('redner_amd') else torch.device('cpu')


def _np(t):
    return t.detach().cpu().numpy()


def _scene(seed, device):
    rng = np.random.RandomState(seed)
    mats = []
    for k in range(3):
        glossy = rng.rand() < 0.5
        mats.append(Material(
            diffuse_reflectance=torch.tensor(rng.uniform(0.1, 0.8, 3).astype(np.float32), device=device, requires_grad=True),
            specular_reflectance=torch.tensor((rng.uniform(0.05, 0.5, 3) if glossy else np.zeros(3)).astype(np.float32),
                                              device=device, requires_grad=glossy),
            roughness=torch.tensor([float(rng.uniform(0.05, 0.6)) if glossy else 1.0], device=device, requires_grad=glossy),
            two_sided=bool(rng.rand() < 0.5)))
    mats.append(Material(diffuse_reflectance=torch.zeros(3, device=device)))
    shapes = []
    for k in range(3):
        nt = int(rng.randint(2, 7))
        centres = rng.uniform([-1.5, -1.5, -0.5], [1.5, 1.5, 1.5], (nt, 1, 3))
        verts = (centres + rng.uniform(-0.9, 0.9, (nt, 3, 3))).reshape(-1, 3).astype(np.float32)
        idx = np.arange(3 * nt, dtype=np.int32).reshape(nt, 3)
        shapes.append(Shape(torch.tensor(verts, device=device, requires_grad=True), torch.tensor(idx, device=device), k))
    lights = []
    for k in range(int(rng.randint(1, 3))):
        c = rng.uniform([-2.0, -2.0, -7.0], [2.0, 2.0, -5.5])
        h = float(rng.uniform(0.4, 1.0))
        lv = np.array([[c[0] - h, c[1] - h, c[2]], [c[0] + h, c[1] - h, c[2]], [c[0] - h, c[1] + h, c[2]], [c[0] + h, c[1] + h, c[2]]],
                      np.float32)
        shapes.append(Shape(torch.tensor(lv, device=device), torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32, device=device), 3))
        lights.append(AreaLight(len(shapes) - 1, torch.tensor(rng.uniform(5.0, 25.0, 3).astype(np.float32), requires_grad=True),
                                two_sided=bool(rng.rand() < 0.3)))
    cam = Camera(position=torch.tensor((np.array([0.0, 0.0, -5.0]) + rng.uniform(-0.5, 0.5, 3)).astype(np.float32), requires_grad=True),
                 look_at=torch.tensor(rng.uniform(-0.3, 0.3, 3).astype(np.float32), requires_grad=True),
                 up=torch.tensor([0.0, 1.0, 0.0], requires_grad=True), fov=torch.tensor([float(rng.uniform(35.0, 60.0))]),
                 clip_near=1e-2, resolution=(20, 24))
    return Scene(cam, shapes, mats, lights)


def _render(backend, seed, spp, mb, stripe=None):
    dev = torch.device('cpu')
    sc = _scene(seed, dev)
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=backend.SamplerType.sobol, device=_sdev(backend), backend=backend)
    img = RenderFunction.apply(seed, *args)
    h, w, _ = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    up = torch.stack([1.0 + 0.5 * torch.sin(0.4 * xx + 0.2 * yy), 1.0 + 0.5 * torch.cos(0.3 * yy), 1.0 - 0.3 * torch.sin(0.2 * (xx + yy))], 2)
    if stripe is not None:                     # the upstream gradient on every K-th pixel only (see _few_element_check)
        keep = torch.zeros(h * w)
        keep[stripe[0]::stripe[1]] = 1
        up = up * keep.reshape(h, w, 1)
    (img * up.to(img.device)).sum().backward()
    out = {'image': _np(img)}
    for i, s in enumerate(sc.shapes):
        if s.vertices.grad is not None:
            out['shape%d' % i] = s.vertices.grad.numpy()
    for i, m in enumerate(sc.materials):
        for n, t in (('diffuse', m.diffuse_reflectance), ('specular', m.specular_reflectance), ('roughness', m.roughness)):
            t = t.mipmap[0] if hasattr(t, 'mipmap') else t
            if t is not None and t.grad is not None:
                out['mat%d_%s' % (i, n)] = t.grad.numpy()
    for i, l in enumerate(sc.area_lights):
        out['light%d' % i] = l.intensity.grad.numpy()
    for n in ('position', 'look_at', 'up'):
        out['cam_' + n] = getattr(sc.camera, n).grad.numpy()
    return out


FLIPS = {}          # GPU leg: scene -> tensors in which ONE OR TWO edge samples landed on other edges than the oracle's (see _edge_flip)
SCENE = [None]
SEEN = set()


def _edge_flip(k, m, r):
    """GPU leg only.  The hierarchical edge pick threads one random number through ~100 rescalings, so it is chaotic in the
    shading position, and positions that went through sin / cos / pow -- a glossy bounce -- can differ
    from the oracle's in the last bit, because the device's libm is not glibc.  Such a sample picks ANOTHER edge: a different,
    equally valid draw that moves the gradient of the two edges involved -- FOUR vertex rows, possibly of two shapes -- and
    nothing else.  A per-vertex tensor that fails the 1e-4 bar is counted as part of such a flip (not as a pass: flips are
    listed and budgeted by the caller) when the rows that differ by more than 1e-5 of the tensor's norm are at most four IN
    THE WHOLE SCENE, every other row of the tensor agrees to 1e-5, and the tensor has rows that agree."""
    if not ON_GPU or r.ndim != 2 or r.shape[0] < 3 or r.shape[1] != 3:
        return False
    n = np.linalg.norm(r.astype(np.float64))
    rows = np.linalg.norm(m.astype(np.float64) - r.astype(np.float64), axis=1)
    few = int((rows > 1e-5 * n).sum())
    so_far = sum(e['rows'] for kk, e in FLIPS.get(SCENE[0], {}).items() if kk != k)
    if 0 < few < r.shape[0] and so_far + few <= 4:
        FLIPS.setdefault(SCENE[0], {})[k] = {'rows': few, 'rel_l2': float(np.linalg.norm(rows) / max(n, 1e-300))}
        return True
    return False


def _distance(mine, ref, skip=()):
    """-> None if the two results agree (image bit for bit, tensors to 1e-4), else a description."""
    SEEN.add(SCENE[0])
    if set(mine) != set(ref):
        return 'keys differ'
    if not np.array_equal(mine['image'], ref['image']):
        return 'image differs'
    # every tensor to 1e-4 of ITS OWN norm -- also the three camera gradients (position / look_at / up: three projections of the
    # same per-pixel addends, of which `up` nearly cancels: norm 0.05 beside 1-5; rounds 3-4 held the three to the largest of
    # them).  The reference sums them with fp32 atomics whose rounding is relative to the ADDENDS, so its single pass carries
    # 1e-4 ... 3e-4 of noise in such a tensor: a failure here goes to _few_element_check (the oracle's striped fp64 sum).
    for k in ref:
        if k in skip:
            continue
        n = np.linalg.norm(ref[k].astype(np.float64))
        d = np.linalg.norm(mine[k].astype(np.float64) - ref[k].astype(np.float64))
        if not d <= 1e-4 * n + 1e-9:
            if _edge_flip(k, mine[k], ref[k]):
                continue
            return '%s: %.3e' % (k, d / max(n, 1e-300))
    return None


ORACLE_UNSTABLE = []


def _compare(mine, ref, render_oracle):
    """-> None or the first failure that the striped sum does not explain.  GPU leg: a failure is first checked against the
    ORACLE's own reproducibility -- the reference reads scratch it never wrote (module docstring); the legs run under
    MALLOC_MMAP_THRESHOLD_=1024 (large chunks are fresh zero mappings) AND MALLOC_PERTURB_=255 (glibc zero-fills every
    chunk: also the small ones, and the large ones should the process run out of mappings beside the HIP runtime), which
    makes that scratch read as zero; should a second oracle render of a failing scene still differ from the first, the
    scene is listed (ORACLE_UNSTABLE) and compared with the second render.  (Found the hard way: with an 8 KiB threshold and
    no zero fill the oracle's own two renders of four pixel-centre scenes differed by up to 30 % in a vertex gradient.)"""
    bad = _compare_once(mine, ref, render_oracle)
    # (round 6: also on the CPU legs -- two of three full CPU-suite runs failed ONE scene of ONE leg, a different leg each time,
    #  on a few-element tensor whose single pass and striped sum had moved together by 3e-4 between runs of the same scene
    #  against a deterministic harness: the oracle's run-to-run behaviour, not its fp32 noise)
    if bad and bad not in ('keys differ',):
        ref2 = render_oracle(None)
        if _distance(ref2, ref) is not None:
            ORACLE_UNSTABLE.append(SCENE[0])
            return _compare_once(mine, ref2, render_oracle)
    return bad


def _compare_once(mine, ref, render_oracle):
    excused = set()
    while True:
        bad = _distance(mine, ref, excused)
        if not bad or bad in ('keys differ', 'image differs'):
            return bad
        bad = _few_element_check(bad, mine, ref, render_oracle)
        if bad:
            return bad
        excused.add(_distance(mine, ref, excused).split(':')[0])


def _few_element_check(bad, mine, ref, render_oracle):
    """A tensor of a few elements (a constant roughness, a light's intensity) is a sum over all samples that the reference
    accumulates with fp32 atomics; where the sum nearly cancels, its single pass is itself only good to a few 1e-4 (the same
    effect the ref64 fixtures take out, tests/golden/make_golden.py).  Such a failure is re-examined against the oracle's
    fp64 sum of 16 pixel-striped passes, with the distance between that sum and the single pass added to the bar."""
    k = bad.split(':')[0]
    if k not in ref or ref[k].size > 16:
        return bad
    K = 16
    fine = sum(render_oracle((j, K))[k].astype(np.float64) for j in range(K))
    n = np.linalg.norm(fine)
    own = np.linalg.norm(ref[k].astype(np.float64) - fine) / n
    d = np.linalg.norm(mine[k].astype(np.float64) - fine) / n
    if d <= 1e-4 + own:
        return None
    # once more with fresh oracle passes (their fp32 atomics interleave differently every time): the better of the two sums decides
    fine2 = sum(render_oracle((j, K))[k].astype(np.float64) for j in range(K))
    n2 = np.linalg.norm(fine2)
    d2 = np.linalg.norm(mine[k].astype(np.float64) - fine2) / n2
    own2 = np.linalg.norm(ref[k].astype(np.float64) - fine2) / n2
    if d2 <= 1e-4 + own2:
        return None
    return '%s: %.3e / %.3e against two striped sums (oracle single pass: %.3e / %.3e)' % (k, d, d2, own, own2)


def _scene_rich(seed, device):
    """The soup of _scene plus what the general kernels are for: interpolated shading normals, uv coordinates with a
    mip-mapped diffuse texture and a roughness texture, sometimes an environment map (then possibly no area light at all),
    an orthographic camera now and then."""
    rng = np.random.RandomState(1000 + seed)
    sc = _scene(seed, device)

    def tex(c, lo, hi, size):
        t = torch.tensor(rng.uniform(lo, hi, (size, size, c)).astype(np.float32))
        return Texture([l.to(device).requires_grad_(True) for l in scenes._mip_chain(t)],
                       uv_scale=torch.tensor(rng.uniform(0.5, 3.0, 2).astype(np.float32), device=device, requires_grad=True))
    sc.materials[0].diffuse_reflectance = tex(3, 0.1, 0.8, 8)
    if rng.rand() < 0.5:
        sc.materials[1].specular_reflectance = tex(3, 0.05, 0.4, 4)
        sc.materials[1].roughness = tex(1, 0.1, 0.6, 4)
    for sh in sc.shapes[:3]:
        v = sh.vertices.detach().numpy().reshape(-1, 3, 3)
        n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        n = np.repeat(n[:, None, :], 3, 1) + rng.uniform(-0.3, 0.3, v.shape) * np.linalg.norm(n, axis=1)[:, None, None]
        n /= np.linalg.norm(n, axis=2, keepdims=True)
        if rng.rand() < 0.7:
            sh.normals = torch.tensor(n.reshape(-1, 3).astype(np.float32), device=device, requires_grad=True)
        sh.uvs = torch.tensor(rng.uniform(0.0, 1.0, (v.shape[0] * 3, 2)).astype(np.float32), device=device, requires_grad=True)
    env = None
    if rng.rand() < 0.5:
        img = torch.tensor(rng.uniform(0.05, 1.5, (8, 16, 3)).astype(np.float32))
        img[int(rng.randint(0, 8)), int(rng.randint(0, 16))] = 40.0
        env = EnvironmentMap(Texture([l.to(device).requires_grad_(True) for l in scenes._mip_chain(img)]))
        if rng.rand() < 0.4:                       # environment light only
            keep = [s for s in sc.shapes[:3]]
            sc = Scene(sc.camera, keep, sc.materials, [], envmap=env)
            return sc
    cam = sc.camera
    r = rng.rand()
    if r < 0.45:                               # orthographic / fisheye / panorama, sometimes through a viewport
        kind = 1 if r < 0.2 else (2 if r < 0.33 else 3)
        vp = (2, 3, 17, 22) if rng.rand() < 0.4 else None
        cam = Camera(position=cam.position, look_at=cam.look_at, up=cam.up, clip_near=cam.clip_near, resolution=cam.resolution,
                     camera_type=kind, fov=torch.tensor([45.0]), viewport=vp)
    return Scene(cam, sc.shapes, sc.materials, sc.area_lights, envmap=env)


def _render_rich(backend, seed, spp, mb, pixel_center, stripe=None):
    dev = torch.device('cpu')
    sc = _scene_rich(seed, dev)
    sampler = backend.SamplerType.independent if seed % 4 == 3 else backend.SamplerType.sobol      # PCG32 every fourth scene
    # Fisheye / panorama primary rays go through sin / cos / atan2 and the hierarchical edge pick is chaotic in the shading
    # position.  Rounds 1-3 compared those cameras on the GPU without the secondary-edge estimator; the
    # kernels' transcendental functions are glibc's now (csrc/libm_exact.h), so every camera runs with it.
    sec = True
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=sampler, device=_sdev(backend), backend=backend,
                                          sample_pixel_center=pixel_center, use_secondary_edge_sampling=sec)
    img = RenderFunction.apply(seed, *args)
    h, w, _ = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    up = torch.stack([1.0 + 0.5 * torch.sin(0.4 * xx + 0.2 * yy), 1.0 + 0.5 * torch.cos(0.3 * yy), 1.0 - 0.3 * torch.sin(0.2 * (xx + yy))], 2)
    if stripe is not None:                     # the upstream gradient on every K-th pixel only (see _few_element_check)
        keep = torch.zeros(h * w)
        keep[stripe[0]::stripe[1]] = 1
        up = up * keep.reshape(h, w, 1)
    (img * up.to(img.device)).sum().backward()
    out = {'image': _np(img)}
    for i, s in enumerate(sc.shapes):
        for n in ('vertices', 'normals', 'uvs'):
            t = getattr(s, n)
            if t is not None and t.grad is not None:
                out['shape%d_%s' % (i, n)] = t.grad.numpy()
    for i, m in enumerate(sc.materials):
        for n, t in (('diffuse', m.diffuse_reflectance), ('specular', m.specular_reflectance), ('roughness', m.roughness)):
            for lv, l in enumerate(t.mipmap if hasattr(t, 'mipmap') else [t]):
                if l is not None and l.grad is not None:
                    out['mat%d_%s_L%d' % (i, n, lv)] = l.grad.numpy()
            if hasattr(t, 'uv_scale') and t.uv_scale is not None and t.uv_scale.grad is not None:
                out['mat%d_%s_uv_scale' % (i, n)] = t.uv_scale.grad.numpy()
    for i, l in enumerate(sc.area_lights):
        out['light%d' % i] = l.intensity.grad.numpy()
    if sc.envmap is not None:
        for lv, l in enumerate(sc.envmap.values.mipmap):
            if l.grad is not None:
                out['envmap_L%d' % lv] = l.grad.numpy()
    for n in ('position', 'look_at', 'up'):
        out['cam_' + n] = getattr(sc.camera, n).grad.numpy()
    return out


def _mesh(rng, kind, centre, size):
    """Closed or open meshes with SHARED vertices (silhouette / dihedral logic of the edge list, src/edge.cpp:233-383)."""
    if kind == 0:          # tetrahedron
        v = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float64)
        f = [[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]]
    elif kind == 1:        # cube
        v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64)
        f = [[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]]
    else:                  # 4 x 4 height field
        n = 4
        g = np.linspace(-1, 1, n)
        v = np.array([[x, y, 0.3 * rng.uniform(-1, 1)] for y in g for x in g], np.float64)
        f = []
        for j in range(n - 1):
            for i in range(n - 1):
                a = j * n + i
                f += [[a, a + n, a + 1], [a + 1, a + n, a + n + 1]]
    # a random rotation
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return (centre + size * (v @ R.T)).astype(np.float32), np.asarray(f, np.int32)


def _scene_mesh(seed, device):
    rng = np.random.RandomState(5000 + seed)
    mats, shapes = [], []
    for k in range(3):
        glossy = rng.rand() < 0.4
        gen = None
        if rng.rand() < 0.4:
            gen = Texture([l.to(device).requires_grad_(True) for l in
                           scenes._mip_chain(torch.tensor(rng.uniform(0, 1, (4, 4, 5)).astype(np.float32)))])
        nmap = None
        if rng.rand() < 0.3:
            nm = rng.uniform(-0.3, 0.3, (4, 4, 3)); nm[:, :, 2] = 1.0
            nm = 0.5 + 0.5 * nm / np.linalg.norm(nm, axis=2, keepdims=True)
            nmap = Texture([l.to(device) for l in scenes._mip_chain(torch.tensor(nm.astype(np.float32)))])
        mats.append(Material(
            diffuse_reflectance=torch.tensor(rng.uniform(0.1, 0.8, 3).astype(np.float32), device=device, requires_grad=True),
            specular_reflectance=torch.tensor((rng.uniform(0.05, 0.5, 3) if glossy else np.zeros(3)).astype(np.float32), device=device),
            roughness=torch.tensor([float(rng.uniform(0.05, 0.6)) if glossy else 1.0], device=device),
            generic_texture=gen, normal_map=nmap, two_sided=bool(rng.rand() < 0.3), use_vertex_color=bool(rng.rand() < 0.3)))
    mats.append(Material(diffuse_reflectance=torch.zeros(3, device=device)))
    for k in range(int(rng.randint(2, 5))):
        v, f = _mesh(rng, int(rng.randint(0, 3)), rng.uniform([-1.6, -1.6, -0.3], [1.6, 1.6, 1.8]), float(rng.uniform(0.4, 0.9)))
        uv = torch.tensor(rng.uniform(0, 1, (len(v), 2)).astype(np.float32), device=device, requires_grad=True)
        col = torch.tensor(rng.uniform(0.1, 0.9, (len(v), 3)).astype(np.float32), device=device, requires_grad=True)
        shapes.append(Shape(torch.tensor(v, device=device, requires_grad=True), torch.tensor(f, device=device), int(rng.randint(0, 3)),
                            uvs=uv, colors=col))
    c = rng.uniform([-2.0, -2.0, -7.0], [2.0, 2.0, -5.5])
    lv = np.array([[c[0] - 0.8, c[1] - 0.8, c[2]], [c[0] + 0.8, c[1] - 0.8, c[2]], [c[0] - 0.8, c[1] + 0.8, c[2]], [c[0] + 0.8, c[1] + 0.8, c[2]]],
                  np.float32)
    shapes.append(Shape(torch.tensor(lv, device=device), torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32, device=device), 3))
    lights = [AreaLight(len(shapes) - 1, torch.tensor(rng.uniform(5.0, 25.0, 3).astype(np.float32), requires_grad=True),
                        two_sided=bool(rng.rand() < 0.3), directly_visible=bool(rng.rand() < 0.8))]
    dist = None
    if rng.rand() < 0.3:
        dist = torch.tensor(rng.uniform(-0.05, 0.05, 8).astype(np.float32), requires_grad=True)
    cam = Camera(position=torch.tensor((np.array([0.0, 0.0, -5.0]) + rng.uniform(-0.5, 0.5, 3)).astype(np.float32), requires_grad=True),
                 look_at=torch.tensor(rng.uniform(-0.3, 0.3, 3).astype(np.float32), requires_grad=True),
                 up=torch.tensor([0.0, 1.0, 0.0], requires_grad=True), fov=torch.tensor([float(rng.uniform(35.0, 60.0))]),
                 clip_near=1e-2, resolution=(22, 20), distortion_params=dist)
    return Scene(cam, shapes, mats, lights)


MESH_CHANNELS = ['radiance', 'alpha', 'depth', 'position', 'geometry_normal', 'shading_normal', 'uv', 'diffuse_reflectance',
                 'specular_reflectance', 'roughness', 'vertex_color', 'shape_id', 'triangle_id', 'material_id']


def _render_mesh(backend, seed, spp, mb, stripe=None):
    dev = torch.device('cpu')
    sc = _scene_mesh(seed, dev)
    rng = np.random.RandomState(9000 + seed)
    names = ['radiance'] + [c for c in MESH_CHANNELS[1:] if rng.rand() < 0.35] if seed % 2 else ['radiance']
    if rng.rand() < 0.3:
        names = names[1:] + names[:1]                 # radiance last
    ch = [getattr(backend.channels, c) for c in names]
    args = RenderFunction.serialize_scene(sc, spp, mb, channels=ch, sampler_type=backend.SamplerType.sobol, device=_sdev(backend), backend=backend)
    img = RenderFunction.apply(seed, *args)
    h, w, c = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    base = [1.0 + 0.5 * torch.sin(0.4 * xx + 0.2 * yy), 1.0 + 0.5 * torch.cos(0.3 * yy), 1.0 - 0.3 * torch.sin(0.2 * (xx + yy))]
    up = torch.stack([base[k % 3] * (1.0 + 0.1 * (k // 3)) for k in range(c)], 2)
    if stripe is not None:
        keep = torch.zeros(h * w)
        keep[stripe[0]::stripe[1]] = 1
        up = up * keep.reshape(h, w, 1)
    (img * up.to(img.device)).sum().backward()
    out = {'image': _np(img)}
    for i, s in enumerate(sc.shapes):
        for n in ('vertices', 'uvs', 'colors'):
            t = getattr(s, n)
            if t is not None and t.grad is not None:
                out['shape%d_%s' % (i, n)] = t.grad.numpy()
    for i, m in enumerate(sc.materials):
        for n, t in (('diffuse', m.diffuse_reflectance), ('generic', m.generic_texture)):
            if t is None:
                continue
            for lv, l in enumerate(t.mipmap):
                if l.grad is not None:
                    out['mat%d_%s_L%d' % (i, n, lv)] = l.grad.numpy()
    out['light0'] = sc.area_lights[0].intensity.grad.numpy()
    for n in ('position', 'look_at', 'up', 'distortion_params'):
        t = getattr(sc.camera, n)
        if t is not None and t.grad is not None:
            out['cam_' + n] = t.grad.numpy()
    return out


MESH_SEEDS = range(1 + int(os.environ.get('FUZZ_OFFSET', '0')), 121, int(os.environ.get('FUZZ_STRIDE', '1')))
def _scene_odd(seed, device):
    """Corners of the interface: triangles that cross the camera's near plane or lie behind it, a viewport on a perspective
    camera, lights that are not directly visible, a rotated environment map that may be invisible to the camera."""
    rng = np.random.RandomState(7000 + seed)
    sc = _scene(seed, device)
    # a big triangle through the camera plane (z of the camera is about -5)
    v = np.array([[rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.0, 1.0)],
                  [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-1.0, 1.0)],
                  [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-6.5, -4.0)]], np.float32)
    shapes = list(sc.shapes)
    shapes.insert(3, Shape(torch.tensor(v, device=device, requires_grad=True), torch.tensor([[0, 1, 2]], dtype=torch.int32, device=device),
                           int(rng.randint(0, 3))))
    lights = []
    for l in sc.area_lights:
        lights.append(AreaLight(l.shape_id + 1, l.intensity, two_sided=l.two_sided, directly_visible=bool(rng.rand() < 0.5)))
    env = None
    if rng.rand() < 0.5:
        img = torch.tensor(rng.uniform(0.05, 1.0, (4, 8, 3)).astype(np.float32))
        ang = rng.uniform(0, 2 * np.pi)
        e2w = torch.tensor([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0], [-np.sin(ang), 0, np.cos(ang), 0], [0, 0, 0, 1]], dtype=torch.float32)
        env = EnvironmentMap(Texture([l.to(device).requires_grad_(True) for l in scenes._mip_chain(img)]), env_to_world=e2w,
                             directly_visible=bool(rng.rand() < 0.6))
    cam = sc.camera
    vp = (3, 2, 19, 20) if rng.rand() < 0.5 else None
    cam = Camera(position=cam.position, look_at=cam.look_at, up=cam.up, intrinsic_mat=cam.intrinsic_mat, clip_near=float(rng.choice([1e-2, 0.5, 2.0])),
                 resolution=cam.resolution, viewport=vp)
    return Scene(cam, shapes, sc.materials, lights, envmap=env)


def _render_odd(backend, seed, spp, mb, stripe=None):
    dev = torch.device('cpu')
    sc = _scene_odd(seed, dev)
    sampler = backend.SamplerType.independent if seed % 3 == 0 else backend.SamplerType.sobol
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=sampler, device=_sdev(backend), backend=backend)
    img = RenderFunction.apply(seed, *args)
    h, w, _ = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    up = torch.stack([1.0 + 0.5 * torch.sin(0.4 * xx + 0.2 * yy), 1.0 + 0.5 * torch.cos(0.3 * yy), 1.0 - 0.3 * torch.sin(0.2 * (xx + yy))], 2)
    if stripe is not None:
        keep = torch.zeros(h * w)
        keep[stripe[0]::stripe[1]] = 1
        up = up * keep.reshape(h, w, 1)
    (img * up.to(img.device)).sum().backward()
    out = {'image': _np(img)}
    for i, s in enumerate(sc.shapes):
        if s.vertices.grad is not None:
            out['shape%d' % i] = s.vertices.grad.numpy()
    for i, m in enumerate(sc.materials):
        t = m.diffuse_reflectance.mipmap[0]
        if t.grad is not None:
            out['mat%d_diffuse' % i] = t.grad.numpy()
    for i, l in enumerate(sc.area_lights):
        out['light%d' % i] = l.intensity.grad.numpy()
    if sc.envmap is not None:
        for lv, l in enumerate(sc.envmap.values.mipmap):
            if l.grad is not None:
                out['envmap_L%d' % lv] = l.grad.numpy()
    for n in ('position', 'look_at', 'up'):
        out['cam_' + n] = getattr(sc.camera, n).grad.numpy()
    return out


ODD_SEEDS = range(1 + int(os.environ.get('FUZZ_OFFSET', '0')), 121, int(os.environ.get('FUZZ_STRIDE', '1')))
def _blob(rng, subdiv):
    """A deformed icosphere with shared vertices (hundreds of silhouette / crease candidates for the edge hierarchies)."""
    t = (1 + 5 ** 0.5) / 2
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.array(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        mid, nf = {}, []
        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    v = np.asarray(v)
    k = rng.uniform(1.0, 3.0, 3)
    r = 1.0 + 0.25 * np.sin(k[0] * v[:, 0] * 3) * np.cos(k[1] * v[:, 1] * 3) + 0.1 * np.sin(k[2] * v[:, 2] * 5)
    return (v * r[:, None]).astype(np.float32), np.asarray(f, np.int32)


def _render_blob(backend, seed, stripe=None):
    dev = torch.device('cpu')
    rng = np.random.RandomState(3000 + seed)
    v, f = _blob(rng, 1 + seed % 2)
    v = v * float(rng.uniform(0.8, 1.2)) + rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    glossy = seed % 2 == 0
    mats = [Material(diffuse_reflectance=torch.tensor(rng.uniform(0.2, 0.8, 3).astype(np.float32), requires_grad=True),
                     specular_reflectance=torch.tensor((rng.uniform(0.1, 0.4, 3) if glossy else np.zeros(3)).astype(np.float32)),
                     roughness=torch.tensor([float(rng.uniform(0.1, 0.5)) if glossy else 1.0])),
            Material(diffuse_reflectance=torch.tensor([0.5, 0.5, 0.5], requires_grad=True)),
            Material(diffuse_reflectance=torch.zeros(3))]
    blob = Shape(torch.tensor(v, requires_grad=True), torch.tensor(f), 0)
    floor = Shape(torch.tensor([[-3.0, -1.6, -3.0], [3.0, -1.6, -3.0], [-3.0, -1.6, 3.0], [3.0, -1.6, 3.0]], requires_grad=True),
                  torch.tensor([[0, 2, 1], [1, 2, 3]], dtype=torch.int32), 1)
    c = rng.uniform([-2.0, 2.5, -3.0], [2.0, 4.0, 0.0])
    light = Shape(torch.tensor([[c[0] - 1, c[1], c[2] - 1], [c[0] + 1, c[1], c[2] - 1], [c[0] - 1, c[1], c[2] + 1], [c[0] + 1, c[1], c[2] + 1]],
                               dtype=torch.float32), torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32), 2)
    cam = Camera(position=torch.tensor([0.0, 0.8, -5.0], requires_grad=True), look_at=torch.tensor([0.0, 0.0, 0.0]),
                 up=torch.tensor([0.0, 1.0, 0.0]), fov=torch.tensor([45.0]), clip_near=1e-2, resolution=(32, 32))
    sc = Scene(cam, [blob, floor, light], mats, [AreaLight(2, torch.tensor([25.0, 25.0, 25.0], requires_grad=True), two_sided=True)])
    args = RenderFunction.serialize_scene(sc, 2, 2 + seed % 2, sampler_type=backend.SamplerType.sobol, device=_sdev(backend), backend=backend)
    img = RenderFunction.apply(seed, *args)
    h, w, _ = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    up = torch.stack([1.0 + 0.5 * torch.sin(0.4 * xx + 0.2 * yy), 1.0 + 0.5 * torch.cos(0.3 * yy), 1.0 - 0.3 * torch.sin(0.2 * (xx + yy))], 2)
    if stripe is not None:
        keep = torch.zeros(h * w)
        keep[stripe[0]::stripe[1]] = 1
        up = up * keep.reshape(h, w, 1)
    (img * up.to(img.device)).sum().backward()
    return {'image': _np(img), 'blob': blob.vertices.grad.numpy(), 'floor': floor.vertices.grad.numpy(),
            'mat0': mats[0].diffuse_reflectance.mipmap[0].grad.numpy(), 'mat1': mats[1].diffuse_reflectance.mipmap[0].grad.numpy(),
            'light0': sc.area_lights[0].intensity.grad.numpy(), 'cam_position': cam.position.grad.numpy()}


STRIDE = int(os.environ.get('FUZZ_STRIDE', '1'))       # the variant runs below take every third scene
FIRST = 1 + int(os.environ.get('FUZZ_OFFSET', '0'))     # ... the GPU legs every eighth, each from another first seed
PLAIN_SEEDS, RICH_SEEDS = range(FIRST, 201, STRIDE), range(FIRST, 161, STRIDE)


def _main(lib):
    """lib = path of the CPU harness build, or 'gpu': the product library on cuda:0 (the GPU leg)."""
    global MINE_DEVICE, ON_GPU
    from redner_amd import _capi
    if lib == 'gpu':
        assert torch.cuda.is_available(), 'the GPU leg needs a GPU'
        _capi.load()
        assert _capi.is_product_library(), _capi.library_path()
        MINE_DEVICE, ON_GPU = torch.device('cuda:0'), True
    else:
        _capi.load(lib)
    from redner_amd import redner
    oracle = oracle_util.load_oracle()
    failures = {}
    for seed in PLAIN_SEEDS:
        spp, mb = 2 + seed % 3, 1 + seed % 3
        ref, mine = _render(oracle, seed, spp, mb), _render(redner, seed, spp, mb)
        SCENE[0] = 'plain %d' % seed
        bad = _compare(mine, ref, lambda st: _render(oracle, seed, spp, mb, st))
        if bad is None and not ref['image'].any():
            bad = 'black image'
        if bad:
            failures['plain %d' % seed] = bad
    for seed in RICH_SEEDS:
        spp, mb, pc = 2 + seed % 4, 1 + seed % 5, seed % 5 == 0
        ref, mine = _render_rich(oracle, seed, spp, mb, pc), _render_rich(redner, seed, spp, mb, pc)
        SCENE[0] = 'rich %d' % seed
        bad = _compare(mine, ref, lambda st: _render_rich(oracle, seed, spp, mb, pc, st))
        if bad:
            failures['rich %d' % seed] = bad
    for seed in MESH_SEEDS:
        spp, mb = 2 + seed % 3, seed % 4
        ref, mine = _render_mesh(oracle, seed, spp, mb), _render_mesh(redner, seed, spp, mb)
        SCENE[0] = 'mesh %d' % seed
        bad = _compare(mine, ref, lambda st: _render_mesh(oracle, seed, spp, mb, st))
        if bad:
            failures['mesh %d' % seed] = bad
    for seed in ODD_SEEDS:
        spp, mb = 1 + seed % 6, seed % 7
        ref, mine = _render_odd(oracle, seed, spp, mb), _render_odd(redner, seed, spp, mb)
        SCENE[0] = 'odd %d' % seed
        bad = _compare(mine, ref, lambda st: _render_odd(oracle, seed, spp, mb, st))
        if bad:
            failures['odd %d' % seed] = bad
    for seed in range(FIRST, 25, STRIDE):             # 80- and 320-triangle blobs above a floor
        ref, mine = _render_blob(oracle, seed), _render_blob(redner, seed)
        SCENE[0] = 'blob %d' % seed
        bad = _compare(mine, ref, lambda st: _render_blob(oracle, seed, st))
        if bad:
            failures['blob %d' % seed] = bad
    # an optimisation loop: the same connectivity with moved vertices (hierarchies refitted, edge structures rebuilt), then only
    # materials / lights changed (edge structures shared with the previous Scene), then the camera moved -- every step against
    # the oracle, which builds everything from scratch each time
    for seed in range(FIRST, 21, STRIDE):
        for step in range(6):
            def loop_step(backend):
                sc = _scene_mesh(seed, torch.device('cpu'))
                rng = np.random.RandomState(100 * seed + step)
                with torch.no_grad():
                    if step in (1, 2, 5):
                        for sh in sc.shapes[:-1]:
                            sh.vertices += torch.tensor(rng.normal(0, 0.03 * step, tuple(sh.vertices.shape)).astype(np.float32))
                    if step in (3, 5):
                        for m in sc.materials[:3]:
                            m.diffuse_reflectance.mipmap[0].mul_(float(rng.uniform(0.5, 1.2)))
                        sc.area_lights[0].intensity.mul_(float(rng.uniform(0.5, 1.5)))
                    if step in (4, 5):
                        sc.camera.position += torch.tensor(rng.normal(0, 0.1, 3).astype(np.float32))
                args = RenderFunction.serialize_scene(sc, 2, 2, sampler_type=backend.SamplerType.sobol, device=_sdev(backend), backend=backend)
                img = RenderFunction.apply(seed, *args)
                img.sum().backward()
                o = {'image': _np(img), 'cam_position': sc.camera.position.grad.numpy(), 'light0': sc.area_lights[0].intensity.grad.numpy()}
                for i, sh in enumerate(sc.shapes[:-1]):
                    o['shape%d' % i] = sh.vertices.grad.numpy()
                return o
            ref, mine = loop_step(oracle), loop_step(redner)
            SCENE[0] = 'loop %d step %d' % (seed, step)
            bad = _distance(mine, ref)
            if bad and ON_GPU:                       # (see _compare: is the oracle's own render reproducible?)
                ref2 = loop_step(oracle)
                if _distance(ref2, ref) is not None:
                    ORACLE_UNSTABLE.append(SCENE[0])
                    bad = _distance(mine, ref2)
            if bad:
                failures['loop %d step %d' % (seed, step)] = bad
    # degenerate geometry: a triangle with two coinciding corners, with collinear corners, and two identical triangles
    # (coplanar, overlapping: the closest-hit tie rule and the edge list's merging of duplicate edges)
    for seed in range(FIRST, 61, STRIDE):
        outs = []
        for backend in (oracle, redner):
            rng = np.random.RandomState(400 + seed)
            sc = _scene(seed, torch.device('cpu'))
            with torch.no_grad():
                v = sc.shapes[0].vertices
                kind = int(rng.randint(0, 3))
                if kind == 0:
                    v[1] = v[0]
                elif kind == 1:
                    v[2] = 0.5 * (v[0] + v[1])
                elif v.shape[0] >= 6:
                    v[3:6] = v[0:3]
            args = RenderFunction.serialize_scene(sc, 3, 2, sampler_type=backend.SamplerType.sobol, device=_sdev(backend), backend=backend)
            img = RenderFunction.apply(seed, *args)
            img.sum().backward()
            o = {'image': _np(img), 'cam_position': sc.camera.position.grad.numpy()}
            for i, sh in enumerate(sc.shapes[:3]):
                o['shape%d' % i] = sh.vertices.grad.numpy()
            outs.append(o)
        ref, mine = outs
        if any(not np.isfinite(x).all() for x in ref.values()):
            if any(not np.array_equal(np.isfinite(ref[k]), np.isfinite(mine[k])) for k in ref):
                failures['degenerate %d' % seed] = 'non-finite values in other places than the oracle'
            continue
        SCENE[0] = 'degenerate %d' % seed
        bad = _distance(mine, ref)
        if bad:
            failures['degenerate %d' % seed] = bad
    # sample blocks (what the ranks of a multi-GPU job render, redner_amd/distributed.py): 2 or 3 blocks of 2 samples each,
    # summed in block order, against the oracle's single call over all samples.  Without mip-mapped textures: with them the
    # reference's primary-edge pass reads ray differentials that EARLIER SAMPLES left in its scratch (DESIGN.md section 1,
    # "stale scratch"), a chain that a block starting at sample k > 0 cannot continue -- its first samples pick other texture
    # levels for a few edge samples than the single call does (another draw of the same estimator; measured here: vertex
    # gradients 0.5-2 % apart on 20 x 22 frames with a normal map and 4-6 samples, nothing without mip levels).
    from redner_amd.distributed import render_blocked
    for seed in range(FIRST, 41, STRIDE):
        blocks = 2 + seed % 2
        outs = []
        for backend, R in ((oracle, None), (redner, blocks)):
            sc = _scene_mesh(seed, torch.device('cpu')) if seed % 2 else _scene(seed, torch.device('cpu'))
            for m in sc.materials:
                m.normal_map = m.generic_texture = None
            args = RenderFunction.serialize_scene(sc, 2 * blocks, 1 + seed % 2, sampler_type=backend.SamplerType.sobol,
                                                  device=_sdev(backend), backend=backend)
            img = render_blocked(seed, args, R) if R else RenderFunction.apply(seed, *args)
            img.sum().backward()
            o = {'image': _np(img), 'cam_position': sc.camera.position.grad.numpy()}
            for i, sh in enumerate(sc.shapes):
                if sh.vertices.grad is not None:
                    o['shape%d' % i] = sh.vertices.grad.numpy()
            outs.append(o)
        ref, mine = outs
        n = np.linalg.norm(ref['image'].astype(np.float64))
        if np.linalg.norm(mine['image'].astype(np.float64) - ref['image']) > 1e-6 * n:       # fp32 sums in another order
            failures['blocks %d' % seed] = 'image'
        mine['image'] = ref['image']
        SCENE[0] = 'blocks %d' % seed
        bad = _distance(mine, ref)
        if bad:
            failures['blocks %d' % seed] = bad
    # screen-space gradient images (RenderFunction.visualize_screen_gradient, tests/test_screen_gradient.py)
    for seed in range(FIRST, 41, STRIDE):
        kw = dict(num_samples=2 + seed % 3, max_bounces=seed % 3,
                  sampler_type=(oracle.SamplerType.independent if seed % 2 else oracle.SamplerType.sobol))
        imgs = []
        for backend in (oracle, redner):
            kw['sampler_type'] = backend.SamplerType.independent if seed % 2 else backend.SamplerType.sobol
            sc = _scene_mesh(seed, torch.device('cpu')) if seed % 4 == 0 else _scene(seed, torch.device('cpu'))
            ch = [backend.channels.radiance] if seed % 3 else [backend.channels.diffuse_reflectance]
            imgs.append(RenderFunction.visualize_screen_gradient(None, seed, sc, channels=ch, backend=backend, device=_sdev(backend),
                                                                 **kw).cpu().numpy())
        n = np.linalg.norm(imgs[0].astype(np.float64))
        d = np.linalg.norm(imgs[1].astype(np.float64) - imgs[0].astype(np.float64))
        if not d <= 1e-4 * n + 1e-9:
            failures['screen gradient %d' % seed] = '%.3e' % (d / max(n, 1e-300))
    print('FLIPS ' + json.dumps(FLIPS))
    print('ORACLE_UNSTABLE ' + json.dumps(ORACLE_UNSTABLE))
    print('SCENES %d' % len(SEEN))
    print('FUZZ ' + json.dumps(failures))


# the default schedule (sample batches, specialised stage kernels), one sample per launch, no stage specialisation, ragged
# batches (3 samples, then what is left), several batches per call (2 samples each under the lane cap)
@pytest.mark.parametrize('variant', ['', 'RDR_BATCH=1 FUZZ_STRIDE=3', 'RDR_FORCE_GENERAL=1 FUZZ_STRIDE=3', 'RDR_BATCH=3 FUZZ_STRIDE=3',
                                     'RDR_BATCH_LANES=1000 FUZZ_STRIDE=3',
                                     # the gather's hand-over paths (tiny budgets / list capacities); caches and refits off
                                     'RDR_GATHER_BUDGET=2 RDR_GATHER_CAPS=3,5 FUZZ_STRIDE=4', 'RDR_GATHER_BUDGET=1 RDR_GATHER_CAPS=0,0 FUZZ_STRIDE=4',
                                     'RDR_NO_REFIT=1 RDR_NO_EDGE_CACHE=1 FUZZ_STRIDE=4'])
def test_random_scenes_hostsim_vs_oracle(hostsim_backend, variant):
    from conftest import HOSTSIM_LIB
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MALLOC_MMAP_THRESHOLD_='1024', MALLOC_PERTURB_='255',
               PYTHONPATH=os.pathsep.join([os.path.dirname(here), here, os.environ.get('PYTHONPATH', '')]))
    for k in ('RDR_BATCH', 'RDR_FORCE_GENERAL', 'FUZZ_STRIDE', 'RDR_BATCH_LANES', 'RDR_GATHER_BUDGET', 'RDR_GATHER_CAPS', 'RDR_NO_REFIT',
              'RDR_NO_EDGE_CACHE'):
        env.pop(k, None)
    env.update(dict(kv.split('=') for kv in variant.split()))
    out = subprocess.check_output([sys.executable, os.path.abspath(__file__), HOSTSIM_LIB], env=env, timeout=1500).decode()
    line = [l for l in out.splitlines() if l.startswith('FUZZ ')][-1]
    assert json.loads(line[5:]) == {}


# The GPU leg: the product library on cuda:0 against the live oracle (oracle/_ref travels to the GPU box), every eighth scene of
# every family above, under the default schedule (sample batches, specialised kernels, side streams), with ragged batches, and
# with the refilling traversal kernel forced onto every queue (it normally serves queues of >= 2^22 lanes only: bench.py's).
GPU_KEYS = ('RDR_BATCH', 'RDR_FORCE_GENERAL', 'FUZZ_STRIDE', 'FUZZ_OFFSET', 'RDR_BATCH_LANES', 'RDR_TRACE_REFILL_ALL', 'RDR_NO_OVERLAP',
            'RDR_TRACE_REFILL', 'RDR_WORKERS')


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['FUZZ_STRIDE=8', 'RDR_BATCH=3 FUZZ_STRIDE=8 FUZZ_OFFSET=3',
                                     'RDR_TRACE_REFILL_ALL=1 FUZZ_STRIDE=8 FUZZ_OFFSET=5'])
def test_random_scenes_gpu_vs_oracle(gpu_backend, variant):
    here = os.path.dirname(os.path.abspath(__file__))
    # MALLOC_PERTURB_=255: glibc then fills every chunk it hands out with 0x00 (perturb byte ^ 0xff) -- the scratch the
    # reference reads without having written it is zero whether the chunk is a fresh mapping or recycled heap (see _compare)
    env = dict(os.environ, MALLOC_MMAP_THRESHOLD_='1024', MALLOC_PERTURB_='255',
               PYTHONPATH=os.pathsep.join([os.path.dirname(here), here, os.environ.get('PYTHONPATH', '')]))
    for k in GPU_KEYS:
        env.pop(k, None)
    env.update(dict(kv.split('=') for kv in variant.split()))
    out = subprocess.check_output([sys.executable, os.path.abspath(__file__), 'gpu'], env=env, timeout=900).decode()
    grab = lambda tag: [l for l in out.splitlines() if l.startswith(tag + ' ')][-1][len(tag) + 1:]
    flips, scenes, unstable = json.loads(grab('FLIPS')), int(grab('SCENES')), json.loads(grab('ORACLE_UNSTABLE'))
    path = os.environ.get('RDR_PARITY_REPORT')
    if path:
        with open(path, 'a') as f:
            f.write(json.dumps({'case': 'fuzz_gpu_vs_live_oracle [%s]' % variant, 'backend': 'gpu', 'scenes': scenes, 'edge_flips': flips,
                                'oracle_not_reproducible': unstable}) + '\n')
    assert json.loads(grab('FUZZ')) == {}, (grab('FUZZ')[:3000], 'oracle not reproducible: %s' % unstable, 'flips: %s' % flips)
    assert scenes >= 60
    # Rounds 1-3 allowed two scenes per leg in which a single edge sample landed on another edge (_edge_flip: the device's
    # sin / cos / pow were not glibc's).  In the exact build they are (csrc/libm_exact.h) and none is allowed; the default build
    # -- the device's own libm -- gets the old budget, stated in parity_util.DEFAULT_BUILD_FUZZ_FLIPS_PER_LEG (first full run of
    # the three legs on it, round 6: one scene in one leg, `mesh 118`, two rows each in two shapes).
    from parity_util import DEFAULT_BUILD_FUZZ_FLIPS_PER_LEG, libm_exact
    allowed = 0 if libm_exact() else DEFAULT_BUILD_FUZZ_FLIPS_PER_LEG
    assert len(flips) <= allowed, flips


if __name__ == '__main__':
    _main(sys.argv[1])
