"""Test scenes, defined once and buildable against any `redner`-API backend.

Parameters follow the reference's own test scripts (cited per scene) so the parity tests read
like them; geometry that needs a mesh file comes from tests/golden/*.npz (see make_golden.py).
"""
import os
import numpy as np
import torch

from redner_amd.render_pytorch import Camera, Shape, Material, AreaLight, Scene, Texture

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _t(x, device, dtype=torch.float32, grad=False):
    t = torch.tensor(x, dtype=dtype, device=device)
    if grad:
        t.requires_grad_(True)
    return t


def single_triangle(device, resolution=(64, 64)):
    """tests/test_single_triangle.py:17-81, with the perturbed vertices of :128-131 as the
    differentiable input (BASELINE config 1)."""
    cam = Camera(position=_t([0.0, 0.0, -5.0], 'cpu'), look_at=_t([0.0, 0.0, 0.0], 'cpu'),
                 up=_t([0.0, 1.0, 0.0], 'cpu'), fov=_t([45.0], 'cpu'), clip_near=1e-2, resolution=resolution)
    mats = [Material(diffuse_reflectance=_t([0.5, 0.5, 0.5], device))]
    tri = Shape(_t([[-2.0, 1.5, 0.3], [0.9, 1.2, -0.3], [-0.4, -1.4, 0.2]], device, grad=True),
                _t([[0, 1, 2]], device, torch.int32), 0)
    light = Shape(_t([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]], device),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 0)
    return Scene(cam, [tri, light], mats, [AreaLight(1, _t([20.0, 20.0, 20.0], 'cpu'))])


def two_triangles(device, resolution=(256, 256)):
    """tests/test_two_triangles.py:11-55, perturbed vertices of :72-79 (BASELINE config 2)."""
    cam = Camera(position=_t([0.0, 0.0, -5.0], 'cpu'), look_at=_t([0.0, 0.0, 0.0], 'cpu'),
                 up=_t([0.0, 1.0, 0.0], 'cpu'), fov=_t([45.0], 'cpu'), clip_near=1e-2, resolution=resolution)
    mats = [Material(diffuse_reflectance=_t(c, device)) for c in
            ([0.35, 0.75, 0.35], [0.75, 0.35, 0.35], [0.0, 0.0, 0.0])]
    t0 = Shape(_t([[-1.3, 1.5, 0.1], [1.5, 0.7, -0.2], [-0.8, -1.1, 0.2]], device, grad=True),
               _t([[0, 1, 2]], device, torch.int32), 0)
    t1 = Shape(_t([[-0.5, 1.2, 1.2], [0.3, 1.7, 1.0], [0.5, -1.8, 1.3]], device, grad=True),
               _t([[0, 1, 2]], device, torch.int32), 1)
    light = Shape(_t([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]], device),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 2)
    return Scene(cam, [t0, t1, light], mats, [AreaLight(2, _t([20.0, 20.0, 20.0], 'cpu'))])


def triangle_through_near_plane(device, resolution=(64, 64)):
    """tests/test_single_triangle_clipped.py:12-41: a triangle one corner of which lies BEHIND the camera, so the primary
    edge sampler has to clip its edges against the near plane (src/edge.cpp: project / clip), with the light behind the
    camera as there.  The triangle's vertices and the camera pose carry gradients."""
    cam = Camera(position=_t([0.0, 0.0, -5.0], 'cpu', grad=True), look_at=_t([0.0, 0.0, 0.0], 'cpu', grad=True),
                 up=_t([0.0, 1.0, 0.0], 'cpu', grad=True), fov=_t([45.0], 'cpu'), clip_near=1e-2, resolution=resolution)
    mats = [Material(diffuse_reflectance=_t([0.55, 0.5, 0.45], device, grad=True))]
    tri = Shape(_t([[-1.2, 0.9, 0.2], [1.1, 1.1, -0.2], [-0.4, -1.9, -6.6]], device, grad=True),
                _t([[0, 1, 2]], device, torch.int32), 0)
    light = Shape(_t([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]], device),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 0)
    return Scene(cam, [tri, light], mats, [AreaLight(1, _t([20.0, 20.0, 20.0], 'cpu', grad=True))])


def glossy_floor_blocker(device, resolution=(48, 48)):
    """tests/test_shadow_glossy.py:11-60 (and the other test_shadow_* scripts): a floor, a blocker between it and an area
    light, the camera looking down at the floor.  The floor is a near-mirror (roughness 5e-4, no diffuse part), so what
    the camera sees is the reflection of light and blocker: BSDF-sampled specular paths, the min_roughness rule of the
    secondary edge sampler (src/edge.cpp:1396-1401) and nearly singular pdfs.  Gradients: blocker and floor vertices,
    the floor's specular reflectance and roughness, the light, the camera position."""
    cam = Camera(position=_t([0.0, 2.0, -4.0], 'cpu', grad=True), look_at=_t([0.0, -2.0, 0.0], 'cpu'),
                 up=_t([0.0, 1.0, 0.0], 'cpu'), fov=_t([45.0], 'cpu'), clip_near=1e-2, resolution=resolution)
    mats = [Material(diffuse_reflectance=_t([0.0, 0.0, 0.0], device), specular_reflectance=_t([0.9, 0.95, 1.0], device, grad=True),
                     roughness=_t([0.0005], device, grad=True)),
            Material(diffuse_reflectance=_t([0.5, 0.45, 0.55], device, grad=True)),
            Material(diffuse_reflectance=_t([0.0, 0.0, 0.0], device))]
    floor = Shape(_t([[-4.0, 0.0, -4.0], [-4.0, 0.0, 4.0], [4.0, 0.0, -4.0], [4.0, 0.0, 4.0]], device, grad=True),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 0)
    blocker = Shape(_t([[0.1, 5.0, 0.0], [-0.4, 7.1, 2.0], [1.4, 5.2, -0.1], [1.1, 7.0, 2.1]], device, grad=True),
                    _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 1)
    light = Shape(_t([[-2.0, 7.0, 4.0], [-2.0, 11.0, 4.0], [2.0, 7.0, 4.0], [2.0, 11.0, 4.0]], device),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 2)
    return Scene(cam, [floor, blocker, light], mats, [AreaLight(2, _t([0.6, 0.55, 0.5], 'cpu', grad=True))])


def triangle_soup_large(device, resolution=(32, 32)):
    """The same at 40x the size (60 k triangles, 180 k edges): the edge-structure kernels beyond one workgroup's worth of anything."""
    return triangle_soup(device, resolution, scale=40)


def triangle_soup(device, resolution=(32, 32), scale=1):
    """Stress input for the edge-structure build (tests/test_edge_build.py): 1 500 seeded triangles in four shapes -- a cloud of
    free triangles, a coarse-grid cloud (vertex coordinates snapped to 1/8: many edges share their Morton code, so the order of
    equal codes and the id tie-break of the radix tree decide the topology), a strip mesh with shared vertices, and a stack of
    exact duplicates (identical bounds, identical codes).  Not a BASELINE configuration."""
    g = torch.Generator().manual_seed(4711)
    cam = Camera(position=_t([0.0, 0.0, -6.0], 'cpu'), look_at=_t([0.0, 0.0, 0.0], 'cpu'),
                 up=_t([0.0, 1.0, 0.0], 'cpu'), fov=_t([45.0], 'cpu'), clip_near=1e-2, resolution=resolution)
    mats = [Material(diffuse_reflectance=_t([0.5, 0.5, 0.5], device)), Material(diffuse_reflectance=_t([0.0, 0.0, 0.0], device))]

    def free(n, snap):
        c = (torch.rand(n, 1, 3, generator=g) - 0.5) * torch.tensor([4.0, 4.0, 2.0])
        v = c + (torch.rand(n, 3, 3, generator=g) - 0.5) * 0.6
        if snap:
            v = torch.round(v * 8.0) / 8.0
            v = v + torch.tensor([[[0.0, 0.0, 0.0], [0.125, 0.0, 0.0], [0.0, 0.125, 0.0]]]) * (torch.rand(n, 1, 1, generator=g) < 0.3)   # some flat ones
        return v.reshape(-1, 3).contiguous(), torch.arange(3 * n, dtype=torch.int32).reshape(n, 3)

    v0, i0 = free(600 * scale, False)
    v1, i1 = free(500 * scale, True)
    # strip: 2 x 101 vertices, 200 triangles
    xs = torch.linspace(-2.0, 2.0, 101)
    top = torch.stack([xs, 0.3 + 0.2 * torch.sin(3 * xs), 1.5 + 0.1 * torch.cos(5 * xs)], 1)
    bot = torch.stack([xs, -0.3 + 0.1 * torch.cos(2 * xs), 1.5 + 0.1 * torch.sin(4 * xs)], 1)
    v2 = torch.cat([top, bot], 0).contiguous()
    i2 = torch.tensor([[k, k + 1, 101 + k] for k in range(100)] + [[k + 1, 102 + k, 101 + k] for k in range(100)], dtype=torch.int32)
    # duplicates: the same triangle 200 times (separate vertices)
    t = torch.tensor([[-0.5, -0.5, -1.0], [0.5, -0.4, -1.1], [0.1, 0.6, -0.9]])
    v3 = t.repeat(200, 1).contiguous()            # (not scaled: thousands of identical codes make the radix tree deeper than the 64-entry traversal stacks)
    i3 = torch.arange(600, dtype=torch.int32).reshape(200, 3)
    shapes = [Shape(v.to(device).requires_grad_(True), i.to(device), 0) for v, i in ((v0, i0), (v1, i1), (v2, i2), (v3, i3))]
    light = Shape(_t([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]], device),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 1)
    return Scene(cam, shapes + [light], mats, [AreaLight(4, _t([20.0, 20.0, 20.0], 'cpu'))])


def bunny_box(device, resolution=(512, 512), vertex_grad=True):
    """tests/scenes/bunny_box.xml as loaded by pyredner.load_mitsuba (tests/test_bunny_box.py):
    Stanford bunny in a Cornell box, 7 shapes / 14 416 triangles, one area light.  The mesh
    arrays were exported once by tests/golden/make_golden.py into bunny_box_scene.npz."""
    z = np.load(os.path.join(GOLDEN, 'bunny_box_scene.npz'))
    n_shapes, n_mats = int(z['num_shapes']), int(z['num_materials'])
    if 'cam_to_world' in z.files:
        cam = Camera(cam_to_world=_t(z['cam_to_world'], 'cpu'), intrinsic_mat=_t(z['intrinsic_mat'], 'cpu'),
                     clip_near=float(z['clip_near']), resolution=resolution)
    else:
        cam = Camera(position=_t(z['cam_position'], 'cpu'), look_at=_t(z['cam_look_at'], 'cpu'),
                     up=_t(z['cam_up'], 'cpu'), intrinsic_mat=_t(z['intrinsic_mat'], 'cpu'),
                     clip_near=float(z['clip_near']), resolution=resolution)
    shapes = []
    for i in range(n_shapes):
        def get(name, dtype=torch.float32):
            key = 'shape%d_%s' % (i, name)
            return _t(z[key], device, dtype) if key in z.files else None
        sh = Shape(get('vertices'), get('indices', torch.int32), int(z['shape%d_material_id' % i]),
                   uvs=get('uvs'), normals=get('normals'))
        shapes.append(sh)
    mats = []
    for i in range(n_mats):
        m = Material(diffuse_reflectance=_t(z['mat%d_diffuse' % i], device),
                     specular_reflectance=_t(z['mat%d_specular' % i], device),
                     roughness=_t(z['mat%d_roughness' % i], device),
                     two_sided=bool(z['mat%d_two_sided' % i]))
        m.compute_specular_lighting = bool(z['mat%d_compute_specular_lighting' % i])
        mats.append(m)
    lights = [AreaLight(int(z['light0_shape_id']), _t(z['light0_intensity'], 'cpu'),
                        two_sided=bool(z['light0_two_sided']))]
    if vertex_grad:
        shapes[int(z['bunny_shape_id'])].vertices.requires_grad_(True)
    return Scene(cam, shapes, mats, lights)


def _subdivide(v, f):
    """One level of midpoint subdivision with shared vertices: every triangle -> 4 (v [N, 3] float32, f [T, 3] int64)."""
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    uniq, inv = torch.unique(torch.sort(e, dim=1).values, dim=0, return_inverse=True)
    mid = 0.5 * (v[uniq[:, 0]] + v[uniq[:, 1]])
    m = inv.reshape(3, -1).t() + v.shape[0]
    a, b, c, ab, bc, ca = f[:, 0], f[:, 1], f[:, 2], m[:, 0], m[:, 1], m[:, 2]
    f2 = torch.cat([torch.stack([a, ab, ca], 1), torch.stack([ab, b, bc], 1), torch.stack([ca, bc, c], 1), torch.stack([ab, bc, ca], 1)], 0)
    return torch.cat([v, mid], 0), f2


def bunny_box_subdivided(device, resolution=(1024, 1024), levels=4):
    """bunny_box with the bunny tessellated `levels` times further (x 4 per level: 3.7 M triangles at 4) and its surface rippled
    by a few 1e-4 so that the small triangles are not coplanar: same picture, a triangle hierarchy of ~250 MB instead of 1 MB --
    far beyond the 4 MiB of L2 per XCD.  The workload of bench.py's `roofline_large`; not a BASELINE
    configuration and not a parity case (no oracle fixture: the oracle's scalar builder would need hours)."""
    sc = bunny_box(device, resolution, vertex_grad=False)
    z = np.load(os.path.join(GOLDEN, 'bunny_box_scene.npz'))
    sh = sc.shapes[int(z['bunny_shape_id'])]
    v, f = sh.vertices.detach(), sh.indices.long()          # (on `device`: seconds on the GPU, half a minute on the host)
    for _ in range(levels):
        v, f = _subdivide(v, f)
    if levels > 0:
        v = v + 2e-4 * torch.sin(900.0 * v[:, [1, 2, 0]]) * torch.cos(700.0 * v[:, [2, 0, 1]])
    sh.vertices = v.contiguous().to(device)
    sh.indices = f.to(torch.int32).contiguous().to(device)
    sh.uvs = None
    sh.normals = None
    return sc


def bunny_box_tile(device, resolution=(512, 512)):
    """BASELINE config 3 at full resolution and full spp, restricted by the camera viewport to the 128 x 128 tile
    (rows 256..384, columns 128..256) that holds the bunny's ears, head and back against the back wall
    ("full-resolution tile via viewport")."""
    sc = bunny_box(device, resolution)
    sc.camera.viewport = (256, 128, 384, 256)
    return sc


def _mip_chain(base):
    """Box-filtered mip levels down to 1x1 ([H, W, C] tensors); stands in for pyredner.Texture's
    own generation (pyredner/texture.py), which is host-side Python outside the hot path."""
    levels = [base.contiguous()]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        t = levels[-1].permute(2, 0, 1)[None]
        t = torch.nn.functional.avg_pool2d(t, 2, ceil_mode=True)
        levels.append(t[0].permute(1, 2, 0).contiguous())
    return levels


def _procedural(h, w, c, phase, lo=0.1, hi=0.9):
    yy, xx = np.meshgrid(np.arange(h) / h, np.arange(w) / w, indexing='ij')
    chans = [0.5 + 0.5 * np.sin(2 * np.pi * ((k + 1) * xx + (k + 2) * 0.5 * yy) + phase + 0.7 * k) for k in range(c)]
    return (lo + (hi - lo) * np.stack(chans, axis=2)).astype(np.float32)


def textured_sphere(device, resolution=(48, 48)):
    """G-buffer / mip-mapped texture scene in the spirit of tests/test_g_buffer.py and
    tests/test_texture.py: a UV sphere with smooth normals, uvs, vertex colours and mip-mapped
    diffuse / roughness / generic textures in front of a normal-mapped wall, one area light."""
    cam = Camera(position=_t([0.3, 0.2, -5.0], 'cpu'), look_at=_t([0.0, 0.0, 0.0], 'cpu'),
                 up=_t([0.0, 1.0, 0.0], 'cpu'), fov=_t([45.0], 'cpu'), clip_near=1e-2, resolution=resolution)
    nu, nv, radius = 12, 8, 1.3
    verts, uvs, cols = [], [], []
    for j in range(nv + 1):
        th = np.pi * j / nv
        for i in range(nu + 1):
            ph = 2 * np.pi * i / nu
            verts.append([radius * np.sin(th) * np.cos(ph), radius * np.cos(th), radius * np.sin(th) * np.sin(ph)])
            uvs.append([i / nu, j / nv])
            cols.append([0.5 + 0.4 * np.cos(ph), 0.5 + 0.4 * np.sin(th), 0.5 + 0.4 * np.sin(ph + th)])
    tris = []
    for j in range(nv):
        for i in range(nu):
            a, b = j * (nu + 1) + i, j * (nu + 1) + i + 1
            c, d = a + nu + 1, b + nu + 1
            if j > 0:
                tris.append([a, b, c])
            if j < nv - 1:
                tris.append([b, d, c])
    verts = np.asarray(verts, np.float32)
    normals = verts / np.linalg.norm(verts, axis=1, keepdims=True)
    sphere = Shape(_t(verts, device, grad=True), _t(tris, device, torch.int32), 0,
                   uvs=_t(np.asarray(uvs, np.float32), device, grad=True), normals=_t(normals, device, grad=True),
                   colors=_t(np.asarray(cols, np.float32), device, grad=True))
    wall = Shape(_t([[-3.0, -3.0, 2.5], [3.0, -3.0, 2.5], [-3.0, 3.0, 2.2], [3.0, 3.0, 2.2]], device, grad=True),
                 _t([[0, 2, 1], [1, 2, 3]], device, torch.int32), 1,
                 uvs=_t([[0.0, 0.0], [2.0, 0.0], [0.0, 2.0], [2.0, 2.0]], device))
    light = Shape(_t([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]], device),
                  _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 2)

    def mips(arr, grad=True):
        levels = _mip_chain(torch.from_numpy(arr))
        return [l.to(device).requires_grad_(grad) for l in levels]

    nm = _procedural(16, 16, 3, 0.3, 0.35, 0.65)
    nm[..., 2] = 0.9                                   # mostly +z tangent-space normals
    mats = [Material(diffuse_reflectance=Texture(mips(_procedural(32, 32, 3, 0.0)), uv_scale=_t([2.0, 1.0], device)),
                     specular_reflectance=_t([0.15, 0.2, 0.25], device, grad=True),
                     roughness=Texture(mips(_procedural(16, 16, 1, 1.0, 0.2, 0.7))),
                     generic_texture=Texture(mips(_procedural(8, 8, 5, 2.0)))),
            Material(diffuse_reflectance=_t([0.6, 0.55, 0.5], device, grad=True), normal_map=Texture(mips(nm))),
            Material(diffuse_reflectance=_t([0.0, 0.0, 0.0], device))]
    return Scene(cam, [sphere, wall, light], mats, [AreaLight(2, _t([25.0, 25.0, 25.0], 'cpu'))])


# ---- non-pinhole cameras and lens distortion (tests/test_camera_distortion.py, test_fisheye in the reference) ----
def two_triangles_ortho(device, resolution=(64, 64)):
    sc = two_triangles(device, resolution)
    c = sc.camera
    sc.camera = Camera(position=c.position, look_at=c.look_at, up=c.up, clip_near=c.clip_near, resolution=resolution,
                       intrinsic_mat=_t([[0.4, 0.0, 0.0], [0.0, 0.4, 0.0], [0.0, 0.0, 1.0]], 'cpu'), camera_type=1)
    return sc


def two_triangles_distorted(device, resolution=(64, 64)):
    sc = two_triangles(device, resolution)
    c = sc.camera
    sc.camera = Camera(position=c.position, look_at=c.look_at, up=c.up, fov=_t([45.0], 'cpu'), clip_near=c.clip_near,
                       resolution=resolution,
                       distortion_params=_t([0.10, -0.05, 0.01, 0.02, 0.01, -0.005, 0.01, -0.008], 'cpu', grad=True))
    return sc


def _bunny_box_inside(device, resolution, camera_type):
    sc = bunny_box(device, resolution)
    z = np.load(os.path.join(GOLDEN, 'bunny_box_scene.npz'))
    pos = np.array([0.1, 1.2, 0.8], np.float32)      # inside the box, so every direction sees geometry
    sc.camera = Camera(position=_t(pos, 'cpu'), look_at=_t([0.0, 0.6, 0.0], 'cpu'), up=_t([0.0, 1.0, 0.0], 'cpu'),
                       fov=_t([45.0], 'cpu'), clip_near=float(z['clip_near']), resolution=resolution,
                       camera_type=camera_type)
    return sc


def bunny_box_fisheye(device, resolution=(32, 32)):
    return _bunny_box_inside(device, resolution, 2)


def bunny_box_panorama(device, resolution=(32, 32)):
    return _bunny_box_inside(device, resolution, 3)


def envmap_sphere(device, resolution=(48, 48)):
    """Environment-lit scene in the spirit of tests/test_envmap.py / test_teapot_reflectance.py: a glossy
    UV sphere with smooth normals on a diffuse ground quad, lit only by a mip-mapped lat-long map."""
    from redner_amd.render_pytorch import EnvironmentMap
    base = textured_sphere(device, resolution)
    sphere = base.shapes[0]
    ground = Shape(_t([[-4.0, -1.4, -4.0], [4.0, -1.45, -4.0], [-4.0, -1.5, 4.0], [4.0, -1.35, 4.0]], device, grad=True),
                   _t([[0, 2, 1], [1, 2, 3]], device, torch.int32), 1)
    mats = [Material(diffuse_reflectance=_t([0.3, 0.25, 0.2], device, grad=True),
                     specular_reflectance=_t([0.4, 0.4, 0.45], device, grad=True),
                     roughness=_t([0.25], device, grad=True)),
            Material(diffuse_reflectance=_t([0.5, 0.5, 0.5], device, grad=True))]
    h, w = 16, 32
    yy, xx = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing='ij')
    sky = 0.3 + 0.7 * (1 - yy)
    sun = 30.0 * np.exp(-((xx - 0.3) ** 2 + (yy - 0.25) ** 2) / 0.004)
    img = np.stack([sky + sun, 0.9 * sky + 0.9 * sun, 1.2 * sky + 0.7 * sun], axis=2).astype(np.float32)
    levels = [l.to(device).requires_grad_(True) for l in _mip_chain(torch.from_numpy(img))]
    ang = 0.4
    e2w = _t([[np.cos(ang), 0.0, np.sin(ang), 0.0], [0.0, 1.0, 0.0, 0.0],
              [-np.sin(ang), 0.0, np.cos(ang), 0.0], [0.0, 0.0, 0.0, 1.0]], 'cpu', grad=True)
    env = EnvironmentMap(Texture(levels), env_to_world=e2w)
    return Scene(base.camera, [sphere, ground], mats, [], envmap=env)


def envmap_convex(device, resolution=(48, 48)):
    """The glossy sphere of envmap_sphere ALONE under the environment map: a convex object, so every path leaves the scene
    at its first bounce and the deeper path depths have no live lanes.  The reference's backward sweep skips such a depth
    before its edge sampler draws (src/pathtracer.cpp:432-436): the Sobol' dimensions of the shallower edge passes must not
    move (rendered with max_bounces 3)."""
    full = envmap_sphere(device, resolution)
    return Scene(full.camera, [full.shapes[0]], [full.materials[0]], [], envmap=full.envmap)


def misc_features(device, resolution=(40, 56), viewport=None):
    """Odds and ends of the interface in one scene: separate uv / normal index buffers, a two-sided light, a light
    that is not directly visible, two area lights (light CDF), a non-square image and an optional viewport
    (pyredner/shape.py uv_indices / normal_indices, pyredner/area_light.py, pyredner/camera.py viewport)."""
    cam = Camera(position=_t([0.0, 0.5, -6.0], 'cpu'), look_at=_t([0.0, 0.0, 0.0], 'cpu'),
                 up=_t([0.0, 1.0, 0.0], 'cpu'), fov=_t([50.0], 'cpu'), clip_near=1e-2, resolution=resolution,
                 viewport=viewport)
    # a pyramid whose uv / normal buffers are indexed separately from the positions
    verts = _t([[0.0, 1.2, 0.0], [-1.0, -0.6, -1.0], [1.0, -0.6, -1.0], [1.0, -0.6, 1.0], [-1.0, -0.6, 1.0]], device, grad=True)
    idx = _t([[0, 2, 1], [0, 3, 2], [0, 4, 3], [0, 1, 4]], device, torch.int32)
    uvs = _t([[0.5, 1.0], [0.0, 0.0], [1.0, 0.0], [0.25, 0.5], [0.75, 0.5], [0.5, 0.2]], device, grad=True)
    uv_idx = _t([[0, 2, 1], [3, 5, 4], [0, 1, 2], [4, 3, 5]], device, torch.int32)
    nrm = np.array([[0.0, 1.0, 0.0], [-0.7, 0.2, -0.7], [0.7, 0.2, -0.7], [0.7, 0.2, 0.7], [-0.7, 0.2, 0.7], [0.0, 0.4, -0.9],
                    [0.9, 0.4, 0.0]], np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    n_idx = _t([[0, 2, 1], [0, 6, 2], [0, 4, 3], [0, 1, 5]], device, torch.int32)
    pyramid = Shape(verts, idx, 0, uvs=uvs, normals=_t(nrm, device, grad=True), uv_indices=uv_idx, normal_indices=n_idx)
    floor = Shape(_t([[-4.0, -0.6, -4.0], [4.0, -0.6, -4.0], [-4.0, -0.6, 4.0], [4.0, -0.6, 4.0]], device, grad=True),
                  _t([[0, 2, 1], [1, 2, 3]], device, torch.int32), 1)
    light_a = Shape(_t([[-2.5, 2.5, -1.0], [-1.5, 2.5, -1.0], [-2.5, 2.5, 0.0], [-1.5, 2.5, 0.0]], device),
                    _t([[0, 1, 2], [1, 3, 2]], device, torch.int32), 2)
    light_b = Shape(_t([[1.5, 1.0, -3.0], [2.5, 1.0, -3.0], [1.5, 2.0, -3.2], [2.5, 2.0, -3.2]], device),
                    _t([[0, 2, 1], [1, 2, 3]], device, torch.int32), 2)
    tex = Texture(_mip_chain(torch.from_numpy(_procedural(16, 16, 3, 0.5))), uv_scale=_t([1.0, 1.0], device))
    for l in tex.mipmap:
        l.data = l.data.to(device)
    mats = [Material(diffuse_reflectance=Texture([l.to(device).requires_grad_(True) for l in tex.mipmap]),
                     specular_reflectance=_t([0.1, 0.1, 0.1], device, grad=True), roughness=_t([0.4], device, grad=True),
                     two_sided=True),
            Material(diffuse_reflectance=_t([0.45, 0.5, 0.4], device, grad=True)),
            Material(diffuse_reflectance=_t([0.0, 0.0, 0.0], device))]
    lights = [AreaLight(2, _t([30.0, 28.0, 25.0], 'cpu'), two_sided=True),
              AreaLight(3, _t([10.0, 14.0, 20.0], 'cpu'), two_sided=False, directly_visible=False)]
    return Scene(cam, [pyramid, floor, light_a, light_b], mats, lights)


def misc_features_viewport(device, resolution=(40, 56)):
    return misc_features(device, resolution, viewport=(6, 10, 30, 44))


def living_room_standin(device, resolution=(1024, 1024), envmap_variant=False):
    """BASELINE config 5 stand-in.  tests/test_living_room.py downloads its meshes and textures at
    run time and only tests/scenes/living-room-3-scene.xml is in the reference tree (area-lit, two-sided diffuse BSDFs
    with bitmap textures), so the real scene cannot be built offline.  Stand-in, labelled as such everywhere: the
    bunny_box geometry with every material two-sided and a seeded 512 x 512 procedural diffuse texture (8 x 8 box-blurred
    noise in [0.1, 0.8], mip-mapped, uv_scale (2, 2)) on the walls and the bunny, the same rectangle area light,
    max_bounces 6, gradients with respect to the camera pose (position / look_at / up), as the reference script optimises
    (tests/test_living_room.py:30-34,45-47,72-75).  envmap_variant=True adds what BASELINE.json's wording asks for: a
    256 x 512 environment map (vertical gradient + a 4 x 4-texel sun) and seeded specular / roughness textures."""
    sc = bunny_box(device, resolution, vertex_grad=False)
    z = np.load(os.path.join(GOLDEN, 'bunny_box_scene.npz'))
    gen = torch.Generator().manual_seed(20240924)

    def noise_tex(channels, lo, hi, size=512):
        t = torch.rand(size, size, channels, generator=gen)
        t = torch.nn.functional.avg_pool2d(t.permute(2, 0, 1)[None], 8, stride=1, padding=4)[0, :, :size, :size].permute(1, 2, 0)
        t = (t - t.min()) / (t.max() - t.min())
        return (lo + (hi - lo) * t).contiguous()

    light_mat = sc.shapes[int(z['light0_shape_id'])].material_id
    for i, m in enumerate(sc.materials):
        m.two_sided = True
        if i == light_mat:
            continue
        levels = [l.to(device) for l in _mip_chain(noise_tex(3, 0.1, 0.8))[:8]]      # the reference keeps at most 8 levels (src/texture.h:11)
        m.diffuse_reflectance = Texture(levels, uv_scale=_t([2.0, 2.0], device))
        if envmap_variant:
            m.specular_reflectance = Texture([l.to(device) for l in _mip_chain(noise_tex(3, 0.0, 0.3))[:8]], uv_scale=_t([2.0, 2.0], device))
            m.roughness = Texture([l.to(device) for l in _mip_chain(noise_tex(1, 0.05, 0.6))[:8]], uv_scale=_t([2.0, 2.0], device))
    cam = sc.camera
    for t in (cam.position, cam.look_at, cam.up):
        t.requires_grad_(True)
    if envmap_variant:
        from redner_amd.render_pytorch import EnvironmentMap
        h, w = 256, 512
        yy = (torch.arange(h, dtype=torch.float32) + 0.5) / h
        img = (0.2 + 1.3 * (1 - yy))[:, None, None].expand(h, w, 3).clone()
        img[40:44, 300:304] = 4e4
        env = EnvironmentMap(Texture([l.to(device) for l in _mip_chain(img)[:8]]))
        sc = Scene(cam, sc.shapes, sc.materials, sc.area_lights, envmap=env)
    return sc


def living_room_standin_envmap(device, resolution=(1024, 1024)):
    return living_room_standin(device, resolution, envmap_variant=True)
