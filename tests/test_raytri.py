"""The closest-hit / any-hit rule (redner_amd/csrc/raytri.h): the product's SAH hierarchy and
the Embree stand-in of the oracle must both equal a brute-force scan with the shared fp32
predicate -- same (shape, triangle) for every ray, including grazing and tie cases."""
import ctypes

import numpy as np
import pytest
import torch

import scenes
from redner_amd.render_pytorch import RenderFunction


def _brute_force(verts_list, inds_list, rays, any_hit):
    # numpy fp32 restatement of rt::ray_triangle / rt::closer (no fused operations in numpy)
    f = np.float32
    tris = []
    for s, (v, i) in enumerate(zip(verts_list, inds_list)):
        for p, (a, b, c) in enumerate(i):
            tris.append((v[a], v[b], v[c], s, p))
    A = np.array([t[0] for t in tris], f); B = np.array([t[1] for t in tris], f); C = np.array([t[2] for t in tris], f)
    S = np.array([t[3] for t in tris]); P = np.array([t[4] for t in tris])
    out = np.full((len(rays), 2), -1, np.int32)
    e1, e2 = B - A, C - A
    for r, ray in enumerate(rays):
        o, tn, d, tf = ray[0:3], ray[3], ray[4:7], ray[7]
        if tf < 0:
            continue
        px = d[1] * e2[:, 2] - d[2] * e2[:, 1]; py = d[2] * e2[:, 0] - d[0] * e2[:, 2]; pz = d[0] * e2[:, 1] - d[1] * e2[:, 0]
        det = (e1[:, 0] * px + e1[:, 1] * py) + e1[:, 2] * pz
        with np.errstate(divide='ignore', invalid='ignore'):
            inv = f(1) / det
            s = o[None, :] - A
            u = ((s[:, 0] * px + s[:, 1] * py) + s[:, 2] * pz) * inv
            qx = s[:, 1] * e1[:, 2] - s[:, 2] * e1[:, 1]; qy = s[:, 2] * e1[:, 0] - s[:, 0] * e1[:, 2]; qz = s[:, 0] * e1[:, 1] - s[:, 1] * e1[:, 0]
            v = ((d[0] * qx + d[1] * qy) + d[2] * qz) * inv
            t = ((e2[:, 0] * qx + e2[:, 1] * qy) + e2[:, 2] * qz) * inv
        ok = (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > tn) & (t < tf)
        idx = np.nonzero(ok)[0]
        if len(idx) == 0:
            continue
        if any_hit:
            out[r] = (0, 0)          # only hit / miss is defined for any-hit
            continue
        order = sorted(idx, key=lambda k: (t[k], S[k], P[k]))
        out[r] = (S[order[0]], P[order[0]])
    return out


def _scene_handle(backend, builder, res):
    sc = getattr(scenes, builder)(torch.device('cpu'), resolution=(res, res))
    args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=backend.SamplerType.sobol, device=torch.device('cpu'),
                                          backend=backend)
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    return sc, u


def _rays(sc, n, seed):
    rng = np.random.default_rng(seed)
    lo = np.min([s.vertices.detach().numpy().min(0) for s in sc.shapes], 0) - 0.5
    hi = np.max([s.vertices.detach().numpy().max(0) for s in sc.shapes], 0) + 0.5
    o = rng.uniform(lo, hi, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 1e-3, d, np.inf
    # rays aimed exactly at vertices / edge midpoints (ties between adjacent triangles)
    sh = sc.shapes[-1]
    v = sh.vertices.detach().numpy(); ind = sh.indices.numpy()
    k = min(n // 4, len(ind))
    tgt = np.concatenate([v[ind[:k, 0]], 0.5 * (v[ind[:k, 0]] + v[ind[:k, 1]])])
    dd = tgt - rays[:len(tgt), 0:3]
    rays[:len(tgt), 4:7] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    rays[::17, 7] = 1.5          # finite tfar
    rays[::29, 7] = -1.0         # dead slots
    return rays


@pytest.mark.parametrize('builder', ['two_triangles', 'bunny_box'])
@pytest.mark.parametrize('any_hit', [0, 1])
def test_hierarchy_equals_brute_force(hostsim_backend, builder, any_hit):
    from redner_amd import _capi
    sc, u = _scene_handle(hostsim_backend, builder, 16)
    n = 600 if builder == 'bunny_box' else 2000
    rays = _rays(sc, n, 7)
    hits = np.zeros((n, 2), np.int32)
    rc = _capi.lib().rdr_scene_trace(u.scene._handle, rays.ctypes.data_as(ctypes.c_void_p),
                                     hits.ctypes.data_as(ctypes.c_void_p), n, any_hit)
    assert rc == 0
    ref = _brute_force([s.vertices.detach().numpy() for s in sc.shapes], [s.indices.numpy() for s in sc.shapes], rays, any_hit)
    if any_hit:
        assert np.array_equal(hits[:, 0] >= 0, ref[:, 0] >= 0)
    else:
        assert np.array_equal(hits, ref)
    assert (hits[:, 0] >= 0).sum() > 50          # the comparison is not vacuous


@pytest.mark.gpu
@pytest.mark.parametrize('any_hit', [0, 1])
def test_gpu_traversal_equals_host_rule(gpu_backend, any_hit):
    """The gfx950 kernel returns bit-identical hit ids to the brute-force rule (numpy)."""
    from redner_amd import _capi
    dev = torch.device('cuda:0')
    sc = scenes.bunny_box(dev, resolution=(16, 16))
    args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=gpu_backend.SamplerType.sobol, device=dev)
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    cpu_sc = scenes.bunny_box(torch.device('cpu'), resolution=(16, 16))
    n = 600
    rays = _rays(cpu_sc, n, 11)
    d_rays = torch.from_numpy(rays).to(dev)
    d_hits = torch.zeros(n, 2, dtype=torch.int32, device=dev)
    rc = _capi.lib().rdr_scene_trace(u.scene._handle, d_rays.data_ptr(), d_hits.data_ptr(), n, any_hit)
    assert rc == 0
    hits = d_hits.cpu().numpy()
    ref = _brute_force([s.vertices.detach().numpy() for s in cpu_sc.shapes], [s.indices.numpy() for s in cpu_sc.shapes], rays, any_hit)
    if any_hit:
        assert np.array_equal(hits[:, 0] >= 0, ref[:, 0] >= 0)
    else:
        assert np.array_equal(hits, ref)


# ---- the reference's own known-answer test at this boundary, through the product ------------------------------------
def _kat_scene(device):
    """test_scene_intersect (src/scene.cpp:761-848): one triangle at z = 1, a 1 x 1 camera at the origin looking down +z
    with identity intrinsics; ray 0 = (0,0,0)->(0,0,1) must hit (shape 0, triangle 0) at (0,0,1), ray 1 = ->(0,0,-1) must miss."""
    from redner_amd.render_pytorch import Camera, Material, Scene, Shape
    cam = Camera(position=torch.tensor([0.0, 0.0, 0.0]), look_at=torch.tensor([0.0, 0.0, 1.0]), up=torch.tensor([0.0, 1.0, 0.0]),
                 intrinsic_mat=torch.eye(3), clip_near=1e-2, resolution=(1, 1))
    tri = Shape(torch.tensor([[-1.0, 0.0, 1.0], [1.0, 0.0, 1.0], [0.0, 1.0, 1.0]], device=device),
                torch.tensor([[0, 1, 2]], dtype=torch.int32, device=device), 0)
    return Scene(cam, [tri], [Material(diffuse_reflectance=torch.tensor([0.5, 0.5, 0.5], device=device))], [])


def _run_reference_kat(backend, device):
    from redner_amd import _capi
    sc = _kat_scene(device)
    args = RenderFunction.serialize_scene(sc, 1, 0, channels=[backend.channels.position, backend.channels.shape_id,
                                                            backend.channels.triangle_id],
                                          sampler_type=backend.SamplerType.sobol, device=device, backend=backend,
                                          sample_pixel_center=True)
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    rays = torch.tensor([[0, 0, 0, 1e-3, 0, 0, 1, float('inf')], [0, 0, 0, 1e-3, 0, 0, -1, float('inf')]],
                        dtype=torch.float32, device=device)
    for any_hit in (0, 1):
        hits = torch.full((2, 2), 7, dtype=torch.int32, device=device)
        assert _capi.lib().rdr_scene_trace(u.scene._handle, rays.data_ptr(), hits.data_ptr(), 2, any_hit) == 0
        h = hits.cpu().numpy()
        if any_hit:
            assert h[0, 0] >= 0 and h[1, 0] < 0
        else:
            assert h.tolist() == [[0, 0], [-1, -1]]          # isects[0] = (0, 0), isects[1] = (-1, -1)
    # surface_points[0].position == (0, 0, 1): the fp64 re-intersection, seen through the position channel of the
    # camera ray of that 1 x 1 image (the same ray as ray 0)
    img = RenderFunction.apply(1, *args).cpu().numpy()
    assert np.allclose(img[0, 0, 0:3], [0.0, 0.0, 1.0], atol=1e-6)
    assert img[0, 0, 3] == 0 and img[0, 0, 4] == 0


def test_reference_scene_intersect_kat_hostsim(hostsim_backend):
    _run_reference_kat(hostsim_backend, torch.device('cpu'))


@pytest.mark.gpu
def test_reference_scene_intersect_kat_gpu(gpu_backend):
    _run_reference_kat(gpu_backend, torch.device('cuda:0'))


@pytest.mark.gpu
@pytest.mark.parametrize('env', [{'RDR_TRACE_BINARY': '1'},
                                 {'RDR_TRACE_BINARY': '1', 'RDR_TRACE_REFILL_ALL': '1', 'RDR_TRACE_REFILL': '2,4,2'},
                                 {'RDR_TRACE_BINARY': '1', 'RDR_TRACE_REFILL_ALL': '1', 'RDR_TRACE_REFILL': '4,24,4'}],
                         ids=['binary', 'refill_2_4_2', 'refill_4_24_4'])
def test_gpu_traversal_kernel_forms(gpu_backend, env):
    """exec::trace() picks one of three kernels by queue size (4-wide records for small queues, which is what the tests above
    run; binary records; binary records with lanes that take the next ray of their wave's chunk for large incoherent queues).
    The switches force the other two onto the small queues of the tests (read once per process: a subprocess): the
    brute-force rule, the reference's known-answer vectors and a gradient fixture must hold for each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.join(root, 'tests', 'test_raytri.py'),
           os.path.join(root, 'tests', 'test_backward_parity.py'), '-k',
           'test_gpu_traversal_equals_host_rule or test_reference_scene_intersect_kat_gpu or (test_backward_gpu and bunny_box_32x32x4)']
    r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-500:]


BIG_WORKER = r"""
import os, sys
sys.path[:0] = [%(root)r, %(root)r + '/tests']
import numpy as np, torch
from redner_amd import _capi, redner
from redner_amd.render_pytorch import RenderFunction
import scenes
dev = torch.device('cuda:0')
sc = scenes.bunny_box_subdivided(dev, resolution=(16, 16), levels=2)          # 230 k triangles: > 65 535 node records
args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=redner.SamplerType.sobol, device=dev,
                                      use_primary_edge_sampling=False, use_secondary_edge_sampling=False)
u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
g = torch.Generator().manual_seed(5)
n = 300000
o = (torch.rand(n, 3, generator=g) - 0.5) * 1.6
d = torch.randn(n, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
rays = torch.zeros(n, 8); rays[:, 0:3] = o; rays[:, 3] = 1e-3; rays[:, 4:7] = d; rays[:, 7] = float('inf')
rays[::7, 7] = -1.0                                # dead slots
rays = rays.to(dev).contiguous()
out = {}
for any_hit in (0, 1):
    hits = torch.zeros(n, 2, dtype=torch.int32, device=dev)
    assert _capi.lib().rdr_scene_trace(u.scene._handle, rays.data_ptr(), hits.data_ptr(), n, any_hit) == 0
    h = hits.cpu().numpy()
    out['h%%d' %% any_hit] = h if not any_hit else (h[:, 0] >= 0)
np.savez(%(out)r, **out)
"""


@pytest.mark.gpu
def test_big_hierarchy_kernel_forms_agree(gpu_backend, tmp_path):
    """Hierarchies with more than 65 535 node records take the int-entry forms of the kernels: the 4-wide records (default for a
    queue this size), the plain binary kernel, the refilling kernel with the hybrid LDS + scratch stack and with the 32 / 40-entry
    LDS tiers, each with and without the octant order -- all must return the same hit ids (closest) / the same verdict (any)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    forms = [('wide', {}), ('binary', {'RDR_TRACE_BINARY': '1'}),
             ('refill_hybrid', {'RDR_TRACE_BINARY': '1', 'RDR_TRACE_REFILL_ALL': '1'}),
             ('refill_hybrid_queue_order', {'RDR_TRACE_BINARY': '1', 'RDR_TRACE_REFILL_ALL': '1', 'RDR_REFILL_SORT': '0'}),
             ('refill_tiers', {'RDR_TRACE_BINARY': '1', 'RDR_TRACE_REFILL_ALL': '1', 'RDR_TRACE_HYBRID': '0'}),
             ('refill_tiers_axis_order', {'RDR_TRACE_BINARY': '1', 'RDR_TRACE_REFILL_ALL': '1', 'RDR_TRACE_HYBRID': '0', 'RDR_REFILL_SORT': '2'})]
    first = None
    for name, env in forms:
        out = str(tmp_path / (name + '.npz'))
        script = tmp_path / (name + '.py')
        script.write_text(BIG_WORKER % {'root': root, 'out': out})
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        z = np.load(out)
        if first is None:
            first = {k: z[k] for k in z.files}
            assert (first['h0'][:, 0] >= 0).sum() > 100000 and (first['h0'][::7, 0] == -1).all()
        else:
            assert np.array_equal(z['h0'], first['h0']), name
            assert np.array_equal(z['h1'], first['h1']), name


# ---- the hierarchy the kernels build (bvh_gpu.cpp) ----------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('builder', ['single_triangle', 'two_triangles', 'bunny_box', 'living_room_standin', 'triangle_soup_large'])
def test_device_built_hierarchy_equals_host_builder(gpu_backend, builder):
    """Binned SAH by kernels, level by level: node records, leaf order, triangle records and the 4-wide records equal what the
    host builder (bvh.cpp) makes of the same arrays -- every record (rdr_debug_bvh_check downloads and compares)."""
    from redner_amd import _capi
    dev = torch.device('cuda:0')
    gpu_backend.set_build_flags(_capi.BUILD_NO_REFIT)         # a fresh build, whatever the previous test left in the caches
    try:
        sc = getattr(scenes, builder)(dev, resolution=(16, 16))
        args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=gpu_backend.SamplerType.sobol, device=dev)
        u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
        assert _capi.lib().rdr_debug_bvh_check(u.scene._handle) == 0
    finally:
        gpu_backend.set_build_flags(0)


@pytest.mark.gpu
@pytest.mark.parametrize('any_hit', [0, 1])
def test_refitted_hierarchy_equals_host_rule(gpu_backend, any_hit):
    """A Scene with the previous Scene's connectivity and moved vertices: its hierarchy is a device-side refit of the previous
    build (triangle records, boxes level by level, wide records); hits still equal the brute-force rule, for both forms."""
    from redner_amd import _capi
    dev = torch.device('cuda:0')

    def make(shift):
        sc = scenes.bunny_box(dev, resolution=(16, 16))
        cpu_sc = scenes.bunny_box(torch.device('cpu'), resolution=(16, 16))
        for s, c in zip(sc.shapes, cpu_sc.shapes):
            n = c.vertices.numel()
            delta = shift * torch.sin(torch.arange(n, dtype=torch.float32) * 0.37).reshape(c.vertices.shape)
            c.vertices = (c.vertices.detach() + delta)
            s.vertices = c.vertices.to(dev)
        args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=gpu_backend.SamplerType.sobol, device=dev)
        return cpu_sc, RenderFunction.unpack_args((1, 2), args[0], args[1:])

    _, first = make(0.0)                                   # (keeps the topology cache's build alive)
    cpu_sc, u = make(0.0005)
    assert _capi.lib().rdr_debug_bvh_check(u.scene._handle) == -1        # a refit, not a build
    n = 600
    rays = _rays(cpu_sc, n, 5)
    d_rays = torch.from_numpy(rays).to(dev)
    ref = _brute_force([s.vertices.detach().numpy() for s in cpu_sc.shapes], [s.indices.numpy() for s in cpu_sc.shapes], rays, any_hit)
    d_hits = torch.zeros(n, 2, dtype=torch.int32, device=dev)
    assert _capi.lib().rdr_scene_trace(u.scene._handle, d_rays.data_ptr(), d_hits.data_ptr(), n, any_hit) == 0      # (the 4-wide records)
    hits = d_hits.cpu().numpy()
    if any_hit:
        assert np.array_equal(hits[:, 0] >= 0, ref[:, 0] >= 0)
    else:
        assert np.array_equal(hits, ref)
