"""Degenerate inputs the reference's drivers guard against (src/pathtracer.cpp:292-300: a sample with no live
paths, a scene without lights; src/scene.cpp:197: no light tables at all): nothing may crash, hang or
produce non-finite values, and where the oracle is available the results must equal it."""
import os

import numpy as np
import pytest
import torch

import oracle_util
import scenes
from redner_amd.render_pytorch import Camera, Scene, RenderFunction


def _run(backend, device, sc, spp=2, mb=2):
    for s in sc.shapes:
        s.vertices.requires_grad_(True)
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=backend.SamplerType.sobol, device=device,
                                          backend=backend)
    img = RenderFunction.apply(3, *args)
    img.sum().backward()
    grads = [s.vertices.grad.cpu().numpy() if s.vertices.grad is not None else None for s in sc.shapes]
    return img.detach().cpu().numpy(), grads


def _variants(device):
    def no_lights():
        sc = scenes.two_triangles(device, resolution=(24, 24))
        return Scene(sc.camera, sc.shapes[:2], sc.materials, [])
    def all_miss():
        sc = scenes.two_triangles(device, resolution=(24, 24))
        c = sc.camera
        sc.camera = Camera(position=c.position, look_at=torch.tensor([50.0, 0.0, -5.0]), up=c.up, fov=torch.tensor([20.0]),
                           clip_near=c.clip_near, resolution=(24, 24))
        return sc
    def no_bounces():
        return scenes.two_triangles(device, resolution=(24, 24))
    return {'no_lights': (no_lights, 2), 'all_miss': (all_miss, 2), 'no_bounces': (no_bounces, 0)}


@pytest.mark.parametrize('name', ['no_lights', 'all_miss', 'no_bounces'])
def test_degenerate_hostsim(hostsim_backend, name):
    build, mb = _variants(torch.device('cpu'))[name]
    img, grads = _run(hostsim_backend, torch.device('cpu'), build(), mb=mb)
    assert np.isfinite(img).all()
    if name in ('no_lights', 'all_miss'):
        assert not img.any()
    if oracle_util.oracle_available():
        ref_img, ref_grads = _run(oracle_util.load_oracle(), torch.device('cpu'), build(), mb=mb)
        assert np.array_equal(img, ref_img)
        for g, r in zip(grads, ref_grads):
            if r is None:
                continue
            n = np.linalg.norm(r)
            assert np.linalg.norm(g - r) <= 1e-4 * n + 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['no_lights', 'all_miss', 'no_bounces'])
def test_degenerate_gpu(gpu_backend, name):
    dev = torch.device('cuda:0')
    build, mb = _variants(dev)[name]
    img, grads = _run(gpu_backend, dev, build(), mb=mb)
    assert np.isfinite(img).all()
    if name in ('no_lights', 'all_miss'):
        assert not img.any()
    for g in grads:
        assert g is None or np.isfinite(g).all()
    if oracle_util.oracle_available():                 # oracle/_ref travels to the GPU box with the snapshot
        cpu = torch.device('cpu')
        ref_img, ref_grads = _run(oracle_util.load_oracle(), cpu, _variants(cpu)[name][0](), mb=mb)
        assert np.array_equal(img, ref_img)
        for g, r in zip(grads, ref_grads):
            if r is None:
                continue
            n = np.linalg.norm(r)
            assert np.linalg.norm(g - r) <= 1e-4 * n + 1e-12


def test_empty_scene_hostsim(hostsim_backend):
    """No shapes at all: a black image and no crash (src/scene.cpp:189-195 handles an empty shape list)."""
    sc = scenes.two_triangles(torch.device('cpu'), resolution=(16, 16))
    empty = Scene(sc.camera, [], [], [])
    args = RenderFunction.serialize_scene(empty, 1, 1, sampler_type=hostsim_backend.SamplerType.sobol,
                                          device=torch.device('cpu'), backend=hostsim_backend)
    img = RenderFunction.apply(1, *args)
    assert img.shape == (16, 16, 3) and not img.numpy().any()


@pytest.mark.gpu
def test_empty_scene_twice_gpu(gpu_backend):
    """Two empty Scenes in a row on the GPU (the second one found the first one's cached, empty hierarchy and
    tried to refit it with zero-size launches), then a normal Scene, then an empty one again."""
    dev = torch.device('cuda:0')
    sc = scenes.two_triangles(dev, resolution=(16, 16))
    for k in range(4):
        scene = sc if k == 2 else Scene(sc.camera, [], [], [])
        args = RenderFunction.serialize_scene(scene, 1, 1, sampler_type=gpu_backend.SamplerType.sobol, device=dev, backend=gpu_backend)
        img = RenderFunction.apply(1, *args)
        assert img.shape == (16, 16, 3) and bool(img.detach().cpu().numpy().any()) == (k == 2)


@pytest.mark.gpu
def test_steady_state_render_allocates_nothing_and_reads_no_counts(gpu_backend):
    """After a first forward+backward call has sized the caching allocator, further calls on the same shapes make no
    hipMalloc (the reference allocates its PathBuffer per call, src/pathtracer.cpp:36-152) and the host reads no live-lane
    count back (the reference reads one after every stage, :292,590,833) -- also when a new Scene is built per call, as
    pyredner does (render_pytorch.py:609)."""
    import ctypes
    from redner_amd import _capi
    from golden.make_golden import render_case
    dev = torch.device('cuda:0')

    def counters():
        c = _capi.DebugCounters()
        _capi.lib().rdr_debug_counters_get(ctypes.byref(c))
        return int(c.device_mallocs), int(c.host_count_reads)

    for _ in range(2):                                            # warm the pool (two calls: both sample workers' buffers)
        render_case(gpu_backend, 'bunny_box', 96, 8, 4, device=dev)
    mallocs0, reads0 = counters()
    for _ in range(3):
        render_case(gpu_backend, 'bunny_box', 96, 8, 4, device=dev)
    mallocs1, reads1 = counters()
    assert mallocs1 == mallocs0, (mallocs0, mallocs1)
    assert reads1 == reads0, (reads0, reads1)


def _refit_vs_fresh(tmp_path, lib, dev):
    """A Scene whose index buffers equal the previous Scene's reuses the triangle hierarchy's and the billboard hierarchy's
    topology and the id-sorted half of the edge list, refitting boxes to the moved vertices (scene.cpp / edges.cpp).  Hits and
    edge picks must not depend on that: the result equals a process that builds everything from scratch (RDR_NO_REFIT=1), bit
    for bit on the image and to fp32-atomics noise on the gradients."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import os, sys
sys.path[:0] = [%r, %r + '/tests']
import numpy as np, torch
from redner_amd import _capi
_capi.load(%r)
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
import scenes
dev = torch.device(%r)
def run(shift):
    sc = scenes.bunny_box(dev, (40, 40))
    v = sc.shapes[6].vertices.detach().cpu()
    g = torch.Generator().manual_seed(7)
    moved = (v * (1.0 + 0.15 * shift) + shift * torch.tensor([0.12, 0.05, -0.2]) + 0.01 * shift * torch.randn(v.shape, generator=g)).to(dev).requires_grad_(True)
    sc.shapes[6].vertices = moved
    args = RenderFunction.serialize_scene(sc, 4, 3, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner)
    img = RenderFunction.apply(1, *args)
    img.sum().backward()
    return img.detach().cpu().numpy(), moved.grad.cpu().numpy()
run(0.0)                       # the Scene whose topology the next one inherits (unless RDR_NO_REFIT)
img, grad = run(1.0)
np.savez(sys.argv[1], image=img, grad=grad)
''' % (ROOT, ROOT, lib, dev)
    outs = {}
    for tag, env in (('refit', {}), ('fresh', {'RDR_NO_REFIT': '1'})):
        out = str(tmp_path / (tag + '.npz'))
        subprocess.check_call([sys.executable, '-c', code, out], env=dict(os.environ, **env), timeout=900)
        outs[tag] = np.load(out)
    assert np.array_equal(outs['refit']['image'], outs['fresh']['image'])
    a, b = outs['refit']['grad'].astype(np.float64), outs['fresh']['grad'].astype(np.float64)
    assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
    assert np.abs(b).sum() > 0


def test_refit_after_vertices_moved_equals_fresh_build(hostsim_backend, tmp_path):
    from conftest import HOSTSIM_LIB
    _refit_vs_fresh(tmp_path, HOSTSIM_LIB, 'cpu')


@pytest.mark.gpu
def test_refit_after_vertices_moved_equals_fresh_build_gpu(gpu_backend, tmp_path):
    from redner_amd import _capi
    _refit_vs_fresh(tmp_path, _capi.library_path(), 'cuda:0')


def test_edge_build_beside_the_caller_equals_build_in_place(hostsim_backend, tmp_path):
    """create_scene() starts the edge structures' build on its own thread and the first gradient render joins it (scene.h).
    Covered here: Scenes created back to back (their builds queue up on the build lock), gradient renders in the other order,
    a Scene dropped without ever being rendered backward (the destructor joins), and equality -- bit for bit on the images,
    fp32-atomics noise on the gradients -- with a process that builds inside create_scene() (RDR_SYNC_EDGES=1)."""
    import subprocess
    import sys
    from conftest import HOSTSIM_LIB, ROOT
    code = r'''
import gc, os, sys
sys.path[:0] = [%r, %r + '/tests']
import numpy as np, torch
from redner_amd import _capi
_capi.load(%r)
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
import scenes
dev = torch.device('cpu')
def make(res, shift):
    sc = scenes.bunny_box(dev, (res, res))
    v = (sc.shapes[6].vertices.detach() + shift).requires_grad_(True)
    sc.shapes[6].vertices = v
    args = RenderFunction.serialize_scene(sc, 2, 2, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner)
    return v, args
va, a = make(24, 0.0)
vb, b = make(32, 0.05)
vc, c = make(16, 0.1)
img_a = RenderFunction.apply(1, *a)          # three Scenes alive, three builds in flight or queued
img_b = RenderFunction.apply(2, *b)
img_c = RenderFunction.apply(3, *c)
del img_c, c; gc.collect()                   # never rendered backward: dropped with its build possibly still running
img_b.sum().backward()
img_a.sum().backward()
np.savez(sys.argv[1], img_a=img_a.detach().numpy(), img_b=img_b.detach().numpy(), ga=va.grad.numpy(), gb=vb.grad.numpy())
''' % (ROOT, ROOT, HOSTSIM_LIB)
    outs = {}
    for tag, env in (('beside', {}), ('in_place', {'RDR_SYNC_EDGES': '1'})):
        out = str(tmp_path / (tag + '.npz'))
        subprocess.check_call([sys.executable, '-c', code, out], env=dict(os.environ, **env), timeout=900)
        outs[tag] = np.load(out)
    for k in ('img_a', 'img_b'):
        assert np.array_equal(outs['beside'][k], outs['in_place'][k])
    for k in ('ga', 'gb'):
        x, y = outs['beside'][k].astype(np.float64), outs['in_place'][k].astype(np.float64)
        assert np.abs(y).sum() > 0
        assert np.linalg.norm(x - y) <= 1e-6 * np.linalg.norm(y)


def _shared_gradient_buffer(backend, device):
    """Two materials whose DMaterial entries point at ONE gradient buffer (possible at the C boundary: the DScene holds plain
    pointers): the buffer must receive the sum of what each material would have received alone."""
    sc = scenes.two_triangles(device, resolution=(24, 24))
    args = RenderFunction.serialize_scene(sc, 2, 2, sampler_type=backend.SamplerType.sobol, device=device, backend=backend)
    meta, tensors = args[0], list(args[1:])
    u = RenderFunction.unpack_args((3, 3), meta, tensors)
    grad_img = torch.ones(24, 24, 3, device=device)

    def backward(m):
        d_scene, grads = RenderFunction.create_gradient_buffers(m, tensors)
        backend.render(u.scene, u.options, backend.float_ptr(0), backend.float_ptr(grad_img.data_ptr()), d_scene,
                       backend.float_ptr(0), backend.float_ptr(0))
        return grads

    i0 = meta['materials'][0]['diffuse_reflectance']['levels'][0]
    i1 = meta['materials'][1]['diffuse_reflectance']['levels'][0]
    apart = backward(meta)
    import copy
    shared_meta = copy.copy(meta)
    shared_meta['materials'] = copy.deepcopy(meta['materials'])
    shared_meta['materials'][1]['diffuse_reflectance']['levels'][0] = i0
    shared = backward(shared_meta)
    want = (apart[i0] + apart[i1]).cpu().numpy()
    got = shared[i0].cpu().numpy()
    assert np.linalg.norm(want) > 0
    assert np.linalg.norm(got - want) <= 1e-5 * np.linalg.norm(want)
    cam = meta['camera']['position']
    assert np.linalg.norm((shared[cam] - apart[cam]).cpu().numpy()) <= 1e-5 * np.linalg.norm(apart[cam].cpu().numpy())


def test_gradient_buffer_shared_by_two_entries_hostsim(hostsim_backend):
    _shared_gradient_buffer(hostsim_backend, torch.device('cpu'))


@pytest.mark.gpu
def test_gradient_buffer_shared_by_two_entries_gpu(gpu_backend):
    _shared_gradient_buffer(gpu_backend, torch.device('cuda:0'))


def _many_patches(device):
    """Twelve 45 x 45-vertex patches: 72 900 doubles of vertex gradients in tensors of 6 075 each -- more than the small tier
    of the gradient store holds (render.cpp: GradStore, 65 536 doubles), so the last patches' accumulators live in the
    large tier and the flush has segments in both."""
    n = 45
    u = np.linspace(0.0, 1.0, n, dtype=np.float32)
    gx, gy = np.meshgrid(u, u, indexing='xy')
    idx = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            idx += [[a, a + n, a + 1], [a + 1, a + n, a + n + 1]]
    idx = np.asarray(idx, np.int32)
    shapes = []
    for k in range(12):
        ox, oy = -2.0 + (k % 4), -1.5 + (k // 4)
        z = 0.15 * np.sin(3.0 * gx + k) * np.cos(2.0 * gy) + 0.05 * k
        v = np.stack([ox + gx, oy + gy, z], axis=2).reshape(-1, 3).astype(np.float32)
        shapes.append(scenes.Shape(torch.tensor(v, device=device, requires_grad=True), torch.tensor(idx, device=device), 0))
    light = scenes.Shape(torch.tensor([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]], device=device),
                         torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32, device=device), 1)
    cam = Camera(position=torch.tensor([0.0, 0.0, -5.0]), look_at=torch.tensor([0.0, 0.0, 0.0]), up=torch.tensor([0.0, 1.0, 0.0]),
                 fov=torch.tensor([45.0]), clip_near=1e-2, resolution=(32, 32))
    mats = [scenes.Material(diffuse_reflectance=torch.tensor([0.5, 0.55, 0.45], device=device)),
            scenes.Material(diffuse_reflectance=torch.tensor([0.0, 0.0, 0.0], device=device))]
    return Scene(cam, shapes + [light], mats, [scenes.AreaLight(12, torch.tensor([20.0, 20.0, 20.0]))])


def test_gradient_store_spills_into_the_large_tier_hostsim(hostsim_backend):
    if not oracle_util.oracle_available():
        pytest.skip('oracle not built')
    img, grads = _run(hostsim_backend, torch.device('cpu'), _many_patches(torch.device('cpu')), spp=2, mb=1)
    ref_img, ref_grads = _run(oracle_util.load_oracle(), torch.device('cpu'), _many_patches(torch.device('cpu')), spp=2, mb=1)
    assert np.array_equal(img, ref_img)
    seen = 0
    for g, r in zip(grads, ref_grads):
        if r is None:
            continue
        n = np.linalg.norm(r)
        seen += n > 0
        assert np.linalg.norm(g - r) <= 1e-4 * n + 1e-12
    assert seen >= 10            # patches of both tiers received gradients


def _nonfinite_through_data(backend, device):
    """pyredner asserts isfinite on every serialize_scene (render_pytorch.py:194-270); a value written through `.data` or a
    numpy alias does not bump `_version`, so no version-keyed cache may hide it (advisor finding, round 3)."""
    sc = scenes.two_triangles(device, resolution=(16, 16))
    kw = dict(sampler_type=backend.SamplerType.sobol, device=device, backend=backend)
    RenderFunction.serialize_scene(sc, 1, 1, **kw)
    RenderFunction.serialize_scene(sc, 1, 1, **kw)
    v = sc.shapes[0].vertices
    v.data[0, 0] = float('nan')                      # what `p.data.clamp_()`-style loops do: no version bump
    with pytest.raises(AssertionError):
        RenderFunction.serialize_scene(sc, 1, 1, **kw)
    v.data[0, 0] = 0.0
    # a big static tensor (no gradient): cached by version, and an in-place write through the tensor itself is seen
    big = torch.zeros(300, 300, 3, device=device)
    from redner_amd.render_pytorch import Texture
    sc.materials[0].diffuse_reflectance = Texture([big])
    RenderFunction.serialize_scene(sc, 1, 1, **kw)
    RenderFunction.serialize_scene(sc, 1, 1, **kw)
    big[5, 5, 0] = float('inf')
    with pytest.raises(AssertionError):
        RenderFunction.serialize_scene(sc, 1, 1, **kw)


def test_nonfinite_written_through_data_is_caught_hostsim(hostsim_backend):
    _nonfinite_through_data(hostsim_backend, torch.device('cpu'))


@pytest.mark.gpu
def test_nonfinite_written_through_data_is_caught_gpu(gpu_backend):
    _nonfinite_through_data(gpu_backend, torch.device('cuda:0'))


@pytest.mark.gpu
def test_two_host_threads_on_one_device_gpu(gpu_backend):
    """ctypes releases the GIL: two Python threads may be inside rdr_render / rdr_scene_create at once.  Calls on ONE device are
    serialised by that device's lock (csrc/capi.cpp; calls on different devices are not): the results of interleaved calls equal
    the results of the same calls made one after the other."""
    import threading
    from golden.make_golden import render_case
    dev = torch.device('cuda:0')
    jobs = [('bunny_box', 48, 4, 4), ('two_triangles', 64, 4, 1)]
    alone = [render_case(gpu_backend, *j, device=dev) for j in jobs]
    got = [[], []]

    def work(k):
        for _ in range(3):
            got[k].append(render_case(gpu_backend, *jobs[k], device=dev))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(2):
        assert len(got[k]) == 3
        for out in got[k]:
            assert np.array_equal(out['image'], alone[k]['image'])
            for key, ref in alone[k].items():
                n = np.linalg.norm(ref.astype(np.float64))
                assert np.linalg.norm(out[key].astype(np.float64) - ref) <= 1e-6 * n + 1e-12, (k, key)
