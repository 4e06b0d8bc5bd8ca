"""Drop-in check ON THE GPU: the reference's UNMODIFIED Python package (pyredner/render_pytorch.py, staged by
oracle/Makefile into the untracked oracle/_ref/pyref/ so that it travels to the GPU box) drives the PRODUCT library
(libredner_amd.so) with pyredner.set_use_gpu(True): device-resident vertex / index / image / gradient tensors, the
device index and the caller's stream all go through pyredner's own unpack_args / forward / backward
(/root/reference/pyredner/render_pytorch.py:272-649,652-707,1050-1177).  The oracle leg runs the same script with
set_use_gpu(False) on the reference's own C++ core (oracle/_ref) in the same test.

Bars: image bit-identical, every gradient tensor <= 1e-4 relative L2 (north_star)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYREF = os.path.join(ROOT, 'oracle', '_ref', 'pyref')

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isfile(os.path.join(PYREF, 'pyredner', 'render_pytorch.py')),
                                 reason='oracle/_ref/pyref not staged (make -C oracle)')]

PROLOGUE = r'''
import os, sys
sys.path[:0] = [%(root)r, %(root)r + '/tests', %(root)r + '/oracle/pystubs', %(pyref)r]
import numpy as np, torch
which, out, mode = sys.argv[1], sys.argv[2], sys.argv[3]
if which == 'mine':
    import redner_amd
    from redner_amd import _capi
    redner_amd.install()                      # `import redner` now resolves to redner_amd.redner
    _capi.load()                              # the PRODUCT library; raises if libredner_amd.so is missing
    assert _capi.is_product_library(), _capi.library_path()
else:
    import oracle_util
    sys.modules['redner'] = oracle_util.load_oracle()
import redner, pyredner                        # the reference's package, unmodified
assert os.path.dirname(os.path.abspath(pyredner.__file__)) == %(pyref)r + '/pyredner'
pyredner.set_print_timing(False)
pyredner.set_use_gpu(which == 'mine')
if which == 'mine':
    assert torch.cuda.is_available() and pyredner.get_device().type == 'cuda'


def save(img, grads):
    assert img.device.type == ('cuda' if which == 'mine' else 'cpu')
    for g in grads.values():
        assert g is not None
    np.savez(out, image=img.detach().cpu().numpy(), **{k: g.detach().cpu().numpy() for k, g in grads.items()})


def run(scene, spp, bounces, leaves):
    """serialize -> RenderFunction.apply -> backward, optionally under a caller stream."""
    args = pyredner.RenderFunction.serialize_scene(scene=scene, num_samples=spp, max_bounces=bounces,
                                                   sampler_type=redner.SamplerType.sobol)
    if mode == 'stream' and which == 'mine':
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            img = pyredner.RenderFunction.apply(1, *args)
            loss = (img * img).sum()
            loss.backward()
        torch.cuda.current_stream().wait_stream(s)
    else:
        img = pyredner.RenderFunction.apply(1, *args)
        (img * img).sum().backward()
    save(img, {k: v.grad for k, v in leaves.items()})
'''

SINGLE_TRIANGLE = PROLOGUE + r'''
dev = pyredner.get_device()
cam = pyredner.Camera(position=torch.tensor([0.0, 0.0, -5.0]), look_at=torch.tensor([0.0, 0.0, 0.0]),
                      up=torch.tensor([0.0, 1.0, 0.0]), fov=torch.tensor([45.0]), clip_near=1e-2, resolution=(64, 64))
cam.position.requires_grad = True
mats = [pyredner.Material(diffuse_reflectance=torch.tensor([0.5, 0.5, 0.5], device=dev, requires_grad=True))]
verts = torch.tensor([[-2.0, 1.5, 0.3], [0.9, 1.2, -0.3], [-0.4, -1.4, 0.2]], device=dev, requires_grad=True)
tri = pyredner.Shape(vertices=verts, indices=torch.tensor([[0, 1, 2]], dtype=torch.int32, device=dev),
                     uvs=None, normals=None, material_id=0)
light = pyredner.Shape(vertices=torch.tensor([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]],
                                             device=dev),
                       indices=torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32, device=dev),
                       uvs=None, normals=None, material_id=0)
scene = pyredner.Scene(cam, [tri, light], mats, [pyredner.AreaLight(shape_id=1, intensity=torch.tensor([20.0, 20.0, 20.0]))])
run(scene, 4, 1, {'vertices': verts, 'cam_position': cam.position, 'diffuse': mats[0].diffuse_reflectance.texels})
'''

BUNNY_BOX = PROLOGUE + r'''
os.chdir(%(pyref)r)
scene = pyredner.load_mitsuba('scenes/bunny_box.xml')          # the reference's tests/test_bunny_box.py recipe
dev = pyredner.get_device()
assert scene.shapes[-1].vertices.device.type == dev.type
# tests/test_bunny_box.py:25-32: the bunny's vertices as a function of a translation and Euler angles.  The pose is applied
# with HOST tensor arithmetic in both legs and the result moved to the render device: the renderer is what is compared here,
# not torch's CPU matmul / sin / mean against rocBLAS' (they differ in the last bit, which moves edges across pixel samples)
shape0_vertices = scene.shapes[-1].vertices.detach().cpu().clone()
translation = torch.tensor([0.1, -0.1, 0.1], requires_grad=True)
euler = torch.tensor([0.1, -0.1, 0.1], requires_grad=True)
center = torch.mean(shape0_vertices, 0)
cx, cy, cz = torch.cos(euler[0]), torch.cos(euler[1]), torch.cos(euler[2])
sx, sy, sz = torch.sin(euler[0]), torch.sin(euler[1]), torch.sin(euler[2])
one, zero = torch.ones(()), torch.zeros(())
rx = torch.stack([torch.stack([one, zero, zero]), torch.stack([zero, cx, -sx]), torch.stack([zero, sx, cx])])
ry = torch.stack([torch.stack([cy, zero, sy]), torch.stack([zero, one, zero]), torch.stack([-sy, zero, cy])])
rz = torch.stack([torch.stack([cz, -sz, zero]), torch.stack([sz, cz, zero]), torch.stack([zero, zero, one])])
rot = rz @ (ry @ rx)
posed = (shape0_vertices - center) @ torch.t(rot) + center + translation
scene.shapes[-1].vertices = posed.to(dev)
scene.shapes[-1].vertices.retain_grad()
scene.camera.resolution = (48, 48)
run(scene, 4, 4, {'translation': translation, 'euler': euler, 'vertices': scene.shapes[-1].vertices})
'''


def _both(tmp_path, name, body, mode, timeout=900):
    import numpy as np
    script = tmp_path / (name + '.py')
    script.write_text(body % {'root': ROOT, 'pyref': PYREF})
    outs = {}
    env = dict(os.environ, MALLOC_PERTURB_='255')          # the reference reads scratch it never wrote (profiles/r4_notes.md)
    for which in ('mine', 'oracle'):
        out = str(tmp_path / ('%s_%s.npz' % (name, which)))
        subprocess.check_call([sys.executable, str(script), which, out, mode], timeout=timeout, env=env)
        outs[which] = np.load(out)
    return outs['mine'], outs['oracle']


def _check(a, b):
    import numpy as np
    assert a['image'].shape == b['image'].shape
    assert np.array_equal(a['image'], b['image']), 'image differs from the oracle'
    for k in b.files:
        if k == 'image':
            continue
        ref = b[k].astype(np.float64)
        rel = np.linalg.norm(a[k].astype(np.float64) - ref) / np.linalg.norm(ref)
        assert rel < 1e-4, (k, rel)


@pytest.mark.parametrize('mode', ['plain', 'stream'])
def test_unmodified_pyredner_single_triangle_on_gpu(tmp_path, mode):
    _check(*_both(tmp_path, 'single_triangle', SINGLE_TRIANGLE, mode))


def test_unmodified_pyredner_load_mitsuba_bunny_box_on_gpu(tmp_path):
    _check(*_both(tmp_path, 'bunny_box', BUNNY_BOX, 'plain'))
