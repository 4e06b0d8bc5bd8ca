"""Drop-in check: the reference's UNMODIFIED Python package (pyredner/render_pytorch.py) drives
our `redner` module.  Runs only where the reference checkout is mounted (the build container); the
host debugging harness stands in for the GPU so the whole RenderFunction forward+backward path of
pyredner is exercised and compared with the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import sys
sys.path[:0] = [%(root)r, %(root)r + '/tests', %(root)r + '/oracle/pystubs', %(ref)r]
import numpy as np, torch
which = sys.argv[1]
if which == 'mine':
    from redner_amd import _capi
    _capi.load(%(lib)r)                       # test harness in place of the GPU library
    import redner_amd
    redner_amd.install()                      # `import redner` now resolves to redner_amd.redner
else:
    import oracle_util
    sys.modules['redner'] = oracle_util.load_oracle()
import redner, pyredner                        # the reference's package, unmodified
pyredner.set_use_gpu(False)
cam = pyredner.Camera(position=torch.tensor([0.0, 0.0, -5.0]), look_at=torch.tensor([0.0, 0.0, 0.0]),
                      up=torch.tensor([0.0, 1.0, 0.0]), fov=torch.tensor([45.0]), clip_near=1e-2, resolution=(32, 32))
mats = [pyredner.Material(diffuse_reflectance=torch.tensor([0.5, 0.5, 0.5]))]
tri = pyredner.Shape(vertices=torch.tensor([[-2.0, 1.5, 0.3], [0.9, 1.2, -0.3], [-0.4, -1.4, 0.2]], requires_grad=True),
                     indices=torch.tensor([[0, 1, 2]], dtype=torch.int32), uvs=None, normals=None, material_id=0)
light = pyredner.Shape(vertices=torch.tensor([[-1.0, -1.0, -7.0], [1.0, -1.0, -7.0], [-1.0, 1.0, -7.0], [1.0, 1.0, -7.0]]),
                       indices=torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32), uvs=None, normals=None, material_id=0)
scene = pyredner.Scene(cam, [tri, light], mats, [pyredner.AreaLight(shape_id=1, intensity=torch.tensor([20.0, 20.0, 20.0]))])
args = pyredner.RenderFunction.serialize_scene(scene=scene, num_samples=4, max_bounces=1,
                                               sampler_type=redner.SamplerType.sobol)
img = pyredner.RenderFunction.apply(1, *args)
img.sum().backward()
np.savez(sys.argv[2], image=img.detach().numpy(), grad=tri.vertices.grad.numpy())
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'pyredner')), reason='reference checkout not mounted')
def test_unmodified_pyredner_runs_on_our_redner(hostsim_backend, tmp_path):
    import numpy as np
    from conftest import HOSTSIM_LIB
    script = tmp_path / 'dropin.py'
    script.write_text(SCRIPT % {'root': ROOT, 'ref': REF, 'lib': HOSTSIM_LIB})
    outs = {}
    for which in ('mine', 'oracle'):
        out = str(tmp_path / (which + '.npz'))
        subprocess.check_call([sys.executable, str(script), which, out], stdout=subprocess.DEVNULL, timeout=600)
        outs[which] = np.load(out)
    a, b = outs['mine'], outs['oracle']
    assert np.array_equal(a['image'], b['image'])
    rel = np.linalg.norm(a['grad'].astype(np.float64) - b['grad']) / np.linalg.norm(b['grad'])
    assert rel < 1e-4, rel


# ---- the reference's own tests/test_bunny_box.py recipe: load_mitsuba -> serialize -> render -> backward ------------
BUNNY_SCRIPT = r'''
import os, sys
sys.path[:0] = [%(root)r, %(root)r + '/tests', %(root)r + '/oracle/pystubs', %(ref)r]
import numpy as np, torch
which = sys.argv[1]
if which == 'mine':
    from redner_amd import _capi
    _capi.load(%(lib)r)
    import redner_amd
    redner_amd.install()
else:
    import oracle_util
    sys.modules['redner'] = oracle_util.load_oracle()
import redner, pyredner
pyredner.set_use_gpu(False)
os.chdir(%(ref)r + '/tests')
scene = pyredner.load_mitsuba('scenes/bunny_box.xml')          # goes through redner.load_serialized
scene.shapes[-1].vertices.requires_grad = True
scene.camera.resolution = (24, 24)
args = pyredner.RenderFunction.serialize_scene(scene=scene, num_samples=2, max_bounces=4,
                                               sampler_type=redner.SamplerType.sobol)
img = pyredner.RenderFunction.apply(1, *args)
img.sum().backward()
np.savez(sys.argv[2], image=img.detach().numpy(), grad=scene.shapes[-1].vertices.grad.numpy())
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'pyredner')), reason='reference checkout not mounted')
def test_load_mitsuba_bunny_box_on_our_redner(hostsim_backend, tmp_path):
    import numpy as np
    from conftest import HOSTSIM_LIB
    script = tmp_path / 'dropin_bunny.py'
    script.write_text(BUNNY_SCRIPT % {'root': ROOT, 'ref': REF, 'lib': HOSTSIM_LIB})
    outs = {}
    for which in ('mine', 'oracle'):
        out = str(tmp_path / (which + '.npz'))
        subprocess.check_call([sys.executable, str(script), which, out], stdout=subprocess.DEVNULL, timeout=900)
        outs[which] = np.load(out)
    a, b = outs['mine'], outs['oracle']
    assert np.array_equal(a['image'], b['image'])
    rel = np.linalg.norm(a['grad'].astype(np.float64) - b['grad']) / np.linalg.norm(b['grad'])
    assert rel < 1e-4, rel


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'tests/scenes/teapot.serialized')), reason='reference checkout not mounted')
def test_load_serialized_equals_reference():
    """redner.load_serialized (zlib + numpy) vs the reference's C++ loader (src/load_serialized.cpp) on every sub-mesh."""
    import numpy as np
    import oracle_util
    from redner_amd import redner
    ref = oracle_util.load_oracle()
    for fn, n in (('bunny_box.serialized', 7), ('teapot.serialized', 6), ('teapot_specular.serialized', 7)):
        path = os.path.join(REF, 'tests/scenes', fn)
        for i in range(n):
            a, b = redner.load_serialized(path, i), ref.load_serialized(path, i)
            for name in ('vertices', 'indices', 'uvs', 'normals'):
                x, y = getattr(a, name), getattr(b, name)
                assert x.dtype == y.dtype and np.array_equal(x.reshape(-1), y.reshape(-1)), (fn, i, name)
        with pytest.raises(RuntimeError):
            redner.load_serialized(path, n)
