"""GPU diagnostic: what differs between the PyTorch and the TensorFlow (stand-in) surface on the environment-lit test case."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'tf_standin')]
import numpy as np, torch
import scenes
from redner_amd import _capi
_capi.load()
from redner_amd import redner as rd
from redner_amd.render_pytorch import RenderFunction
import redner_amd.render_tensorflow as rt
import test_tensorflow_frontend as T
dev = torch.device('cuda:0')
sc = scenes.envmap_sphere(dev, resolution=(24, 24))
for t in T._leaves(sc):
    t.requires_grad_(True)
rt.set_use_gpu(True)
tsc, pairs = T.tf_scene_from(sc, rt)
e, te = sc.envmap, tsc.envmap
for n in ('sample_cdf_ys', 'sample_cdf_xs', 'world_to_env', 'env_to_world'):
    a = getattr(e, n).detach().cpu().numpy(); b = getattr(te, n).numpy()
    print(n, a.shape, 'devices', getattr(e, n).device, getattr(te, n)._t.device, 'max abs diff', float(np.abs(a - b).max()), 'n differ', int((a != b).sum()))
print('pdf_norm', e.pdf_norm, te.pdf_norm, e.pdf_norm == te.pdf_norm)
for k, (l, tl) in enumerate(zip(e.values.mipmap, te.values.mipmap)):
    print('level', k, 'differ', int((l.detach().cpu().numpy() != tl.numpy()).sum()))
kw = dict(sampler_type=rd.SamplerType.sobol)
args = RenderFunction.serialize_scene(sc, 2, 2, device=dev, backend=rd, **kw)
targs = rt.serialize_scene(tsc, 2, 2, backend=rd, **kw)
img = RenderFunction.apply(7, *args).detach().cpu().numpy()
img2 = RenderFunction.apply(7, *args).detach().cpu().numpy()
timg = rt.render(7, *targs).numpy()
print('torch surface twice identical:', np.array_equal(img, img2))
d = np.abs(img - timg)
print('image: pixels differing', int((d > 0).sum()), 'of', d.size, 'max abs', float(d.max()), 'max rel', float((d / np.maximum(np.abs(img), 1e-6)).max()))
m, tm = args[0], targs[0]
for k in sorted(m):
    if k not in tm:
        print('meta key only in torch', k); continue
    if isinstance(m[k], dict):
        for kk in m[k]:
            if m[k][kk] != tm[k].get(kk) and not isinstance(m[k][kk], int):
                print('meta', k, kk, m[k][kk], tm[k].get(kk))
print('num tensors', len(args) - 1, len(targs) - 1)
for i, (a, b) in enumerate(zip(args[1:], targs[1:])):
    a = a.detach().cpu().numpy(); b = b._t.detach().cpu().numpy() if hasattr(b, '_t') else np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b):
        print('tensor', i, a.shape, b.shape, 'differs', float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.shape == b.shape else '')
