"""Wall time of the Scene-build phases (RDR_DEBUG_DUMP=<dir> makes the library print them): bunny_box, vertices moved every iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from redner_amd import _capi
_capi.load()
from redner_amd import redner as rd
from redner_amd.render_pytorch import RenderFunction
import scenes
dev = torch.device('cuda:0')
sc = scenes.bunny_box(dev, resolution=(32, 32))
for it in range(4):
    for s in sc.shapes: s.vertices = s.vertices + 1e-4 * torch.sin(torch.arange(s.vertices.numel(), device=dev, dtype=torch.float32) + it).reshape(s.vertices.shape)
    args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=rd.SamplerType.sobol, device=dev, backend=rd)
    sys.stderr.write('--- iteration %d\n' % it)
    t0 = time.time()
    img = RenderFunction.apply(it + 1, *args)
    sys.stderr.write('forward total %.2f ms\n' % ((time.time() - t0) * 1e3))
