"""Process exit while the edge builder thread may still be inside the HIP runtime (scene.cpp: EdgeBuilder::drain): must exit cleanly."""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from redner_amd import redner as rd
from redner_amd.render_pytorch import RenderFunction
import scenes
dev = torch.device('cuda:0')
sc = scenes.bunny_box(dev, resolution=(64, 64))
args = RenderFunction.serialize_scene(sc, 1, 2, sampler_type=rd.SamplerType.sobol, device=dev, backend=rd)
u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
import builtins
builtins._leak = u            # never destroyed explicitly
print('exiting with a build in flight')
