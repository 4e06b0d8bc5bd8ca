"""Wall time of the scene-build phases on the benchmark scene (run on the GPU box):
RDR_DEBUG_DUMP=1 python tools/scene_build_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from redner_amd import redner as rd
from redner_amd.render_pytorch import RenderFunction
import scenes

dev = torch.device('cuda:0')
sc = scenes.bunny_box(dev, resolution=(1024, 1024))
args = RenderFunction.serialize_scene(sc, 4, 4, sampler_type=rd.SamplerType.sobol, device=dev, backend=rd)
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    torch.cuda.synchronize()
    print('unpack_args + Scene: %.2f ms' % ((time.time() - t0) * 1e3), file=sys.stderr)
