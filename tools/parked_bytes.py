import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes, redner_amd
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
dev = torch.device('cuda:0')
sc = scenes.bunny_box(dev, resolution=(1024, 1024))
args = RenderFunction.serialize_scene(sc, 32, 4, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner)
img = RenderFunction.apply(1, *args)
img.sum().backward()
torch.cuda.synchronize()
print('parked after 1024x1024x32 forward+backward: %.1f GB' % (redner_amd.trim_cache() / 1e9))
