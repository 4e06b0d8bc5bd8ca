"""A render beside a process-wide allocation that leaves ~30 GB of device memory: the sample batches must shrink instead of
the call failing (render.cpp: batch policy).  RDR_DEBUG_BATCH=1 prints the batch sizes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
dev = torch.device('cuda:0')
free, total = torch.cuda.mem_get_info(dev)
hold = torch.empty(int(free - 30e9), dtype=torch.uint8, device=dev)
print('holding %.0f GB of %.0f GB' % (hold.numel() / 1e9, total / 1e9))
sc = scenes.bunny_box(dev, resolution=(1024, 1024))
args = RenderFunction.serialize_scene(sc, 32, 4, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner)
img = RenderFunction.apply(1, *args)
img.sum().backward()
torch.cuda.synchronize()
print('ok', float(img.detach().mean()), sum(float(s.vertices.grad.abs().sum()) for s in sc.shapes if s.vertices.grad is not None))
