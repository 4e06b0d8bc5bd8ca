"""Wall time of Scene construction INCLUDING its edge structures (RDR_SYNC_EDGES=1 joins the build inside the constructor) on the
52 k-triangle soup of tests/scenes.py: device-built hierarchies vs RDR_EDGE_HOST_BUILD=1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['RDR_SYNC_EDGES'] = '1'
import torch
from redner_amd import redner as rd
from redner_amd.render_pytorch import RenderFunction
import scenes
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'triangle_soup_large'
sc = getattr(scenes, name)(dev)
ts = []
for it in range(8):
    for s in sc.shapes:
        s.vertices = (s.vertices.detach() + 1e-4).requires_grad_(True)
    args = RenderFunction.serialize_scene(sc, 1, 2, sampler_type=rd.SamplerType.sobol, device=dev, backend=rd)
    torch.cuda.synchronize()
    t0 = time.time()
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    torch.cuda.synchronize()
    ts.append((time.time() - t0) * 1e3)
    del u
ts = sorted(ts[2:])
print('%s: Scene incl. edge structures  median %.2f ms  min %.2f ms  (%s)' % (name, ts[len(ts) // 2], ts[0], 'host hierarchies' if os.environ.get('RDR_EDGE_HOST_BUILD') else 'device hierarchies'))
