"""Wall time per optimisation-loop iteration at the sizes pyredner loops typically run (256x256, 4 spp): every
iteration builds the Scene, renders forward and backward -- as pyredner.RenderFunction does (render_pytorch.py:609)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from redner_amd import redner as rd
from redner_amd.render_pytorch import RenderFunction
import scenes

dev = torch.device('cuda:0')
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# SMALL_LOOP_SCENE=<builder of tests/scenes.py> (default bunny_box), e.g. envmap_sphere / living_room_standin_envmap
sc = getattr(scenes, os.environ.get('SMALL_LOOP_SCENE', 'bunny_box'))(dev, resolution=(res, res))
verts = [s.vertices for s in sc.shapes]
bounces = 6 if os.environ.get('SMALL_LOOP_SCENE', '').startswith('living_room') else 4
for v in verts:
    v.requires_grad_(True)
# `move`: the vertices change every iteration (a geometry optimisation: every Scene builds its edge structures);
# otherwise only what does not enter them could change (materials / lights: the structures are shared, scene.h)
move = len(sys.argv) > 3 and sys.argv[3] == 'move'
times = []
for it in range(12):
    if move:
        with torch.no_grad():
            for v in verts:
                v.add_(1e-4 * torch.sin(torch.arange(v.numel(), device=v.device, dtype=torch.float32) + it).reshape(v.shape))
    torch.cuda.synchronize()
    t0 = time.time()
    args = RenderFunction.serialize_scene(sc, spp, bounces, sampler_type=rd.SamplerType.sobol, device=dev, backend=rd)
    t1 = time.time()
    img = RenderFunction.apply(it + 1, *args)
    torch.cuda.synchronize()
    t2 = time.time()
    img.sum().backward()
    torch.cuda.synchronize()
    t3 = time.time()
    times.append((t1 - t0, t2 - t1, t3 - t2))
for name, k in (('serialize', 0), ('forward (Scene + render)', 1), ('backward', 2)):
    xs = sorted(t[k] for t in times[2:])
    print('%-26s median %.2f ms  min %.2f ms' % (name, xs[len(xs) // 2] * 1e3, xs[0] * 1e3))
tot = sorted(sum(t) for t in times[2:])
print('iteration%s         median %.2f ms  (%dx%d, %d spp: %.2f Msamples/s)' % (' (moving)' if move else '         ', tot[len(tot) // 2] * 1e3, res, res, spp, res * res * spp / tot[len(tot) // 2] / 1e6))
