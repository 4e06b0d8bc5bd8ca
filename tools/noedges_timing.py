"""Gradient render of the config-5 stand-ins WITHOUT the edge estimators (material / texture / light optimisation): Msamples/s
forward + backward at 1024 x 1024.  RDR_BATCH=1 = one sample per launch."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
dev = torch.device('cuda:0')
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name in ('living_room_standin', 'living_room_standin_envmap'):
    sc = getattr(scenes, name)(dev, resolution=(1024, 1024))
    for t in (sc.camera.position, sc.camera.look_at, sc.camera.up):
        t.requires_grad_(False)
    for m in sc.materials:
        m.diffuse_reflectance.mipmap[0].requires_grad_(True)
    ts = []
    for i in range(4):
        args = RenderFunction.serialize_scene(sc, spp, 6, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner,
                                              use_primary_edge_sampling=False, use_secondary_edge_sampling=False)
        torch.cuda.synchronize(); t0 = time.time()
        img = RenderFunction.apply(i, *args)
        img.sum().backward()
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    t = min(ts[1:])
    print('%-28s %d spp: %.1f ms  %.2f Msamples/s' % (name, spp, t * 1e3, 1024 * 1024 * spp / t / 1e6))
