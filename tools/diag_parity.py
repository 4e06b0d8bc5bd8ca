#!/usr/bin/env python3
"""Diagnostic (not a test): render one scene with the GPU product and with the oracle on the same
box and print per-tensor relative L2, with the edge estimators switched on/off independently.
usage: python tools/diag_parity.py <scene> <res> <spp> <max_bounces>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402
from redner_amd import redner  # noqa: E402
from redner_amd.render_pytorch import RenderFunction  # noqa: E402
import scenes  # noqa: E402
import oracle_util  # noqa: E402

name, res, spp, mb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ref = oracle_util.load_oracle()


def run(be, dev, pe, se):
    sc = getattr(scenes, name)(dev, resolution=(res, res))
    for l in sc.area_lights:
        l.intensity.requires_grad_(True)
    if sc.camera.position is not None:
        sc.camera.position.requires_grad_(True)
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=be.SamplerType.sobol, device=dev, backend=be,
                                          use_primary_edge_sampling=pe, use_secondary_edge_sampling=se)
    img = RenderFunction.apply(1, *args)
    img.sum().backward()
    g = {'image': img.detach().cpu()}
    for i, sh in enumerate(sc.shapes):
        if sh.vertices.grad is not None:
            g['shape%d.vertices' % i] = sh.vertices.grad.cpu()
    for i, l in enumerate(sc.area_lights):
        g['light%d.intensity' % i] = l.intensity.grad.cpu()
    if sc.camera.position is not None:
        g['cam.position'] = sc.camera.position.grad.cpu()
    return g


for pe, se in ((False, False), (True, False), (False, True), (True, True)):
    a = run(redner, torch.device('cuda:0'), pe, se)
    b = run(ref, torch.device('cpu'), pe, se)
    print('primary_edges=%d secondary_edges=%d' % (pe, se))
    for k in b:
        print('   %-20s rel_l2 %.3e  |ref| %.4e' % (k, oracle_util.rel_l2(a[k], b[k]), float(b[k].double().norm())))
