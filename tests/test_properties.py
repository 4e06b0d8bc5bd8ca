"""Size-independent properties of the gradient render, checked at small sizes on the CPU harness and at the sizes BASELINE.json
quotes on the GPU (where the oracle would need hours): the backward pass is LINEAR in the upstream image gradient -- the
reference multiplies every path / edge contribution by d_image and nothing else looks at it (src/path_contribution.cpp:156-626,
src/edge.cpp:447-450, 1396-1420) -- so  backward(a g1 + b g2) = a backward(g1) + b backward(g2)  on the same Sobol' points,
backward(0) = 0 exactly, and two calls with the same seed give the same image bit for bit."""
import numpy as np
import pytest
import torch

import scenes
from redner_amd.render_pytorch import RenderFunction


def _inputs(sc):
    ts = [s.vertices for s in sc.shapes if s.vertices.requires_grad]
    for m in sc.materials:
        t = m.diffuse_reflectance.mipmap[0] if hasattr(m.diffuse_reflectance, 'mipmap') else m.diffuse_reflectance
        if isinstance(t, torch.Tensor) and t.requires_grad:
            ts.append(t)
    ts += [t for t in (sc.camera.position, sc.camera.look_at, sc.camera.up) if isinstance(t, torch.Tensor) and t.requires_grad]
    return ts


def _linearity(backend, device, build, res, spp, mb, tol):
    sc = build(device, resolution=(res, res))
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=backend.SamplerType.sobol, device=device, backend=backend)
    img = RenderFunction.apply(5, *args)
    again = RenderFunction.apply(5, *RenderFunction.serialize_scene(sc, spp, mb, sampler_type=backend.SamplerType.sobol,
                                                                    device=device, backend=backend))
    assert torch.equal(img.detach(), again.detach())
    ins = _inputs(sc)
    assert ins
    gen = torch.Generator().manual_seed(7)
    g1 = torch.rand(img.shape, generator=gen).to(device)
    g2 = (torch.rand(img.shape, generator=gen) - 0.5).to(device)
    a, b = 2.5, -0.75

    def back(g):
        return [t.detach().cpu().double().numpy() for t in torch.autograd.grad(img, ins, grad_outputs=g, retain_graph=True)]

    r1, r2, r12, r0 = back(g1), back(g2), back(a * g1 + b * g2), back(torch.zeros_like(img))
    for x, y, z, zero in zip(r1, r2, r12, r0):
        assert not zero.any()
        want = a * x + b * y
        scale = a * np.linalg.norm(x) + abs(b) * np.linalg.norm(y)
        assert scale > 0
        assert np.linalg.norm(z - want) <= tol * scale, np.linalg.norm(z - want) / scale


CASES = {'two_triangles': (scenes.two_triangles, 1), 'bunny_box': (scenes.bunny_box, 4),
         'living_room_standin': (scenes.living_room_standin, 6)}


@pytest.mark.parametrize('name,res,spp', [('two_triangles', 48, 4), ('bunny_box', 24, 2), ('living_room_standin', 20, 2)])
def test_backward_is_linear_hostsim(hostsim_backend, name, res, spp):
    build, mb = CASES[name]
    # fp32 rounding of a g1 + b g2 and of the gradient tensors themselves: a few 1e-7 of the scale
    _linearity(hostsim_backend, torch.device('cpu'), build, res, spp, mb, 2e-6)


# config 2 at its size, config 3's frame, config 4's frame, the config-5 stand-in's frame
@pytest.mark.gpu
@pytest.mark.parametrize('name,res,spp', [('two_triangles', 256, 16), ('bunny_box', 512, 4), ('bunny_box', 1024, 2),
                                          ('living_room_standin', 1024, 1)])
def test_backward_is_linear_gpu(gpu_backend, name, res, spp):
    build, mb = CASES[name]
    _linearity(gpu_backend, torch.device('cuda:0'), build, res, spp, mb, 2e-6)
