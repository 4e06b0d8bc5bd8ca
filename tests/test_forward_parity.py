"""Forward image parity against the oracle (golden fixtures made by tests/golden/make_golden.py
from the reference's own C++ core).  Tolerance: 1e-4 relative L2 (BASELINE.json north_star); the
fp64 pipeline actually lands at ~1e-7 or exact."""
import os

import numpy as np
import pytest
import torch

import scenes
from golden.make_golden import CASES as GOLDEN_CASES
from oracle_util import rel_l2
from redner_amd.render_pytorch import RenderFunction

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = [(name,) + tuple(case) for name, case in GOLDEN_CASES.items()]
TOL = 1e-4


def _render(backend, device, builder, res, spp, mb, channels=None, opts=None):
    sc = getattr(scenes, builder)(device, resolution=res if isinstance(res, tuple) else (res, res))
    ch = None if channels is None else [getattr(backend.channels, c) for c in channels]
    opts = dict(opts or {})
    sampler = getattr(backend.SamplerType, opts.pop('sampler', 'sobol'))
    args = RenderFunction.serialize_scene(sc, spp, mb, channels=ch, sampler_type=sampler,
                                          device=device, backend=backend, **opts)
    with torch.no_grad():
        return RenderFunction.apply(1, *args)


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_forward_hostsim(hostsim_backend, case):
    img = _render(hostsim_backend, torch.device('cpu'), *case[1:])
    name = case[0]
    gold = torch.from_numpy(np.load(os.path.join(GOLD, name + '.npz'))['image'])
    assert rel_l2(img, gold) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_forward_gpu(gpu_backend, case):
    img = _render(gpu_backend, torch.device('cuda:0'), *case[1:])
    name = case[0]
    gold = torch.from_numpy(np.load(os.path.join(GOLD, name + '.npz'))['image'])
    assert torch.isfinite(img).all()
    assert rel_l2(img, gold) < TOL
