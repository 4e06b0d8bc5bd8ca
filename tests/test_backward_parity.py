"""Gradient parity against the oracle: every gradient tensor of the reference's backward pass
(vertices via the continuous adjoint + primary/secondary edge sampling, light intensity,
diffuse reflectance, camera position) within 1e-4 relative L2 (BASELINE.json north_star) on
identical Sobol' sequences.  The oracle's own run-to-run noise (fp32 atomics) is ~4e-7."""
import os

import numpy as np
import pytest
import torch

from golden.make_golden import CASES, render_case
from oracle_util import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


def _check(backend, device, name):
    out = render_case(backend, *CASES[name], device=device)
    gold = np.load(os.path.join(GOLD, name + '.npz'))
    assert set(out.keys()) == set(gold.files)
    worst = 0.0
    for k in gold.files:
        g = torch.from_numpy(gold[k])
        mine = torch.from_numpy(out[k])
        assert torch.isfinite(mine).all(), k
        if float(g.double().norm()) == 0.0:
            assert float(mine.double().norm()) < 1e-12, k
            continue
        e = rel_l2(mine, g)
        worst = max(worst, e)
        assert e < TOL, (k, e)
    return worst


@pytest.mark.parametrize('name', list(CASES))
def test_backward_hostsim(hostsim_backend, name):
    _check(hostsim_backend, torch.device('cpu'), name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_backward_gpu(gpu_backend, name):
    _check(gpu_backend, torch.device('cuda:0'), name)
