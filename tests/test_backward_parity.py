"""Gradient parity against the oracle: every gradient tensor of the reference's backward pass
(vertices via the continuous adjoint + primary/secondary edge sampling, light intensity,
diffuse reflectance, camera) within 1e-4 relative L2 (BASELINE.json north_star) on identical
random sequences -- whole tensors, no masks.  The oracle's own run-to-run noise (fp32 atomics) is ~4e-7.

The config-size cases (BASELINE configs 2 and 3) live in tests/test_config_parity.py."""
import os

import numpy as np
import pytest
import torch

from golden.make_golden import CASES, SAMPLE_EXACT_ON_CPU_ONLY, render_case
from parity_util import GOLD, assert_parity, compare, record


def _check(backend, device, name, tag):
    out = render_case(backend, *CASES[name], device=device)
    rep = compare(out, np.load(os.path.join(GOLD, name + '.npz')))
    record(name, rep, tag)
    assert_parity(rep, name)


@pytest.mark.parametrize('name', list(CASES))
def test_backward_hostsim(hostsim_backend, name):
    _check(hostsim_backend, torch.device('cpu'), name, 'hostsim')


@pytest.mark.gpu
@pytest.mark.parametrize('name', [c for c in CASES if c not in SAMPLE_EXACT_ON_CPU_ONLY])
def test_backward_gpu(gpu_backend, name):
    """Includes bunny_box_96x96x8: big enough that side streams, the second sample worker and the wave-summed
    gradient scatters are all active, compared with the oracle's fixture."""
    _check(gpu_backend, torch.device('cuda:0'), name, 'gpu')
