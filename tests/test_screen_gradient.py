"""screen_gradient_image: the two-channel d(pixel)/d(screen position) image that
redner.render() fills from d_primary_intersection (src/primary_intersection.cpp:111-114) and the primary-edge
estimator (src/edge.cpp:765-773), as driven by RenderFunction.visualize_screen_gradient
(pyredner/render_pytorch.py:983-1048; the reference's tests/test_screen_gradient.py only saves pictures of it).
Compared with the oracle's fixtures, whole image, 1e-4 relative L2."""
import os

import numpy as np
import pytest
import torch

from golden.make_golden import SCREEN_GRADIENT_CASES, screen_gradient_case
from parity_util import GOLD, assert_parity, compare, record


def _check(backend, device, name, tag):
    out = screen_gradient_case(backend, *SCREEN_GRADIENT_CASES[name], device=device)
    gold = np.load(os.path.join(GOLD, name + '.npz'))
    assert float(np.abs(gold['screen_gradient']).sum()) > 0          # the comparison is not vacuous
    rep = compare(out, gold)
    record(name, rep, tag)
    assert_parity(rep, name)


@pytest.mark.parametrize('name', list(SCREEN_GRADIENT_CASES))
def test_screen_gradient_hostsim(hostsim_backend, name):
    _check(hostsim_backend, torch.device('cpu'), name, 'hostsim')


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(SCREEN_GRADIENT_CASES))
def test_screen_gradient_gpu(gpu_backend, name):
    _check(gpu_backend, torch.device('cuda:0'), name, 'gpu')
