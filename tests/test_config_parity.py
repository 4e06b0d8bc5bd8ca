"""Parity at the sizes BASELINE.json quotes, GPU vs the oracle's fixtures, whole tensors:

  config 2   two_triangles 256 x 256 x 64 spp, max_bounces 1          (tests/test_two_triangles.py:11-55,72-79)
  config 3   bunny_box 512 x 512, max_bounces 4 (tests/test_bunny_box.py:25-32), in the reduced form section 8d
             prescribes: the full frame at 8 spp, and a full-resolution 128 x 128 viewport tile at the full 128 spp

Forward image and every gradient tensor (bunny vertices incl. both edge estimators, light, materials, camera)."""
import os

import numpy as np
import pytest
import torch

from golden.make_golden import CONFIG_CASES, render_case
from parity_util import GOLD, assert_parity, compare, record


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CONFIG_CASES))
def test_config_size_gpu(gpu_backend, name):
    out = render_case(gpu_backend, *CONFIG_CASES[name], device=torch.device('cuda:0'))
    rep = compare(out, np.load(os.path.join(GOLD, name + '.npz')))
    record(name, rep, 'gpu')
    assert_parity(rep, name)


def test_config_fixtures_present():
    for name in CONFIG_CASES:
        z = np.load(os.path.join(GOLD, name + '.npz'))
        assert 'image' in z.files and any(k.startswith('grad_') for k in z.files)
