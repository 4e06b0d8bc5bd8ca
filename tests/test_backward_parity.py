"""Gradient parity against the oracle: every gradient tensor of the reference's backward pass
(vertices via the continuous adjoint + primary/secondary edge sampling, light intensity,
diffuse reflectance, camera) within 1e-4 relative L2 (BASELINE.json north_star) on identical
random sequences -- whole tensors, no masks.  The oracle's own run-to-run noise (fp32 atomics) is ~4e-7.

The config-size cases (BASELINE configs 2 and 3) live in tests/test_config_parity.py."""
import os

import numpy as np
import pytest
import torch

from golden.make_golden import CASES, render_case
from parity_util import DEFAULT_BUILD_DRAWS_OTHER_SAMPLES, GOLD, assert_parity, compare, libm_exact, record


def _check(backend, device, name, tag):
    out = render_case(backend, *CASES[name], device=device)
    rep = compare(out, np.load(os.path.join(GOLD, name + '.npz')))
    if device.type == 'cuda' and not libm_exact():
        tag += '-default-build'
        if name in DEFAULT_BUILD_DRAWS_OTHER_SAMPLES:
            # the device's own libm: other, equally valid edge samples (parity_util.py lists the cases).  The forward image
            # involves no chaotic decision and is held as everywhere; the gradients are held to the statistical test.
            record(name, rep, tag + '-image-only')
            assert rep['image']['rel_l2'] < 1e-6, (name, rep['image'])
            return
    record(name, rep, tag)
    assert_parity(rep, name)


@pytest.mark.parametrize('name', list(CASES))
def test_backward_hostsim(hostsim_backend, name):
    _check(hostsim_backend, torch.device('cpu'), name, 'hostsim')


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_backward_gpu(gpu_backend, name):
    """Includes bunny_box_96x96x8: big enough that side streams, the second sample worker and the wave-summed
    gradient scatters are all active, compared with the oracle's fixture -- and the fisheye / panorama cameras WITH secondary
    edge sampling (make_golden.CHAOTIC_PICK_CASES): sample-exact on the GPU since its sin / cos / atan2 are glibc's."""
    _check(gpu_backend, torch.device('cuda:0'), name, 'gpu')


def _gather_overflow(budget, caps, lib, dev, cases):
    """The NEE-mode gather hands work over in three ways -- subtrees to SecEdgeGatherSub when a lane's pop budget is spent,
    candidates to the 256-entry lists when a slot has more than 8, and the slot to the reference-order walk when even those
    lists (or the heavy / work registries) are full.  Tiny budgets and registry sizes force every one of them; the picks
    must not change (bit-identical fixtures).  Run in a subprocess: the switches are read once per process."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import os, sys
sys.path[:0] = [%r, %r + '/tests']
import numpy as np, torch
from redner_amd import _capi
_capi.load(%r)
from redner_amd import redner
from golden.make_golden import CASES, render_case
from parity_util import GOLD, assert_parity, compare
for name in %r:
    rep = compare(render_case(redner, *CASES[name], device=torch.device(%r)), np.load(os.path.join(GOLD, name + '.npz')))
    assert_parity(rep, name)
    assert sum(e['flipped_rows'] for e in rep.values()) == 0, name
''' % (ROOT, ROOT, lib, cases, dev)
    env = dict(os.environ, RDR_GATHER_BUDGET=budget)
    if caps is not None:
        env['RDR_GATHER_CAPS'] = caps
    subprocess.check_call([sys.executable, '-c', code], env=env, timeout=900)


@pytest.mark.parametrize('budget,caps', [('4', None), ('2', '3,5'), ('1', '0,0'), ('6', '100000,2')])
def test_gather_overflow_paths_hostsim(budget, caps, tmp_path):
    from conftest import HOSTSIM_LIB
    _gather_overflow(budget, caps, HOSTSIM_LIB, 'cpu', ('bunny_box_32x32x4', 'bunny_box_96x96x8', 'envmap_sphere_48x48x4'))


@pytest.mark.gpu
@pytest.mark.parametrize('budget,caps', [('2', '3,5'), ('1', '0,0')])
def test_gather_overflow_paths_gpu(gpu_backend, budget, caps, tmp_path):
    """The same hand-over paths on the GPU build (subtree hand-off across lanes, the 256-entry lists, the fallback walk)."""
    from redner_amd import _capi
    _gather_overflow(budget, caps, _capi.library_path(), 'cuda:0', ('bunny_box_32x32x4', 'bunny_box_96x96x8'))
