import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


HOSTSIM_LIB = os.path.join(ROOT, 'tests', 'hostsim', '_build', 'libredner_hostsim.so')


@pytest.fixture(scope='session')
def hostsim_backend():
    """The product's stage bodies + host driver compiled for the CPU debugging harness
    (tests/hostsim).  Test infrastructure: lets the host logic be checked without a GPU."""
    import subprocess
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'tests', 'hostsim'), '-j8'],
                          stdout=subprocess.DEVNULL)
    from redner_amd import _capi
    _capi.load(HOSTSIM_LIB)
    from redner_amd import redner
    yield redner


@pytest.fixture(scope='session')
def gpu_backend():
    """The product: libredner_amd.so on cuda:0.  Fails loudly if it is not what got loaded."""
    import torch
    from redner_amd import _capi
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    _capi.load()          # default = redner_amd/lib/libredner_amd.so
    assert _capi.library_path().endswith('libredner_amd.so')
    from redner_amd import redner
    return redner
