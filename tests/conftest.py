import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


# The parity tests hold the GPU to the oracle sample for sample, also where a transcendental function feeds a chaotic decision
# (fisheye / panorama cameras + the hierarchical edge pick): they load the build whose kernels compute sin / cos / ... as glibc
# does (libredner_amd_exact.so; include/redner_amd.h: rdr_libm_exact).  Subprocesses started by tests inherit the choice.
# tests/test_default_library_gpu.py covers the default build (the device's own libm), which bench.py and smoke() run.
os.environ.setdefault('REDNER_AMD_LIBM', 'exact')


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
    _capi.load()          # redner_amd/lib/libredner_amd_exact.so (REDNER_AMD_LIBM=exact, above)
    assert _capi.is_product_library(), _capi.library_path()
    assert _capi.lib().rdr_libm_exact() == (1 if os.environ.get('REDNER_AMD_LIBM') == 'exact' else 0)
    from redner_amd import redner
    return redner
