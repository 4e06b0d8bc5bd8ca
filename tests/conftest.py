import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


# Two builds of the product library exist: libredner_amd.so (the device's own sin / cos / ...: what
# bench.py, smoke() and users run) and libredner_amd_exact.so (glibc-exact transcendental functions: sample-exact against the
# oracle also where such a function feeds a chaotic decision).  EVERY test that takes `gpu_backend` runs on BOTH (round 6;
# "the library that is timed is not the library that is fuzzed"); the fixture sets REDNER_AMD_LIBM for the
# subprocesses tests start.  parity_util.DEFAULT_BUILD_DRAWS_OTHER_SAMPLES lists, explicitly, the cases in which the default
# build may legitimately draw other (equally valid) edge samples, and what they are held to instead.
# RDR_TEST_LIBM=exact|default restricts a run to one build.
GPU_BUILDS = [b for b in ('exact', 'default') if os.environ.get('RDR_TEST_LIBM', b) == b]

# The fixtures are small frames; the benchmark is a large one.  Below 2^19 lanes per launch set the library keeps the one-launch
# hierarchical pick and un-compacted adjoint lists (fewer launches: render.cpp `large_forms`).  The test session asks for the
# LARGE-frame forms at every size, so that what the benchmark runs is what the fixtures, the fuzz legs and the harness check;
# tests/test_tuning.py (`small_frame_forms`, `pickh_one_launch`, `no_nee_compact`) covers the small-frame forms.
os.environ.setdefault('RDR_LARGE_FRAME_FORMS', '1')


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


@pytest.fixture(scope='session', params=GPU_BUILDS)
def gpu_backend(request):
    """The product on cuda:0 -- libredner_amd_exact.so, then libredner_amd.so (session-scoped parameter: pytest groups the tests
    by build).  Fails loudly if it is not what got loaded."""
    import torch
    from redner_amd import _capi
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    os.environ.pop('REDNER_AMD_LIB', None)
    if request.param == 'exact':
        os.environ['REDNER_AMD_LIBM'] = 'exact'
    else:
        os.environ.pop('REDNER_AMD_LIBM', None)
    _capi.load()
    assert _capi.is_product_library(), _capi.library_path()
    assert _capi.library_path() == (_capi.EXACT_LIBRARY if request.param == 'exact' else _capi.DEFAULT_LIBRARY)
    assert _capi.lib().rdr_libm_exact() == (1 if request.param == 'exact' else 0)
    from redner_amd import redner
    return redner
