"""Pin the oracle: the reference's own C++ known-answer / finite-difference tests
(redner.test_*, src/redner.cpp:260-271; tol 1e-3, src/test_utils.h) must pass on the oracle
build, i.e. with our Embree stand-in behind test_scene_intersect."""
import pytest

import oracle_util

KATS = ['test_sample_primary_rays', 'test_scene_intersect', 'test_sample_point_on_light', 'test_active_pixels',
        'test_camera_derivatives', 'test_d_bsdf', 'test_d_bsdf_sample', 'test_d_bsdf_pdf', 'test_d_intersect',
        'test_d_sample_shape', 'test_atomic', 'test_camera_distortion']


@pytest.mark.skipif(not oracle_util.oracle_available(), reason='oracle/_ref not built')
@pytest.mark.parametrize('name', KATS)
def test_reference_kat(name):
    ref = oracle_util.load_oracle()
    fn = getattr(ref, name)
    try:
        fn(False)        # use_gpu = False (the failing path calls exit(1))
    except TypeError:
        fn()
