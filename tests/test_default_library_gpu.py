"""The DEFAULT build of the library (libredner_amd.so: the device's own sin / cos / ..., what bench.py and smoke() run) against
the oracle's fixtures.  The rest of the suite loads the glibc-exact build (tests/conftest.py); this file starts a process without
that choice and holds the default build to the same bars -- image bit for bit, every gradient tensor to 1e-4 -- on the cases where
no transcendental function feeds a chaotic decision: perspective / orthographic cameras, i.e. every BASELINE configuration.
Under an environment map the radiance itself goes through atan2 / acos (direction -> texel, src/envmap.h), so those images agree
to the last bit of those functions instead: <= 1e-6 relative L2.
(Fisheye / panorama cameras with secondary edge sampling draw other, equally valid edge samples with this build:
tests/test_statistical_parity.py is the check that applies to them.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys
sys.path[:0] = [%(root)r, %(root)r + '/tests']
import numpy as np, torch
from redner_amd import _capi
_capi.load()
assert _capi.library_path() == _capi.DEFAULT_LIBRARY and _capi.lib().rdr_libm_exact() == 0
from redner_amd import redner
from golden.make_golden import CASES, CONFIG_CASES, render_case
from parity_util import GOLD, compare, summary
rep = {}
for name in sys.argv[1:]:
    case = CASES.get(name) or CONFIG_CASES[name]
    out = render_case(redner, *case, device=torch.device('cuda:0'))
    gold = np.load(os.path.join(GOLD, name + '.npz'))
    r = compare(out, gold, name)
    d = np.linalg.norm(out['image'].astype(np.float64) - gold['image']) / np.linalg.norm(gold['image'].astype(np.float64))
    rep[name] = {'image_identical': bool(np.array_equal(out['image'], gold['image'])), 'image_rel_l2': float(d), 'worst': summary(r)['worst_rel_l2'],
                 'failed': [k for k, e in r.items() if not e.get('zero_reference') and not e['rel_l2'] < e['tol']]}
print('REPORT ' + json.dumps(rep))
'''

CASES_DEFAULT = ['single_triangle_64x64x4', 'two_triangles_64x64x16', 'bunny_box_96x96x8', 'two_triangles_ortho_64x64x4',
                 'textured_sphere_gbuffer_48x48x4', 'envmap_sphere_48x48x4', 'living_room_standin_40x40x2',
                 'glossy_floor_blocker_48x48x4', 'bunny_box_512x512x8']


@pytest.mark.gpu
def test_default_build_against_the_fixtures(tmp_path):
    script = tmp_path / 'default_lib.py'
    script.write_text(SCRIPT % {'root': ROOT})
    env = {k: v for k, v in os.environ.items() if k not in ('REDNER_AMD_LIBM', 'REDNER_AMD_LIB')}
    r = subprocess.run([sys.executable, str(script)] + CASES_DEFAULT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('REPORT ')][-1][7:])
    assert set(rep) == set(CASES_DEFAULT)
    for name, e in rep.items():
        if 'envmap' in name:
            assert e['image_rel_l2'] < 1e-6, (name, e)
        else:
            assert e['image_identical'], (name, e)
        assert not e['failed'] and e['worst'] < 1e-4, (name, e)


@pytest.mark.gpu
def test_both_builds_load_side_by_side():
    """The default build next to the exact one this process renders with: both are in-tree products of one source
    (include/redner_amd.h: rdr_libm_exact tells them apart)."""
    import ctypes
    from redner_amd import _capi
    assert os.path.exists(_capi.DEFAULT_LIBRARY) and os.path.exists(_capi.EXACT_LIBRARY)
    fast, exact = ctypes.CDLL(_capi.DEFAULT_LIBRARY), ctypes.CDLL(_capi.EXACT_LIBRARY)
    assert fast.rdr_libm_exact() == 0 and exact.rdr_libm_exact() == 1
    ch = (ctypes.c_int * 2)(0, 3)              # radiance + position
    assert fast.rdr_compute_num_channels(ch, 2, 0) == exact.rdr_compute_num_channels(ch, 2, 0) == 6
