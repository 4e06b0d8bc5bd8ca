"""Sample batches (render.cpp: BatchView): small frames render several Sobol' samples as ONE set of lanes -- lane = sample x
pixel -- in forward and in gradient renders of plain scenes.  What the reference decides per sample (the live-lane
compaction whose ranks number the secondary-edge sampler's slots, src/pathtracer.cpp:504-505; the dimension counters that
advance only for samples whose lists are not empty, :432-436, 590-706; the order in which a pixel's fp32 image sums take the
launches' contributions, :283,378) is kept per sample, so a batched render must equal the one-sample-at-a-time render:
the image bit for bit, the gradients bit for bit on the sequential CPU harness and to the order of fp64 atomics on the GPU.
RDR_BATCH=1 switches batching off (read once per process, hence the subprocesses).  A third run pretends that the device has
100 MB left (RDR_MEM_AVAILABLE_MB): the batches shrink to what fits (render.cpp) and nothing else changes; a fourth caps the
batches at three samples, so that a call is several batches and its last one is smaller than the buffers were made for."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# plain scenes (the lean kernels) and the general kernels: orthographic / lens-distorted cameras, two lights + separate uv /
# normal indices + viewport + pixel-centre samples, the dead-depth case is not batched (environment light)
CASES = ('bunny_box_32x32x4', 'two_triangles_64x64x16', 'bunny_box_96x96x8', 'two_triangles_ortho_64x64x4',
         'two_triangles_distorted_64x64x4', 'misc_features_40x56x4', 'misc_features_viewport_40x56x4',
         # mip-mapped textures / environment light: batched since round 4 (chain mode; stale hit positions replayed)
         'textured_sphere_gbuffer_48x48x4', 'envmap_sphere_48x48x4', 'living_room_standin_40x40x2',
         'textured_sphere_ids_radiance_last_48x48x3',
         # ... unless both edge estimators are off: then it is
         'living_room_standin_envmap_noedges_32x32x4', 'envmap_sphere_noedges_48x48x4', 'misc_features_noedges_40x56x4')

REPLAYED = ('envmap_sphere_48x48x4',)       # environment light AND edge sampling: see test_batches_equal_single_samples_hostsim

CODE = r'''
import sys
sys.path[:0] = [%(root)r, %(root)r + '/tests']
import numpy as np, torch
from redner_amd import _capi
_capi.load(%(lib)r)
from redner_amd import redner
from golden.make_golden import render_case, CASES
dev = torch.device(%(dev)r)
out = {}
for name in %(cases)r:
    for k, v in render_case(redner, *CASES[name], device=dev).items():
        out[name + '/' + k] = v
np.savez(sys.argv[1], **out)
'''


def _both(tmp_path, lib, dev):
    code = CODE % {'root': ROOT, 'lib': lib, 'dev': dev, 'cases': CASES}
    paths = []
    for tag, env in (('one', {'RDR_BATCH': '1'}), ('batched', {}), ('tight', {'RDR_MEM_AVAILABLE_MB': '100'}),
                     ('ragged', {'RDR_BATCH': '3'})):
        p = str(tmp_path / (tag + '.npz'))
        e = dict(os.environ, **env)
        if tag not in ('one', 'ragged'):
            e.pop('RDR_BATCH', None)
        if tag != 'tight':
            e.pop('RDR_MEM_AVAILABLE_MB', None)
        subprocess.check_call([sys.executable, '-c', code, p], env=e, timeout=900)
        paths.append(np.load(p))
    return paths


def test_batches_equal_single_samples_hostsim(hostsim_backend, tmp_path):
    from conftest import HOSTSIM_LIB
    one, batched, tight, ragged = _both(tmp_path, HOSTSIM_LIB, 'cpu')
    worst = 0.0
    for k in one.files:
        if k.split('/')[0] in REPLAYED and not k.endswith('/image'):
            # environment light + edge sampling: the stale hit-position reads across the samples of a batch are replayed after
            # the sweep (render.cpp: replay_stale_hits) -- the same terms, added in another order: fp64 rounding, no more
            for other in (batched, tight, ragged):
                a, b = one[k].astype(np.float64), other[k].astype(np.float64)
                err = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)
                worst = max(worst, err)
                assert err <= 1e-6, (k, err)                  # (the tensors are fp32: one ulp of a component is 6e-8)
            continue
        assert np.array_equal(one[k], batched[k]), k           # sequential harness: every tensor bit for bit
        assert np.array_equal(one[k], tight[k]), k
        assert np.array_equal(one[k], ragged[k]), k
    print('replayed cases: worst rel-L2 batched vs one sample per launch %.2e' % worst)


@pytest.mark.gpu
def test_batches_equal_single_samples_gpu(gpu_backend, tmp_path):
    from redner_amd import _capi
    one, batched, tight, ragged = _both(tmp_path, _capi.library_path(), 'cuda:0')
    for other in (batched, tight, ragged):
        for k in one.files:
            if k.endswith('/image'):
                assert np.array_equal(one[k], other[k]), k     # the image: fp32 sums in the reference's order
            else:
                a, b = one[k].astype(np.float64), other[k].astype(np.float64)
                n = np.linalg.norm(a)
                assert np.linalg.norm(a - b) <= 2e-6 * n + 1e-30, (k, np.linalg.norm(a - b) / max(n, 1e-300))
