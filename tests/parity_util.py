"""Shared comparison of a rendered case against the oracle's fixture: whole-tensor relative L2 per tensor, no masks.

`flipped_rows` is reported (not used to excuse anything): the number of vertex rows whose error stands out of the
fp32-atomics noise floor.  A Monte-Carlo decision that differs from the oracle (one edge sample landing on another
edge) changes <= 4 rows by an O(1/N) amount; 0 means the two runs drew the same samples everywhere."""
import json
import os

import numpy as np
import torch

from oracle_util import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4           # BASELINE.json north_star: 1e-4 relative L2, forward image and every gradient tensor


# ---- what the DEFAULT build (libredner_amd.so: the device's own sin / cos / atan2 / acos / pow) is held to -----------------------
# Every GPU test runs on both builds (tests/conftest.py: gpu_backend).  The default build meets the same bars as the exact one --
# image bit for bit, every gradient tensor to 1e-4 -- EXCEPT where a transcendental function feeds the chaotic hierarchical edge
# pick (src/edge.cpp:1160-1230: one random number through ~100 rescalings): ocml and glibc differ in the last bit of a few per
# cent of results, the pick lands on another edge, and the gradient is another, equally valid, draw of the same estimator.
# The complete list of what may differ, and what holds instead:
DEFAULT_BUILD_DRAWS_OTHER_SAMPLES = {
    # fixture: why -> on the default build the forward image is still held to 1e-6 and the gradients to the statistical test
    # (tests/test_statistical_parity.py: 24 seeds, paired z < 5 against the oracle's per-seed functionals)
    'bunny_box_fisheye_32x32x4': 'fisheye primary rays: sin / cos / atan2 decide the first-hit position the secondary-edge pick starts from',
    'bunny_box_panorama_32x32x4': 'panorama primary rays: sin / cos decide the first-hit position the secondary-edge pick starts from',
}
# random scenes against the live oracle (tests/test_fuzz_parity.py): glossy bounces go through pow / sin / cos; scenes per leg
# (of ~107) in which ONE OR TWO edge samples may land on another edge (<= 4 vertex rows move, everything else agrees: _edge_flip)
DEFAULT_BUILD_FUZZ_FLIPS_PER_LEG = 2        # the exact build: 0


def libm_exact():
    """Is the loaded library the glibc-exact build?"""
    from redner_amd import _capi
    return bool(_capi.lib().rdr_libm_exact())


ACCUMULATOR_ELEMS = 16   # tensors this small (light intensity, constant reflectance, camera) collect millions of fp32 atomics


def compare(out, gold, name=None):
    """-> {tensor: {'rel_l2': e, 'tol': 1e-4, 'flipped_rows': k, ...}}; asserts keys match and values are finite.

    Every tensor is held to 1e-4, no exception.  Where the fixture carries `ref64_<tensor>` -- the oracle's own estimator on the
    same samples with the error of its fp32 atomics taken out (sum of 64 pixel-striped oracle passes in fp64,
    make_golden.add_ref64) -- THAT is the value compared with, and the distance to the oracle's single fp32 pass is only
    reported (`single_pass_rel_l2`, next to `oracle_selfdiff`: how far the single pass moves under an equivalent evaluation
    order, and `ref64_convergence`: K = 16 vs K = 64 stripes).

    `name`: the case.  Where tests/golden/<name>_ref_order.npz exists (make_ref_order.py), a camera tensor is compared with
    `harness64_<tensor>`: the fp64 sum of the SAME addends whose fp32 sum in the reference's own order (`harness32_`) equals the
    reference run on one thread (`oracle1t_`) -- tests/test_accumulation_order.py holds the fixture to that -- i.e. the oracle's
    value with its accumulation error, and nothing else, removed.  (Round 4 capped the bar of one tensor, the camera position of
    bunny_box 512 x 512 x 8, at 5e-4 because even 256-stripe sums of the oracle had not settled; that cap is gone.)"""
    gold = {k: gold[k] for k in (gold.files if hasattr(gold, 'files') else gold)}
    prefixes = ('selfdiff_', 'ref64_', 'ref64conv_', 'ref256_', 'ref256conv_', 'ref16_')
    aux = {p: {k[len(p):]: v for k, v in gold.items() if k.startswith(p)} for p in prefixes}
    gold = {k: v for k, v in gold.items() if not k.startswith(prefixes)}
    for k, v in aux['ref256_'].items():              # a fixture with a finer striped sum: that one is the value compared with
        aux['ref64_'][k] = v
        aux['ref64conv_'][k] = aux['ref256conv_'][k]
    assert set(out.keys()) == set(gold.keys()), (sorted(out.keys()), sorted(gold.keys()))
    ref_order = {}
    if name and os.path.exists(os.path.join(GOLD, name + '_ref_order.npz')):
        z = np.load(os.path.join(GOLD, name + '_ref_order.npz'))
        ref_order = {k[len('harness64_'):]: z[k].astype(np.float64) for k in z.files if k.startswith('harness64_')}
    rep = {}
    for k, gv in gold.items():
        g, mine = torch.from_numpy(np.asarray(gv)), torch.from_numpy(np.asarray(out[k]))
        assert torch.isfinite(mine).all(), k
        gn = float(g.double().norm())
        if gn == 0.0:
            rep[k] = {'rel_l2': float(mine.double().norm()), 'flipped_rows': 0, 'zero_reference': True}
            continue
        entry = {'rel_l2': rel_l2(mine, g), 'tol': TOL, 'flipped_rows': 0}
        if k in aux['ref64_']:
            entry['single_pass_rel_l2'] = entry['rel_l2']
            entry['rel_l2'] = rel_l2(mine, torch.from_numpy(np.asarray(aux['ref64_'][k])))
            entry['against'] = 'ref256' if k in aux['ref256_'] else 'ref64'
            entry['ref64_convergence'] = float(aux['ref64conv_'][k])       # between the two finest striped sums of the fixture
            if entry['ref64_convergence'] > TOL:
                entry['oracle_not_converged'] = True          # reported; the bar stays 1e-4
        if k in ref_order:
            entry['rel_l2_vs_striped_sum'] = entry['rel_l2']
            entry['rel_l2'] = rel_l2(mine, torch.from_numpy(ref_order[k]))
            entry['against'] = 'harness64 (reference-order accumulation proof, tests/test_accumulation_order.py)'
        if k in aux['selfdiff_']:
            entry['oracle_selfdiff'] = float(aux['selfdiff_'][k])
        if k in aux['ref16_']:
            # a large tensor with a striped oracle sum (make_ref_rows.py): reported next to the single pass; rows are judged
            # against it (the fp32 accumulation error of a hot vertex is not a different sample)
            g16 = torch.from_numpy(np.asarray(aux['ref16_'][k]))
            entry['rel_l2_vs_ref16'] = rel_l2(mine, g16)
            entry['single_pass_vs_ref16'] = rel_l2(g, g16)
            g = g16
        if k.endswith('_vertices') and g.dim() == 2 and g.shape[0] > ACCUMULATOR_ELEMS:
            row_err = (mine.double() - g.double()).norm(dim=1)
            entry['flipped_rows'] = int((row_err > 1e-5 * gn).sum())
        rep[k] = entry
    return rep


def assert_parity(rep, name=''):
    """Whole-tensor bar of 1e-4, every tensor, no row dropping."""
    for k, e in rep.items():
        if e.get('zero_reference'):
            assert e['rel_l2'] < 1e-12, (name, k, e)
        else:
            assert e['rel_l2'] < e['tol'], (name, k, e)
            # a tensor judged against the harness' fp64 sum (a value this project generated, pinned to the oracle by
            # tests/test_accumulation_order.py) must ALSO stay near the oracle's own striped sum -- looser (that sum has not
            # settled: 4.8e-4 at K = 256), but a bar that does not rest on self-generated data
            if 'rel_l2_vs_striped_sum' in e:
                assert e['rel_l2_vs_striped_sum'] < 1e-3, (name, k, e)


def summary(rep):
    worst = max((e['rel_l2'] for e in rep.values() if not e.get('zero_reference')), default=0.0)
    return {'worst_rel_l2': worst, 'flipped_rows': sum(e['flipped_rows'] for e in rep.values())}


def record(name, rep, backend_tag):
    """Append the per-tensor numbers of one case to $RDR_PARITY_REPORT (a JSON-lines file), if set."""
    path = os.environ.get('RDR_PARITY_REPORT')
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)     # a report must never fail a test
        with open(path, 'a') as f:
            f.write(json.dumps({'case': name, 'backend': backend_tag, **summary(rep), 'tensors': rep}) + '\n')
