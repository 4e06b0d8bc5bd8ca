"""GPU box: one fixture case under several tunings against the one-sample-per-launch render and the oracle fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch
from redner_amd import _capi as K
K.load()
from redner_amd import redner
from golden.make_golden import render_case, CASES
name = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device('cuda:0')
gold = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
b, res, spp, mb = CASES[name][:4]
ch = CASES[name][4] if len(CASES[name]) > 4 else None
base_opts = dict(CASES[name][5]) if len(CASES[name]) > 5 and CASES[name][5] else {}
def run(t):
    o = dict(base_opts); o['tuning'] = t
    return render_case(redner, b, res, spp, mb, ch, o, device=dev)
one = run({'batch_samples': 1})
def d(a, r):
    a, r = a.astype(np.float64), r.astype(np.float64)
    return np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-300)
keys = [key] if key else [k for k in one if k != 'image']
print('one vs gold:', {k: '%.1e' % d(one[k], gold[k]) for k in keys})
for nm, t in (('default', {}), ('again', {}), ('one_stream', {'flags': K.TUNE_NO_OVERLAP}), ('batch2', {'batch_samples': 2}), ('batch3', {'batch_samples': 3}),
              ('general', {'flags': K.TUNE_FORCE_GENERAL}), ('general+one_stream', {'flags': K.TUNE_FORCE_GENERAL | K.TUNE_NO_OVERLAP}),
              ('no_hoist', {'flags': K.TUNE_NO_HOIST}), ('walk', {'flags': K.TUNE_PICKN_WALK}), ('workers1', {'workers': 1})):
    o = run(t)
    print('%-20s image_eq=%s  vs one: %s   vs gold: %s' % (nm, np.array_equal(o['image'], one['image']),
          {k: '%.1e' % d(o[k], one[k]) for k in keys if d(o[k], one[k]) > 1e-9}, {k: '%.1e' % d(o[k], gold[k]) for k in keys if d(o[k], gold[k]) > 1e-5}))
