"""Print the vertex rows of a golden case that differ most from the fixture (GPU run).  Used to tell
isolated edge-sample flips (a handful of rows, everything else ~1e-7) from real discrepancies."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'tests')); sys.path.insert(0, os.path.join(HERE, '..'))
import numpy as np, torch
from golden.make_golden import CASES, render_case
import redner_amd.redner as rd

name = sys.argv[1]
out = render_case(rd, *CASES[name], device=torch.device('cuda:0'))
gold = np.load(os.path.join(HERE, '..', 'tests', 'golden', name + '.npz'))
for k in gold.files:
    g = gold[k].astype(np.float64); m = out[k].astype(np.float64)
    n = np.linalg.norm(g)
    rel = np.linalg.norm(m - g) / (n if n > 0 else 1)
    print('%-28s |g|=%.4g rel=%.3g' % (k, n, rel))
    if rel > 1e-5 and g.ndim == 2 and g.shape[0] > 16:
        row = np.linalg.norm(m - g, axis=1)
        for i in np.argsort(-row)[:12]:
            print('   row %5d err %.3g  mine %s  gold %s' % (i, row[i], m[i], g[i]))
