#!/usr/bin/env python3
"""Adds `ref16_grad_shape6_vertices` to tests/golden/bunny_box_512x512x8.npz: the bunny's vertex gradient as the fp64 sum of 16
pixel-striped oracle passes (see make_golden.py, "ref64").  One row of that tensor -- vertex 4473, a gradient of 6.8e4, the norm
of the whole tensor -- stands out of the reference's single fp32 pass against the GPU's value (`flipped_rows = 1` in the parity
report); the striped sum takes the fp32 accumulation error out of it.  ~25 min of oracle time on 8 cores."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
if os.environ.get('MALLOC_MMAP_THRESHOLD_') != '65536':
    os.environ['MALLOC_MMAP_THRESHOLD_'] = '65536'
    os.execv(sys.executable, [sys.executable] + sys.argv)
import oracle_util  # noqa: E402
from golden.make_golden import CONFIG_CASES, render_case  # noqa: E402

NAME, KEY, K = 'bunny_box_512x512x8', 'grad_shape6_vertices', 16
ref = oracle_util.load_oracle()
acc = None
for k in range(K):
    part = render_case(ref, *CONFIG_CASES[NAME], stripe=(k, K))[KEY].astype(np.float64)
    acc = part if acc is None else acc + part
    print(k, flush=True)
path = os.path.join(HERE, NAME + '.npz')
z = np.load(path)
out = {k: z[k] for k in z.files}
out['ref16_' + KEY] = acc
np.savez_compressed(path, **out)
one = out[KEY].astype(np.float64)
rows = np.linalg.norm(one - acc, axis=1) / np.linalg.norm(acc)
print('single pass vs K=16: rel-L2 %.3e, rows above 1e-5:' % (np.linalg.norm(one - acc) / np.linalg.norm(acc)), np.nonzero(rows > 1e-5)[0], rows[rows > 1e-5])
