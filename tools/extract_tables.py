#!/usr/bin/env python3
"""Extract the two numeric tables the renderer consumes as *data* from the reference checkout:

  * Sobol' direction matrices (Gruenschloss, 1024 dims x 52 bits)  /root/reference/src/sobol.inc
  * Linearly-transformed-cosine matrices tabM (Heitz et al., 128x128x9)  /root/reference/src/ltc.inc

and store them as raw little-endian binaries under redner_amd/data/.  Only the numbers are taken
("we consume the values verbatim as data"); no code is copied.
Run once in the build container: python tools/extract_tables.py
"""
import re, sys, os
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/src'
out = os.path.join(os.path.dirname(__file__), '..', 'redner_amd', 'data')
os.makedirs(out, exist_ok=True)

txt = open(os.path.join(ref, 'sobol.inc')).read()
body = txt[txt.index('matrices_'):]
body = body[body.index('{') + 1: body.index('};')]
vals = re.findall(r'0x[0-9a-fA-F]+|\d+', body)
arr = np.array([int(v.rstrip('ULul'), 0) for v in vals], dtype=np.uint64)
assert arr.size == 1024 * 52, arr.size
arr.tofile(os.path.join(out, 'sobol_1024x52.u64'))
print('sobol', arr.size, hex(int(arr[0])))

txt = open(os.path.join(ref, 'ltc.inc')).read()
i = txt.index('tabM')
body = txt[txt.index('{', i) + 1: txt.index('};', i)]
vals = re.findall(r'[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?', body)
arr = np.array([float(v) for v in vals], dtype=np.float32)
print('ltc', arr.size)
assert arr.size == 128 * 128 * 9, arr.size
arr.tofile(os.path.join(out, 'ltc_tabM_128x128x9.f32'))
