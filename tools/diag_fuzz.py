"""GPU box: one scene of tests/test_fuzz_parity.py (family seed) by the oracle and by the product library under several tunings;
prints per-tensor distances and the rows that stand out.  python tools/diag_fuzz.py mesh 118"""
import os, sys
if os.environ.get('MALLOC_MMAP_THRESHOLD_') != '1024' or os.environ.get('MALLOC_PERTURB_') != '255':
    os.environ['MALLOC_MMAP_THRESHOLD_'] = '1024'
    os.environ['MALLOC_PERTURB_'] = '255'
    os.execv(sys.executable, [sys.executable] + sys.argv)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch
import oracle_util
import test_fuzz_parity as F
from redner_amd import _capi as K
_ = K.load()
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
F.MINE_DEVICE, F.ON_GPU = torch.device('cuda:0'), True
oracle = oracle_util.load_oracle()
family, seed = sys.argv[1], int(sys.argv[2])

def render(backend, tuning):
    # tuning reaches unpack_args through meta: patch serialize_scene for this call
    orig = RenderFunction.serialize_scene
    def ser(*a, **kw):
        if backend is redner and tuning:
            kw['tuning'] = tuning
        return orig(*a, **kw)
    RenderFunction.serialize_scene = staticmethod(ser)
    try:
        if family == 'plain':
            return F._render(backend, seed, 2 + seed % 3, 1 + seed % 3)
        if family == 'rich':
            return F._render_rich(backend, seed, 2 + seed % 4, 1 + seed % 5, seed % 5 == 0)
        if family == 'mesh':
            return F._render_mesh(backend, seed, 2 + seed % 3, seed % 4)
        if family == 'odd':
            return F._render_odd(backend, seed, 1 + seed % 6, seed % 7)
        if family == 'blob':
            return F._render_blob(backend, seed)
    finally:
        RenderFunction.serialize_scene = staticmethod(orig)

ref = render(oracle, None)
print('spp/mb', 2 + seed % 3, seed % 4, 'keys', sorted(ref))
for name, t in (('default', {}), ('refill_all', {'flags': K.TUNE_REFILL_ALL}), ('batch1', {'batch_samples': 1}),
                ('one_stream', {'flags': K.TUNE_NO_OVERLAP}), ('general', {'flags': K.TUNE_FORCE_GENERAL}), ('binary', {'flags': K.TUNE_TRACE_BINARY}),
                ('walk', {'flags': K.TUNE_PICKN_WALK}), ('fused', {'flags': K.TUNE_PICKH_FUSED}), ('default again', {})):
    mine = render(redner, t)
    line = []
    for k in sorted(ref):
        r, m = ref[k].astype(np.float64), mine[k].astype(np.float64)
        n = np.linalg.norm(r)
        d = np.linalg.norm(m - r) / max(n, 1e-300)
        if d > 1e-6:
            extra = ''
            if r.ndim == 2 and r.shape[0] > 1:
                rows = np.linalg.norm(m - r, axis=1) / max(n, 1e-300)
                extra = ' rows>' + str([(int(i), float('%.2e' % rows[i])) for i in np.argsort(-rows)[:6] if rows[i] > 1e-5])
            line.append('%s %.2e%s' % (k, d, extra))
    print('%-14s image_equal=%s  %s' % (name, np.array_equal(ref['image'], mine['image']), '; '.join(line) or 'all <= 1e-6'))
