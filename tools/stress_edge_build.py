"""GPU box: edge structures (list, both hierarchies) of many small random scenes built by the product -- edge builder thread
beside a forward render on the calling thread, as in a training loop -- against the oracle's, link for link.  Hunts races in
the device build (radix sort, radix tree, treelets).  python tools/stress_edge_build.py [rounds]"""
import ctypes, os, sys, glob, importlib.util
if os.environ.get('MALLOC_MMAP_THRESHOLD_') != '1024' or os.environ.get('MALLOC_PERTURB_') != '255':
    os.environ['MALLOC_MMAP_THRESHOLD_'] = '1024'
    os.environ['MALLOC_PERTURB_'] = '255'
    os.execv(sys.executable, [sys.executable] + sys.argv)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
import oracle_util
import test_fuzz_parity as F
from redner_amd import _capi as K
K.load()
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
ref = oracle_util.load_oracle()
sys.modules.setdefault('redner', ref)
p = glob.glob(os.path.join(ROOT, 'oracle', '_ref', 'redner_dbg*.so'))[0]
spec = importlib.util.spec_from_file_location('redner_dbg', p); dbg = importlib.util.module_from_spec(spec); spec.loader.exec_module(dbg)
lib = K.lib()
lib.rdr_debug_dump_edges.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
bad = {}
n = 0
for rnd in range(rounds):
    for fam, make, seeds in (('plain', F._scene, range(1, 201, 3)), ('rich', F._scene_rich, range(1, 161, 3)), ('odd', F._scene_odd, range(1, 121, 3)),
                             ('mesh', F._scene_mesh, range(1, 121, 3))):
        for seed in seeds:
            cpu = torch.device('cpu')
            sc = make(seed, cpu)
            a, b = '/tmp/se_mine.txt', '/tmp/se_ref_%s_%d.txt' % (fam, seed)
            if not os.path.exists(b):
                args = RenderFunction.serialize_scene(sc, 1, 2, sampler_type=ref.SamplerType.sobol, device=cpu, backend=ref)
                u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
                dbg.dump(u.scene, b)
            sc = make(seed, cpu)
            args = RenderFunction.serialize_scene(sc, 2, 2, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner)
            u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
            img = torch.zeros(sc.camera.resolution[0], sc.camera.resolution[1], 3, device=dev)
            vp = args[0]['camera']['viewport']
            img = torch.zeros(vp[2] - vp[0], vp[3] - vp[1], 3, device=dev)
            redner.render(u.scene, u.options, redner.float_ptr(img.data_ptr()), redner.float_ptr(0), None, redner.float_ptr(0), redner.float_ptr(0))
            assert lib.rdr_debug_dump_edges(u.scene._handle, a.encode()) == 0
            n += 1
            la, lb = open(a).read().split('\n'), open(b).read().split('\n')
            ok = len(la) == len(lb)
            if ok:
                for x, y in zip(la, lb):
                    xs, ys = x.split(), y.split()
                    if len(xs) != len(ys): ok = False; break
                    if len(xs) >= 7 and not x.startswith(('edges', 'cs', 'ncs', 'expand')):
                        if xs[:5] != ys[:5] or any(float(p) != float(q) for p, q in zip(xs[5:], ys[5:])): ok = False; break
                    elif xs != ys: ok = False; break
            if not ok:
                bad.setdefault('%s %d' % (fam, seed), []).append(rnd)
print('scenes built %d, mismatching the oracle: %s' % (n, bad))
