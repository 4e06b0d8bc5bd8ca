#!/usr/bin/env python3
"""The camera gradients of a case under a DEFINED accumulation order (build container; fixtures <case>_ref_order.npz).

Two processes render the same case:
  oracle1t   the reference's C++ core (oracle/_ref) with ONE visible processor (LD_PRELOAD oracle/_ref/nprocs_shim.so): its
             parallel_for runs on the calling thread, so every fp32 atomic add (src/atomic.h:43-141) happens in a defined order --
             sample by sample, kernel by kernel, lane by lane;
  harness    the product's stage bodies on the CPU debugging harness (tests/hostsim), one sample per launch, one host thread,
             with RDR_HOSTSIM_REF_ORDER=1: beside its fp64 accumulators it keeps the reference's floats (`float += (float)term`)
             in the order it issues the adds, which for the camera tensors is the same order (one add per lane in
             d_primary_intersection, then one per slot in compute_primary_edge_derivatives, src/camera.h:255-257,824-826);
             and once more without the switch: the fp64 sums.
If the harness' floats equal the one-thread oracle's, the distance between the oracle's value and the fp64 sum is accumulation
error of the reference's floats and nothing else; the fp64 sum is then the value a GPU result is compared with
(tests/parity_util.py, tests/test_accumulation_order.py).

  python tests/golden/make_ref_order.py <case> [...]          # writes tests/golden/<case>_ref_order.npz
  python tests/golden/make_ref_order.py --leg oracle1t|harness32|harness64 <case> <out.npz>   (internal)"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
ONE_CORE = os.path.join(ROOT, 'oracle', '_ref', 'nprocs_shim.so')
HOSTSIM = os.path.join(ROOT, 'tests', 'hostsim', '_build', 'libredner_hostsim.so')
CAM = ('grad_cam_position', 'grad_cam_look_at', 'grad_cam_up')


def inputs_digest(builder, res, spp, mb):
    """SHA-256 over what a ref-order fixture was rendered from: every tensor the scene builder hands to serialize_scene + the
    render options.  Stored in the fixture (`inputs_sha256`) and recomputed by tests/test_accumulation_order.py: a fixture whose
    scene or options have since changed is STALE and must be regenerated, not compared with."""
    import hashlib
    import torch
    import scenes
    from redner_amd import redner
    from redner_amd.render_pytorch import RenderFunction
    sc = getattr(scenes, builder)(torch.device('cpu'), resolution=res if isinstance(res, tuple) else (res, res))
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=redner.SamplerType.sobol, device=torch.device('cpu'), backend=redner)
    h = hashlib.sha256(('%s %s %d %d' % (builder, res, spp, mb)).encode())
    for t in args[1:]:
        if t is not None:
            a = t.detach().cpu().contiguous().numpy()
            h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def leg(which, case, out):
    import torch  # noqa: F401
    from golden.make_golden import CASES, CONFIG_CASES, render_case
    spec = list(CASES.get(case) or CONFIG_CASES[case])
    while len(spec) < 6:
        spec.append(None)
    if which == 'oracle1t':
        import oracle_util
        backend = oracle_util.load_oracle()
    else:
        from redner_amd import _capi
        _capi.load(HOSTSIM)
        from redner_amd import redner as backend
        spec[5] = dict(spec[5] or {}, tuning={'batch_samples': 1, 'workers': 1})
    res = render_case(backend, *spec)
    np.savez(out, **{k: v for k, v in res.items() if k.startswith('grad_cam_')})


def run_leg(which, case, out):
    env = dict(os.environ, MALLOC_MMAP_THRESHOLD_='65536', MALLOC_PERTURB_='255')
    if which == 'oracle1t':
        env['LD_PRELOAD'] = ONE_CORE
        env['ORACLE_NPROCS'] = '1'
    if which == 'harness32':
        env['RDR_HOSTSIM_REF_ORDER'] = '1'
    subprocess.check_call([sys.executable, os.path.abspath(__file__), '--leg', which, case, out], env=env)
    return np.load(out)


def make(case, tmp='/tmp', out_dir=HERE):
    res = {w: run_leg(w, case, os.path.join(tmp, 'ref_order_%s_%s.npz' % (case, w))) for w in ('oracle1t', 'harness32', 'harness64')}
    out = {}
    for k in res['oracle1t'].files:
        for w in res:
            out['%s_%s' % (w, k)] = res[w][k]
        o, h32, h64 = (res[w][k].astype(np.float64) for w in ('oracle1t', 'harness32', 'harness64'))
        n = np.linalg.norm(h64)
        print('%s %-20s harness floats vs one-thread oracle %.3e | oracle vs fp64 sum %.3e | floats vs fp64 sum %.3e'
              % (case, k, np.linalg.norm(h32 - o) / n, np.linalg.norm(o - h64) / n, np.linalg.norm(h32 - h64) / n), flush=True)
    from golden.make_golden import CASES, CONFIG_CASES
    spec = (CASES.get(case) or CONFIG_CASES[case])
    out['inputs_sha256'] = inputs_digest(*spec[:4])
    np.savez(os.path.join(out_dir, case + '_ref_order.npz'), **out)
    return out


# ---- the job bench.py validates against the live reference: bunny_box 1024 x 1024, 1 spp, forward + backward ----------------
# bench.py's `gpu_vs_reference` compares the GPU with ONE pass of the reference rendered in the same run; that pass sums its
# few-element tensors (camera, light intensity, constant reflectances) with fp32 atomics and is itself 1e-4 ... 1e-2 away from
# the exact sum of its addends (round 5's line printed 3.3e-3 with a note).  The same three legs as above, on bench.py's own job
# (bench.Prepared: d_image = 1, seeds 1 / 1000004, every gradient buffer): the fixture holds, for every gradient tensor of at
# most 64 elements, the one-thread oracle, the harness' floats in reference order and the harness' fp64 sums; bench.py reports
# the GPU against the fp64 sums (`worst_few_element_gradient_rel_l2`, <= 1e-4).
BENCH_JOB = ('bunny_box', 1024, 1, 4)
BENCH_FIXTURE = 'bench_job_bunny_box_1024x1024x1_ref_order.npz'


def bench_leg(which, out):
    import argparse
    import torch
    import bench
    if which == 'oracle1t':
        import oracle_util
        backend = oracle_util.load_oracle()
    else:
        from redner_amd import _capi
        _capi.load(HOSTSIM)
        from redner_amd import redner as backend
    workload, res, spp, mb = BENCH_JOB
    a = argparse.Namespace(workload=workload, res=res, max_bounces=mb)
    cpu = torch.device('cpu')
    p = bench.Prepared(backend, bench.build_scene(a, cpu, res), spp, spp, 0, mb, cpu)
    if which != 'oracle1t':
        p.u.options.tuning.batch_samples, p.u.options.tuning.workers = 1, 1          # one sample per launch, one host thread
    p.step(0)
    np.savez(out, **{'g%d' % i: g.numpy() for i, g in enumerate(p.grads) if g.numel() <= 64},
             big_norms=np.asarray([float(g.double().norm()) for g in p.grads]))


def make_bench_job(tmp='/tmp', out_dir=HERE):
    res = {}
    for w in ('oracle1t', 'harness32', 'harness64'):
        out = os.path.join(tmp, 'ref_order_bench_%s.npz' % w)
        env = dict(os.environ, MALLOC_MMAP_THRESHOLD_='65536', MALLOC_PERTURB_='255')
        if w == 'oracle1t':
            env['LD_PRELOAD'] = ONE_CORE
            env['ORACLE_NPROCS'] = '1'
        if w == 'harness32':
            env['RDR_HOSTSIM_REF_ORDER'] = '1'
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--bench-leg', w, out], env=env)
        res[w] = np.load(out)
    out = {}
    for k in res['oracle1t'].files:
        if k == 'big_norms':
            continue
        for w in res:
            out['%s_%s' % (w, k)] = res[w][k]
        o, h32, h64 = (res[w][k].astype(np.float64) for w in ('oracle1t', 'harness32', 'harness64'))
        n = np.linalg.norm(h64)
        if n > 0:
            print('bench job %-5s (%2d elements) harness floats vs one-thread oracle %.3e | oracle vs fp64 sum %.3e'
                  % (k, o.size, np.linalg.norm(h32 - o) / n, np.linalg.norm(o - h64) / n), flush=True)
    out['inputs_sha256'] = inputs_digest(*BENCH_JOB)
    np.savez(os.path.join(out_dir, BENCH_FIXTURE), **out)
    return out


if __name__ == '__main__':
    if sys.argv[1] == '--leg':
        leg(*sys.argv[2:5])
    elif sys.argv[1] == '--bench-leg':
        bench_leg(*sys.argv[2:4])
    elif sys.argv[1] == '--bench-job':
        make_bench_job()
    elif sys.argv[1] == '--stamp':            # add inputs_sha256 to fixtures made before round 6 (their renders are not repeated)
        from golden.make_golden import CASES, CONFIG_CASES
        for c in sys.argv[2:]:
            path = os.path.join(HERE, BENCH_FIXTURE if c == 'bench-job' else c + '_ref_order.npz')
            z = np.load(path)
            out = {k: z[k] for k in z.files}
            out['inputs_sha256'] = inputs_digest(*(BENCH_JOB if c == 'bench-job' else (CASES.get(c) or CONFIG_CASES[c])[:4]))
            np.savez(path, **out)
            print(c, out['inputs_sha256'][:6])
    else:
        for c in sys.argv[1:]:
            make(c)
