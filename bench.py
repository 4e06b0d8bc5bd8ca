#!/usr/bin/env python3
"""bench.py -- forward+backward throughput of the differentiable path tracer on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric "Msamples/s fwd+bwd (1024^2 x spp)"): the bunny_box scene
(tests/scenes.py, arrays exported from the reference's tests/scenes/bunny_box.xml), 1024x1024,
max_bounces 4, Sobol' sampler, gradients w.r.t. the bunny's vertices (+ edge sampling), as in
tests/test_bunny_box.py.  One step = one forward render + one backward render of `--spp` samples
per pixel PER GPU (default 32, so 8 GPUs reproduce config 4: 1024^2 x 256 spp sharded by sample
index); weak scaling.  value = N * W * H * spp * K / t / 1e6, t = max over ranks of the wall time of
the K steps (barrier + device sync on both sides).  Scene construction (triangle/edge hierarchy
build) is outside the timed region and reported separately, as SURVEY.md section 8d prescribes;
all inputs are resident in HBM when the clock starts.

Extra objects in the JSON line:
  roofline      -- the closest-hit traversal kernel: algorithmic bytes (40 B per ray + 32 B per node
                   record loaded + 36 B per triangle tested, counted by the instrumented kernel
                   variant on the same rays in an untimed pass) / its mean launch time, measured with
                   HIP events on the launch stream inside the timed region; vs 8 TB/s HBM.
  cpu_baseline  -- the reference's own C++ core (oracle/_ref, Embree stand-in) on this box's host
                   cores, on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
RAY_BYTES, HIT_BYTES, NODE_BYTES, TRI_BYTES = 32, 8, 32, 36


class Prepared:
    """Scene + buffers built once; step() = redner.render(forward) + redner.render(backward)."""

    def __init__(self, rd, scene, spp, total_spp, sample_offset, max_bounces, device, seed=1):
        from redner_amd.render_pytorch import RenderFunction
        self.rd, self.device = rd, device
        args = RenderFunction.serialize_scene(scene, total_spp, max_bounces, sampler_type=rd.SamplerType.sobol,
                                              device=device, backend=rd)
        self.meta, self.tensors = args[0], args[1:]
        t0 = time.time()
        self.u = RenderFunction.unpack_args((seed, seed + 1000003), self.meta, self.tensors)
        if device.type == 'cuda':
            torch.cuda.synchronize(device)
        self.scene_build_s = time.time() - t0
        self.u.options.num_samples = spp
        if hasattr(self.u.options, 'sample_offset'):
            self.u.options.sample_offset = sample_offset
            self.u.options.total_samples = total_spp
        vp = self.meta['camera']['viewport']
        self.h, self.w = vp[2] - vp[0], vp[3] - vp[1]
        self.img = torch.zeros(self.h, self.w, 3, device=device)
        self.d_img = torch.ones(self.h, self.w, 3, device=device)      # d(sum(img))/d(img)
        fp = lambda t: rd.float_ptr(t.data_ptr() if t is not None else 0)   # noqa: E731
        self.grads = []

        def z(i):
            if i < 0:
                return None
            g = torch.zeros(self.tensors[i].shape, dtype=torch.float32, device=device)
            self.grads.append(g)
            return g

        cm = self.meta['camera']
        look = cm['cam_to_world'] < 0
        d_cam = rd.DCamera(fp(z(cm['position']) if look else None), fp(z(cm['look_at']) if look else None),
                           fp(z(cm['up']) if look else None), fp(None if look else z(cm['cam_to_world'])),
                           fp(None if look else z(cm['world_to_cam'])), fp(z(cm['intrinsic_mat_inv'])),
                           fp(z(cm['intrinsic_mat'])), fp(None))
        d_shapes = [rd.DShape(fp(z(s['vertices'])), fp(z(s['uvs'])), fp(z(s['normals'])), fp(z(s['colors'])))
                    for s in self.meta['shapes']]

        def d_tex(cls, tm):
            lv = z(tm['levels'][0])
            return cls([fp(lv)], [0], [0], int(lv.shape[0]), fp(z(tm['uv_scale'])))

        d_mats = [rd.DMaterial(d_tex(rd.Texture3, m['diffuse_reflectance']), d_tex(rd.Texture3, m['specular_reflectance']),
                               d_tex(rd.Texture1, m['roughness']), rd.TextureN([], [], [], 0, rd.float_ptr(0)),
                               rd.Texture3([], [], [], 0, rd.float_ptr(0))) for m in self.meta['materials']]
        d_lights = [rd.DAreaLight(fp(z(l['intensity']))) for l in self.meta['lights']]
        idx = device.index if device.index is not None else 0
        self.d_scene = rd.DScene(d_cam, d_shapes, d_mats, d_lights, None, device.type == 'cuda', idx)
        self.seed = seed

    def step(self, i):
        rd, u = self.rd, self.u
        self.img.zero_()
        for g in self.grads:
            g.zero_()
        u.options.seed = self.seed + i
        rd.render(u.scene, u.options, rd.float_ptr(self.img.data_ptr()), rd.float_ptr(0), None, rd.float_ptr(0), rd.float_ptr(0))
        u.options.seed = self.seed + i + 1000003
        rd.render(u.scene, u.options, rd.float_ptr(0), rd.float_ptr(self.d_img.data_ptr()), self.d_scene,
                  rd.float_ptr(0), rd.float_ptr(0))


def trace_stats(reset=False):
    from redner_amd import _capi
    lib = _capi.lib()
    if reset:
        lib.rdr_trace_stats_reset()
        return None
    st = _capi.TraceStats()
    lib.rdr_trace_stats_get(ctypes.byref(st))
    return st


def cpu_baseline(max_bounces):
    """The reference's C++ core (oracle/_ref) on the host cores, bounded sample of the workload."""
    import oracle_util
    import scenes
    if not oracle_util.oracle_available():
        return None
    ref = oracle_util.load_oracle()
    res, spp = 256, 4
    cpu = torch.device('cpu')
    p = Prepared(ref, scenes.bunny_box(cpu, resolution=(res, res)), spp, spp, 0, max_bounces, cpu)
    p.step(0)                                   # warm-up (thread pool, page faults)
    t0 = time.time()
    reps = 0
    while time.time() - t0 < 12.0:
        p.step(reps + 1)
        reps += 1
    dt = time.time() - t0
    return {'value': res * res * spp * reps / dt / 1e6, 'unit': 'Msamples/s', 'cores': os.cpu_count(),
            'kind': 'reference',
            'sample': 'bunny_box %dx%d, %d spp fwd+bwd, max_bounces %d, Sobol, %d repetitions in %.1f s; '
                      'reference C++ core (oracle/_ref) with the BVH Embree stand-in, all host threads'
                      % (res, res, spp, max_bounces, reps, dt)}


def alone_leg(st, alg_bytes_launch):
    """The same kernel in an extra untimed step on one stream (RDR_NO_OVERLAP=1): informative, not the headline."""
    ms = st.closest_ms / max(st.closest_launches, 1)
    achieved = alg_bytes_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {'mean_launch_ms': ms, 'achieved': achieved, 'frac': achieved / HBM_PEAK_GBS,
            'note': 'untimed extra step with every stage on one stream'}


def measured_traffic(a):
    """HBM bytes per closest-hit launch from the PMC passes (rocprofv3 cannot run inside this process): the
    committed measurement in profiles/r1_traffic.json, valid for the default workload only, else None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r1_traffic.json')
    if a.res != 1024 or a.max_bounces != 4 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return float(json.load(f)['hbm_bytes_per_launch'])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--spp', type=int, default=32, help='samples per pixel per GPU per step')
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--max-bounces', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alone-leg', action='store_true', help='skip the extra single-stream step behind roofline.alone (profiling runs)')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % a.gpus
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda:%d' % local_rank)

    from redner_amd import redner
    import scenes
    total_spp = a.spp * world
    prep = Prepared(redner, scenes.bunny_box(dev, resolution=(a.res, a.res)), a.spp, total_spp, rank * a.spp,
                    a.max_bounces, dev)
    reduce_bufs = [prep.img] + [g for g in prep.grads]

    def reduce_all():
        # image + every gradient tensor: all_gather + fixed-order sum (bit-reproducible), see distributed.py
        if world == 1:
            return
        from redner_amd.distributed import _all_gather_sum
        for t in reduce_bufs:
            t.copy_(_all_gather_sum(t, dist.group.WORLD))

    from redner_amd import _capi
    lib = _capi.lib()
    for i in range(a.warmup):
        prep.step(i)
        reduce_all()
    lib.rdr_trace_stats_enable(1, 0)
    trace_stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.time()
    for i in range(a.steps):
        prep.step(a.warmup + i)
        reduce_all()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    st = trace_stats()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    samples = world * a.res * a.res * a.spp * a.steps
    value = samples / dt / 1e6

    # untimed pass with the instrumented traversal variant: node / triangle records per launch
    lib.rdr_trace_stats_enable(0, 1)
    trace_stats(reset=True)
    prep.step(a.warmup)                      # same seed as the first timed step -> same rays
    torch.cuda.synchronize(dev)
    cnt = trace_stats()
    lib.rdr_trace_stats_enable(0, 0)

    # untimed pass on ONE stream: the traversal kernel's launch duration without a neighbour on the GPU (in the timed
    # region the shadow-ray launch of the same bounce runs beside every closest-hit launch)
    alone = None
    if not a.no_alone_leg:
        os.environ['RDR_NO_OVERLAP'] = '1'
        lib.rdr_trace_stats_enable(1, 0)
        trace_stats(reset=True)
        prep.step(a.warmup)
        torch.cuda.synchronize(dev)
        alone = trace_stats()
        lib.rdr_trace_stats_enable(0, 0)
        del os.environ['RDR_NO_OVERLAP']

    out = None
    if rank == 0:
        per_step_launches = cnt.closest_launches
        rays = cnt.closest_rays
        alg_bytes_step = rays * (RAY_BYTES + HIT_BYTES) + cnt.closest_nodes * NODE_BYTES + cnt.closest_tris * TRI_BYTES
        mean_launch_ms = st.closest_ms / max(st.closest_launches, 1)
        alg_bytes_launch = alg_bytes_step / max(per_step_launches, 1)
        achieved = alg_bytes_launch / (mean_launch_ms * 1e-3) / 1e9 if mean_launch_ms > 0 else 0.0
        out = {
            'metric': 'Msamples/s fwd+bwd', 'value': value, 'unit': 'Msamples/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'bunny_box %dx%d, max_bounces %d, Sobol, %d spp per GPU per step fwd+bwd '
                                   '(vertex gradients, primary+secondary edge sampling); %d GPUs = %d spp sharded by '
                                   'sample index' % (a.res, a.res, a.max_bounces, a.spp, world, total_spp),
                       'resolution': [a.res, a.res], 'spp_per_gpu': a.spp, 'max_bounces': a.max_bounces,
                       'parallelism': 'sample-sharded x%d' % world},
            'scene_build_ms': prep.scene_build_s * 1e3,
            'roofline': {'kernel': 'trace_kernel<closest-hit>', 'bound': 'hbm', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': measured_traffic(a),
                         'mean_launch_ms': mean_launch_ms, 'launches_per_step': per_step_launches,
                         'rays_per_step': rays, 'nodes_per_ray': cnt.closest_nodes / max(rays, 1),
                         'tris_per_ray': cnt.closest_tris / max(rays, 1),
                         'algorithmic_bytes_per_launch': alg_bytes_launch,
                         'traversal_share_of_step': (st.closest_ms + st.any_ms) / (dt * 1e3),
                         'alone': alone_leg(alone, alg_bytes_launch) if alone is not None else None},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(a.max_bounces)
            except Exception as e:   # the baseline must never take the GPU number down with it
                out['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
