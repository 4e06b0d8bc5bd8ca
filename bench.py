#!/usr/bin/env python3
"""bench.py -- forward+backward throughput of the differentiable path tracer on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it runs one rank per GPU (backend nccl =
RCCL) -- launched under torch.distributed.run, or, when started plainly (`python bench.py --gpus 8`), it starts its own
ranks through torch.distributed.run (self_launch).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric "Msamples/s fwd+bwd (1024^2 x spp)", config 4): the bunny_box scene
(tests/scenes.py, arrays exported from the reference's tests/scenes/bunny_box.xml), 1024 x 1024,
max_bounces 4, Sobol' sampler, 256 spp, gradients w.r.t. the bunny's vertices with primary and
secondary edge sampling, as in tests/test_bunny_box.py.  One step = one forward render + one backward
render of the WHOLE 256-spp job; N ranks shard it by Sobol' sample index (rank r renders samples
[r*256/N, (r+1)*256/N) of the full frame) and all-gather + sum the image and every
gradient tensor in fixed rank order: strong scaling.  value = W*H*spp*K / t / 1e6 with t = max over
ranks of the wall time of the K steps (barrier + device sync on both sides).  Scene construction is
outside the timed region and reported separately; inputs are resident in HBM.

`--workload living_room_standin` is BASELINE config 5's stand-in (tests/scenes.py): the general kernels
(mip-mapped textures, two-sided materials, ray differentials), max_bounces 6, camera-pose gradients.

The library parks at most 8 GiB of device memory between calls by default; this benchmark owns its GPU and raises the bound to
32 GiB (`config.pool_cap_mb`; RDR_POOL_CAP_MB overrides): two sample workers x 8 Sobol' samples of the frame per launch set
(8 GiB: 2 x 2, 77.7 Msamples/s; 16 GiB: 2 x 4, 84.3; 32 GiB: 2 x 8, 86.0 at the end of round 6; profiles/r6_notes.md).  `config.samples_per_launch` /
`config.sample_workers` say how the library scheduled the timed steps.  per_rank_ms_per_step / per_rank_render_ms_per_step: every
rank's wall time per step and the part of it inside render() (the rest: the one collective per step + waiting for the slowest rank).

Extra objects in the JSON line:
  roofline      -- the closest-hit traversal kernel by SURVEY.md section 8d: algorithmic bytes (40 B per ray +
                   32 B per node record + 36 B per triangle tested, counted per SAMPLE by the instrumented kernel in an
                   untimed pass; the timed region's bytes = bytes per sample x its samples) / the time during which at
                   least one closest-hit launch was in flight (union of HIP-event intervals on the launch streams inside
                   the timed region), vs 8 TB/s.  `per_launch`: the same bytes / the sum of the launches' own durations;
                   `alone`: an extra untimed step with one sample worker and one stream.  Next to it what the hardware
                   counters say (rocprofv3 passes run from inside this process on a 32-spp job, `kernels`): HBM bytes
                   actually moved (`traffic`, hbm_frac_measured), vector-ALU lane utilisation -- and the same for the
                   kernels that dominate the backward pass (fp64 VALU lane-operations/s against the 39.3 T/s the chip can
                   issue, HBM bytes/s against 8 TB/s).
  roofline_large-- the same kernel on a hierarchy that does not fit the L2 (bunny tessellated to 3.7 M triangles), forward
                   render, + three counter passes: what the design does when traversal goes to the MALL / HBM.
  collective / sharded_check -- when a process group exists: the one all_gather per step (backend, bytes, ms), and the
                   gathered sum of one sample per rank against the same blocks rendered on rank 0 (bit-identical).
  cpu_baseline  -- the reference's own C++ core (oracle/_ref, Embree stand-in) on this box's host
                   cores, on bounded samples of the same workload (rank 0, N = 1 only); `gpu_vs_reference`: the full frame
                   at 1 spp rendered by both and compared (image + every gradient tensor).
  self_check    -- untimed: the frame in 8 sample blocks (what 8 ranks render) against one call.
"""
import argparse
import collections
import csv
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9     # CUs x SIMDs x lanes/clk x clock = 39.3 T lane-operations/s (79 TFLOP/s fp64 FMA)
RAY_BYTES, HIT_BYTES, NODE_BYTES, WIDE_NODE_BYTES, TRI_BYTES = 32, 8, 32, 128, 36
BENCH_POOL_CAP_MB = 32768      # buffer-cache bound this benchmark asks for (the library default is 8192): two workers x 8-sample batches
INNER_SPP = 32                 # samples of the job the rocprofv3 passes run (forms the same sample batches as the timed job at 1024 x 1024)


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
        self.d_scene, grads = RenderFunction.create_gradient_buffers(self.meta, self.tensors)
        self.grads = [g for g in grads if g is not None]
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


def build_scene(a, device, res):
    import scenes
    if a.workload == 'living_room_standin':
        return scenes.living_room_standin(device, resolution=(res, res))
    if a.workload == 'living_room_standin_envmap':      # + environment map, specular / roughness textures: the GENERAL kernels
        return scenes.living_room_standin_envmap(device, resolution=(res, res))
    return scenes.bunny_box(device, resolution=(res, res))


def _rel_l2(x, y):
    x, y = x.double().cpu().flatten(), y.double().cpu().flatten()
    n = float(torch.linalg.norm(y))
    return float(torch.linalg.norm(x - y)) / n if n > 0 else float(torch.linalg.norm(x))


def cpu_baseline(a, gpu_rd=None, gpu_dev=None):
    """The reference's C++ core (oracle/_ref) on the host cores, bounded samples of the workload: a small frame at several spp
    and the benchmark's own 1024 x 1024 frame at 1 spp (4 096 chunks of 256 pixels for the reference's thread pool,
    src/parallel.cpp:228-255: enough work for every core of a 256-thread box).  The better rate is reported.  The full-frame
    sample is also rendered by the GPU library with the same seeds and compared -- the checker used as a checker: the job the
    benchmark times, validated at its own resolution (forward image and every gradient tensor)."""
    import oracle_util
    if not oracle_util.oracle_available():
        return None
    ref = oracle_util.load_oracle()
    cpu = torch.device('cpu')
    lines, best, check = [], None, None
    for res, spp, min_reps, budget in ((256, 4, 3, 30.0), (a.res, 1, 3, 90.0)):
        p = Prepared(ref, build_scene(a, cpu, res), spp, spp, 0, a.max_bounces, cpu)
        t0 = time.time()
        p.step(0)                                   # first pass (starts the thread pool, faults the pages in); also the step the GPU is compared with
        times = [time.time() - t0]
        if gpu_rd is not None and res == a.res:
            g = Prepared(gpu_rd, build_scene(a, gpu_dev, res), spp, spp, 0, a.max_bounces, gpu_dev)
            g.step(0)
            torch.cuda.synchronize(gpu_dev)
            errs = [(_rel_l2(x, y), y.numel()) for x, y in zip(g.grads, p.grads) if float(y.abs().sum()) > 0]
            big = [e for e, n in errs if n > 64]          # per-vertex / per-texel tensors
            few = [e for e, n in errs if n <= 64]         # few-element accumulators: the reference sums them with fp32 atomics
            check = {'job': '%s %dx%d, %d spp fwd+bwd' % (a.workload, res, res, spp), 'image_rel_l2': _rel_l2(g.img, p.img),
                     'worst_per_vertex_gradient_rel_l2': max(big, default=0.0),
                     'worst_few_element_gradient_rel_l2': max(few, default=0.0), 'tensors': len(errs),
                     'few_element_reference': 'the single reference pass of this run: its few-element tensors (light, reflectances, '
                                              'camera) carry the error of its fp32 atomics'}
            # ... which a committed fixture takes out for exactly this job (tests/golden/make_ref_order.py --bench-job): the fp64
            # sum of the reference's own addends (its floats in reference order equal the one-thread reference bit for bit)
            fix = os.path.join(ROOT, 'tests', 'golden', 'bench_job_bunny_box_1024x1024x1_ref_order.npz')
            if (a.workload, res, spp, a.max_bounces) == ('bunny_box', 1024, 1, 4) and os.path.exists(fix):
                import numpy as np
                z = np.load(fix)
                vs64 = [_rel_l2(gt, torch.from_numpy(z['harness64_g%d' % i])) for i, gt in enumerate(g.grads)
                        if 'harness64_g%d' % i in z.files and float(np.abs(z['harness64_g%d' % i]).sum()) > 0]
                if vs64:
                    check['worst_few_element_gradient_rel_l2_vs_single_reference_pass'] = check['worst_few_element_gradient_rel_l2']
                    check['worst_few_element_gradient_rel_l2'] = max(vs64)
                    check['few_element_reference'] = ('fp64 sum of the reference\'s addends for this job (committed fixture, '
                                                      'tests/golden/make_ref_order.py --bench-job; %d tensors)' % len(vs64))
                    check['ok'] = bool(check['image_rel_l2'] == 0.0 and check['worst_per_vertex_gradient_rel_l2'] < 1e-4 and max(vs64) < 1e-4)
            del g
        # at least `min_reps` passes, each timed on its own, while the sample stays within its budget of CPU seconds
        while len(times) < min_reps and sum(times) + min(times) < budget:
            t1 = time.time()
            p.step(len(times))
            times.append(time.time() - t1)
        best_t = min(times)
        rate = res * res * spp / best_t / 1e6
        lines.append({'sample': '%dx%d, %d spp' % (res, res, spp), 'value': rate, 'repetitions': len(times),
                      'seconds_per_pass': times, 'value_mean': res * res * spp * len(times) / sum(times) / 1e6})
        if best is None or rate > best[0]:
            best = (rate, res, spp, len(times), best_t)
        del p
    # ... and the launch shape the timed region actually runs -- one 16-sample batch of the full frame: 16.8 M lanes, 15.3 M-ray
    # queues, the refilling traversal kernel, per-sample segment tables -- against the oracle's committed fixture
    # (tests/golden/bunny_box_1024x1024x16.npz: ~10 min of oracle time, not repeated here): image bit for bit, vertex gradient
    if check is not None and gpu_rd is not None and (a.workload, a.res, a.max_bounces) == ('bunny_box', 1024, 4):
        from golden.make_golden import full_frame_check
        same, blocks_err, e = full_frame_check(gpu_rd, 'bunny_box_1024x1024x16', gpu_dev)
        check['batched_16spp_vs_oracle_fixture'] = {'job': 'bunny_box 1024x1024, 16 spp fwd+bwd = one 16-sample batch', 'image_bit_identical': same,
                                                    'largest_block_sum_difference': blocks_err, 'vertex_gradient_rel_l2': e,
                                                    'ok': bool(same and e < 1e-4)}
    rate, res, spp, reps, dt = best
    threads = os.cpu_count()
    sweep = cpu_thread_sweep(a)
    for leg in sweep:
        if leg.get('value', 0.0) > rate:
            rate, res, spp, reps, dt, threads = leg['value'], 256, 4, 3, min(leg['seconds_per_pass']), leg['threads']
    return {'value': rate, 'unit': 'Msamples/s', 'cores': threads, 'host_threads': os.cpu_count(), 'kind': 'reference', 'thread_sweep': sweep,
            'sample': '%s %dx%d, %d spp fwd+bwd, max_bounces %d, Sobol, best of %d passes (%.1f s); reference C++ core '
                      '(oracle/_ref) with the BVH Embree stand-in; the best of: all host threads on the two samples in `samples`, '
                      'and 8 / 16 / 32 / 64 threads on the first (`thread_sweep`); `cores` = the threads of the best run'
                      % (a.workload, res, res, spp, a.max_bounces, reps, dt),
            'samples': lines, 'gpu_vs_reference': check}


def cpu_baseline_leg(a):
    """Body of one leg of the thread sweep (a child process under LD_PRELOAD=oracle/_ref/nprocs_shim.so, ORACLE_NPROCS=n): the
    256 x 256, 4-spp sample of the workload on the reference's C++ core, best of 3 passes; prints one JSON line."""
    import oracle_util
    ref = oracle_util.load_oracle()
    cpu = torch.device('cpu')
    res, spp = 256, 4
    p = Prepared(ref, build_scene(a, cpu, res), spp, spp, 0, a.max_bounces, cpu)
    times = []
    for i in range(3):
        t0 = time.time()
        p.step(i)
        times.append(time.time() - t0)
    print(json.dumps({'threads': int(os.environ.get('ORACLE_NPROCS', '0')), 'value': res * res * spp / min(times) / 1e6,
                      'seconds_per_pass': times}), flush=True)


def cpu_thread_sweep(a, counts=(8, 16, 32, 64)):
    """The reference starts hardware_concurrency() - 1 workers per call (src/parallel.cpp:228-255); on a 256-thread host that is
    not its best configuration for a frame this size.  The same sample on 8 / 16 / 32 / 64 threads (oracle/nprocs_shim.c), each leg
    its own process; cpu_baseline reports the best of these and the all-threads run."""
    shim = os.path.join(ROOT, 'oracle', '_ref', 'nprocs_shim.so')
    if not os.path.exists(shim):
        return []
    out = []
    for n in counts:
        if n > (os.cpu_count() or 1):
            continue
        env = dict(os.environ, LD_PRELOAD=shim, ORACLE_NPROCS=str(n))
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(k, None)
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-baseline-leg', '--workload', a.workload,
                                '--max-bounces', str(a.max_bounces)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               text=True, timeout=180)
            out.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]))
        except Exception as e:
            out.append({'threads': n, 'error': repr(e)})
    return out


def sharded_self_check(a, rd, dev):
    """Untimed: the benchmark's frame rendered as 8 sample blocks (sample_offset = b, total_samples = 8: what 8 ranks would
    render) against the same 8 samples in one call -- image to fp32 summation order, every gradient tensor to 1e-4."""
    spp, blocks = 8, 8
    whole = Prepared(rd, build_scene(a, dev, a.res), spp, spp, 0, a.max_bounces, dev)
    whole.step(0)
    torch.cuda.synchronize(dev)
    img = whole.img.clone()
    grads = [g.clone() for g in whole.grads]
    acc_img = torch.zeros_like(img)
    acc = [torch.zeros_like(g) for g in grads]
    for b in range(blocks):
        part = Prepared(rd, build_scene(a, dev, a.res), spp // blocks, spp, b * (spp // blocks), a.max_bounces, dev)
        part.step(0)
        torch.cuda.synchronize(dev)
        acc_img += part.img
        for x, y in zip(acc, part.grads):
            x += y
        del part
    e_img = _rel_l2(acc_img, img)
    e_grad = max((_rel_l2(x, y) for x, y in zip(acc, grads) if float(y.abs().sum()) > 0), default=0.0)
    return {'job': '%s %dx%d, %d spp in %d sample blocks vs one call' % (a.workload, a.res, a.res, spp, blocks),
            'image_rel_l2': e_img, 'worst_gradient_rel_l2': e_grad, 'ok': bool(e_img < 2e-6 and e_grad < 1e-4)}


def large_hierarchy_leg(levels=4, spp=8, res=1024, max_bounces=4):
    """Untimed: the closest-hit kernel on a hierarchy that does NOT fit the L2 -- bunny_box with the bunny tessellated to 3.7 M
    triangles (tests/scenes.py: bunny_box_subdivided; ~250 MB of node + triangle records against 4 MiB of L2 per XCD), forward
    render of the same frame (camera rays + four bounces of cosine-distributed rays: the queue kinds of the benchmark).  Same
    accounting as `roofline`: records per ray from the instrumented kernel, launch durations from HIP events."""
    import scenes
    from redner_amd import _capi, redner
    from redner_amd.render_pytorch import RenderFunction
    lib = _capi.lib()
    dev = torch.device('cuda:0')
    t0 = time.time()
    sc = scenes.bunny_box_subdivided(dev, resolution=(res, res), levels=levels)
    torch.cuda.synchronize(dev)
    t_mesh = time.time() - t0
    tris = sum(int(s.indices.shape[0]) for s in sc.shapes)
    args = RenderFunction.serialize_scene(sc, spp, max_bounces, sampler_type=redner.SamplerType.sobol, device=dev, backend=redner,
                                          use_primary_edge_sampling=False, use_secondary_edge_sampling=False)
    t0 = time.time()
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    torch.cuda.synchronize(dev)
    t_scene = time.time() - t0
    img = torch.zeros(res, res, 3, device=dev)

    def forward():
        img.zero_()
        redner.render(u.scene, u.options, redner.float_ptr(img.data_ptr()), redner.float_ptr(0), None, redner.float_ptr(0), redner.float_ptr(0))
        torch.cuda.synchronize(dev)
    forward()                                   # warm-up (buffers)
    lib.rdr_trace_stats_enable(1, 0)
    trace_stats(reset=True)
    t0 = time.time()
    forward()
    t_fwd = time.time() - t0
    st = trace_stats()
    lib.rdr_trace_stats_enable(0, 1)
    trace_stats(reset=True)
    forward()
    cnt = trace_stats()
    lib.rdr_trace_stats_enable(0, 0)
    rays = cnt.closest_rays
    alg = rays * (RAY_BYTES + HIT_BYTES) + cnt.closest_nodes * NODE_BYTES + cnt.closest_wide_nodes * WIDE_NODE_BYTES + cnt.closest_tris * TRI_BYTES
    busy = st.closest_union_ms if st.closest_union_ms > 0 else st.closest_ms
    achieved = alg / (busy * 1e-3) / 1e9 if busy > 0 else 0.0
    return {'workload': 'bunny_box_subdivided (levels %d): %d triangles, %dx%d, %d spp forward, max_bounces %d' % (levels, tris, res, res, spp, max_bounces),
            'triangles': tris, 'hierarchy_bytes_estimate': tris * TRI_BYTES + 2 * (tris // 2) * NODE_BYTES,
            'mesh_s': t_mesh, 'scene_build_ms': t_scene * 1e3, 'forward_ms': t_fwd * 1e3, 'image_mean': float(img.mean()),
            'image_sha256': __import__('hashlib').sha256(img.cpu().numpy().tobytes()).hexdigest()[:16],
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'closest_launches': st.closest_launches, 'busy_ms': busy, 'mean_launch_ms': st.closest_ms / max(st.closest_launches, 1),
            'rays': rays, 'rays_per_s': rays / (busy * 1e-3) if busy > 0 else None,
            'nodes_per_ray': cnt.closest_nodes / max(rays, 1), 'wide_nodes_per_ray': cnt.closest_wide_nodes / max(rays, 1),
            'tris_per_ray': cnt.closest_tris / max(rays, 1), 'algorithmic_bytes': alg}


LARGE_LEVELS, LARGE_SPP = 4, 8


def large_hierarchy_counters(a):
    """Three rocprofv3 passes (FETCH_SIZE / WRITE_SIZE / SQ counters, separate passes as MI355X_MICROARCH.md prescribes; gfx950
    FETCH_SIZE counts 64 B per 128-B request: x 2) over the large-hierarchy leg: what the refilling closest-hit kernel really
    moves to and from HBM per launch there, its lane utilisation, the share of wave cycles spent waiting."""
    base = tempfile.mkdtemp(prefix='rdr_prof_large_', dir='/tmp')
    try:
        dirs = [_rocprof(['--pmc'] + c, a, os.path.join(base, tag), inner='--large-inner')
                for tag, c in (('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']), ('sq', SQ_COUNTERS))]
        if not all(dirs):
            return None
        out = {}
        for d in dirs:
            fs = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))
            if not fs:
                return None
            agg, launches = collections.defaultdict(float), set()
            for r in csv.DictReader(open(fs[0])):
                if 'trace_refill_kernel<false' in r['Kernel_Name']:
                    agg[r['Counter_Name']] += float(r['Counter_Value'])
                    launches.add(r['Dispatch_Id'])
            for k, v in agg.items():
                out[k] = v / max(len(launches), 1)
            out['launches_counted'] = len(launches)
        ds = glob.glob(os.path.join(dirs[2], '*', '*_kernel_trace.csv'))
        tot, n = 0.0, 0
        for r in (csv.DictReader(open(ds[0])) if ds else []):
            if 'trace_refill_kernel<false' in r['Kernel_Name']:
                tot += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
                n += 1
        ms = tot / n if n else None
        hbm = (out.get('FETCH_SIZE', 0.0) * 2.0 + out.get('WRITE_SIZE', 0.0)) * 1024.0
        return {'kernel': 'trace_refill_kernel<closest-hit> launches of the leg', 'launches_counted': out.get('launches_counted'),
                'mean_launch_ms_under_counters': ms, 'hbm_bytes_per_launch': hbm,
                'hbm_GBs': hbm / (ms * 1e-3) / 1e9 if ms else None, 'hbm_frac_measured': hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else None,
                'valu_lane_util': out['SQ_THREAD_CYCLES_VALU'] / (64.0 * out['SQ_ACTIVE_INST_VALU']) if out.get('SQ_ACTIVE_INST_VALU') else None,
                'wave_cycles_waiting_frac': out['SQ_WAIT_ANY'] / out['SQ_WAVE_CYCLES'] if out.get('SQ_WAVE_CYCLES') else None}
    finally:
        shutil.rmtree(base, ignore_errors=True)


def alone_leg(st, alg_bytes_step):
    """The same kernel in an extra untimed step of the short job with ONE sample worker and every stage on one stream: no other
    kernel beside a closest-hit launch.  Informative, not the headline.  `alg_bytes_step`: algorithmic bytes of that step."""
    ms = st.closest_ms / max(st.closest_launches, 1)
    achieved = alg_bytes_step / (st.closest_ms * 1e-3) / 1e9 if st.closest_ms > 0 else 0.0
    return {'mean_launch_ms': ms, 'launches': st.closest_launches, 'achieved': achieved, 'frac': achieved / HBM_PEAK_GBS,
            'note': 'untimed extra step, one sample worker, every stage on one stream'}


# ---- hardware counters, collected from inside the run ------------------------------------------------------------------
PROFILE_KERNELS = collections.OrderedDict([       # short name -> substring of the rocprofv3 kernel name
    ('trace_closest', 'trace_kernel<false, false'), ('trace_any', 'trace_kernel<true, false'),
    ('trace_closest_wide', 'trace_wide_kernel<false, false'), ('trace_any_wide', 'trace_wide_kernel<true, false'),
    ('trace_closest_refill', 'trace_refill_kernel<false'), ('trace_any_refill', 'trace_refill_kernel<true'),
    ('SecEdgePickHDescend', 'SecEdgePickHDescend'), ('SecEdgePickHLeaves', 'SecEdgePickHLeaves'), ('SecEdgePickH', 'SecEdgePickH'),
    ('SecEdgeSetup', 'SecEdgeSetup'), ('SecEdgeGatherN', 'SecEdgeGatherN'), ('AdjBounceScatter', 'AdjBounceScatter'),
    ('AdjBounceNee', 'AdjBounceNee'), ('BounceContrib', 'BounceContrib'), ('BounceSample', 'BounceSample'),
    ('AdjPrimary', 'AdjPrimary')])
SQ_COUNTERS = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_THREAD_CYCLES_VALU']


def _rocprof(extra, a, out_dir, env=None, inner='--inner'):
    """One rocprofv3 pass over `bench.py --inner` (32 spp, one forward+backward) or `--large-inner` (the large-hierarchy leg).
    Returns the output directory or None."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None
    cmd = [exe] + extra + ['--kernel-trace', '--output-format', 'csv', '-d', out_dir, '--', sys.executable,
                           os.path.join(ROOT, 'bench.py'), inner, '--res', str(a.res), '--max-bounces', str(a.max_bounces),
                           '--workload', a.workload]
    e = dict(os.environ)
    e.update(env or {})
    e['TMPDIR'] = '/tmp'
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        e.pop(k, None)
    try:
        subprocess.run(cmd, cwd='/tmp', env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
    except Exception:
        return None
    return out_dir


def _kernel_key(name):
    for short, pat in PROFILE_KERNELS.items():
        if pat in name:
            return short
    return None


def _read_counters(out_dir):
    fs = glob.glob(os.path.join(out_dir, '*', '*_counter_collection.csv'))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    if not fs:
        return agg, launches
    for r in csv.DictReader(open(fs[0])):
        k = _kernel_key(r['Kernel_Name'])
        if k is None:
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        launches[k].add(r['Dispatch_Id'])
    return agg, launches


def _read_durations(out_dir):
    """kernel-trace csv -> {short: (launches, mean ms)}"""
    fs = glob.glob(os.path.join(out_dir, '*', '*_kernel_trace.csv'))
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[0])):
        k = _kernel_key(r['Kernel_Name'])
        if k is None:
            continue
        tot[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
        n[k] += 1
    return {k: (n[k], tot[k] / n[k]) for k in n}


def under_profiler():
    """True when this process already runs under rocprofv3 / rocprof (its tool library is preloaded or configured): the
    in-run counter passes would nest a profiler inside a profiler, so they are skipped and `roofline.kernels` stays null."""
    env = os.environ
    if 'rocprof' in env.get('LD_PRELOAD', '').lower():
        return True
    return any(k.startswith(('ROCPROF', 'ROCP_TOOL', 'ROCPROFILER_')) for k in env)


def profile_kernels(a, batch=0):
    """Four rocprofv3 passes on a 2-spp forward+backward of the same workload: kernel durations (stages overlapped as in
    the benchmark, and each kernel on its own with RDR_NO_OVERLAP=1), SQ counters, FETCH_SIZE, WRITE_SIZE (separate passes,
    MI355X_MICROARCH.md "HBM": gfx950 FETCH_SIZE counts 64 B per 128-B request, hence x2).  Returns {kernel: {...}}."""
    base = tempfile.mkdtemp(prefix='rdr_prof_', dir='/tmp')
    out = {}
    try:
        # "alone": one stream of ONE sample worker (a second worker's chain would run beside it), the timed job's samples per launch
        one = {'RDR_NO_OVERLAP': '1', 'RDR_WORKERS': '1'}
        if batch > 0:
            one['RDR_BATCH'] = str(batch)
        d_over = _rocprof([], a, os.path.join(base, 'over'))
        d_alone = _rocprof([], a, os.path.join(base, 'alone'), one)
        d_sq = _rocprof(['--pmc'] + SQ_COUNTERS, a, os.path.join(base, 'sq'), one)
        d_f = _rocprof(['--pmc', 'FETCH_SIZE'], a, os.path.join(base, 'fetch'), one)
        d_w = _rocprof(['--pmc', 'WRITE_SIZE'], a, os.path.join(base, 'write'), one)
        if not (d_over and d_alone and d_sq and d_f and d_w):
            return None
        dur_over, dur_alone = _read_durations(d_over), _read_durations(d_alone)
        sq, sq_n = _read_counters(d_sq)
        fe, fe_n = _read_counters(d_f)
        wr, wr_n = _read_counters(d_w)
        for k in PROFILE_KERNELS:
            if k not in dur_alone or k not in sq:
                continue
            n, ms = dur_alone[k]
            v = sq[k]
            nl = max(len(sq_n[k]), 1)
            lane_ops = v['SQ_THREAD_CYCLES_VALU'] / nl                  # sum over vector instructions of their active lanes, per launch
            lane_util = v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_ACTIVE_INST_VALU']) if v['SQ_ACTIVE_INST_VALU'] else 0.0
            hbm = (fe[k]['FETCH_SIZE'] * 2.0 / max(len(fe_n[k]), 1) + wr[k]['WRITE_SIZE'] / max(len(wr_n[k]), 1)) * 1024.0
            out[k] = {
                'launches_per_sample': n / float(INNER_SPP),
                'mean_launch_ms_alone': ms,
                'mean_launch_ms_overlapped': dur_over.get(k, (0, None))[1],
                'valu_lane_util': lane_util,
                'valu_lane_ops_per_launch': lane_ops,
                'valu_frac_of_peak': lane_ops / (ms * 1e-3) / VALU_LANE_OPS_PEAK if ms > 0 else None,
                'wave_cycles_waiting_frac': v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'] if v['SQ_WAVE_CYCLES'] else None,
                'hbm_bytes_per_launch': hbm,
                'hbm_GBs': hbm / (ms * 1e-3) / 1e9 if ms > 0 else None,
                'hbm_frac_of_peak': hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
            }
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out or None


def inner_run(a):
    """Body of the rocprofv3 passes: one forward+backward of INNER_SPP spp (the sample batches of the timed job), nothing printed."""
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    from redner_amd import redner
    if 'RDR_POOL_CAP_MB' not in os.environ:
        redner.set_pool_cap_mb(BENCH_POOL_CAP_MB)
    prep = Prepared(redner, build_scene(a, dev, a.res), INNER_SPP, INNER_SPP, 0, a.max_bounces, dev)
    prep.step(0)
    torch.cuda.synchronize(dev)


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) started WITHOUT torch.distributed.run: become the launcher -- one rank per GPU on
    this node, rendezvous on 127.0.0.1 at a free port -- with the same arguments.  Rank 0 of the children prints the JSON line."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '1')
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--spp', type=int, default=256, help='samples per pixel of the whole job (sharded over the GPUs)')
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--max-bounces', type=int, default=None)
    ap.add_argument('--workload', default='bunny_box', choices=['bunny_box', 'living_room_standin', 'living_room_standin_envmap'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-self-check', action='store_true', help='skip the untimed sample-block self-check')
    ap.add_argument('--no-alone-leg', action='store_true', help='skip the extra single-stream step behind roofline.alone')
    ap.add_argument('--no-profile', action='store_true', help='skip the rocprofv3 counter passes behind roofline.kernels')
    ap.add_argument('--inner', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--large-inner', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-large-leg', action='store_true', help='skip the untimed large-hierarchy traversal leg behind roofline_large')
    ap.add_argument('--cpu-baseline-leg', action='store_true', help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.max_bounces is None:
        a.max_bounces = 6 if a.workload.startswith('living_room_standin') else 4
    if a.inner:
        return inner_run(a)
    if a.large_inner:
        torch.cuda.set_device(0)
        if 'RDR_POOL_CAP_MB' not in os.environ:
            from redner_amd import redner as _rd
            _rd.set_pool_cap_mb(BENCH_POOL_CAP_MB)
        large_hierarchy_leg(LARGE_LEVELS, LARGE_SPP)
        return None
    if a.cpu_baseline_leg:
        return cpu_baseline_leg(a)

    if a.gpus > 1 and 'RANK' not in os.environ:
        return self_launch(a)          # `python bench.py --gpus N` without a launcher: start the N ranks ourselves
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    # RDR_BENCH_SHARE_GPU=1 (rehearsal of the multi-rank path on a box with fewer GPUs than ranks: gloo, ranks share devices;
    # the number it prints is not a measurement)
    share = os.environ.get('RDR_BENCH_SHARE_GPU') == '1'
    if not share and local_rank >= torch.cuda.device_count():
        raise SystemExit('bench.py: local rank %d but %d GPU(s) visible (RDR_BENCH_SHARE_GPU=1 rehearses the multi-rank path '
                         'on fewer GPUs)' % (local_rank, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda:%d' % dev_index)
    under_launcher = 'RANK' in os.environ and 'MASTER_PORT' in os.environ      # torch.distributed.run, also with one rank
    if world > 1 or under_launcher:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
        assert dist.get_world_size() == world
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE is %d: launch `python bench.py --gpus N` (it starts its own ranks) '
                         'or torch.distributed.run --nproc-per-node N bench.py --gpus N' % (a.gpus, world))
    if not share and world > torch.cuda.device_count():
        raise SystemExit('bench.py: %d ranks but %d GPU(s) visible' % (world, torch.cuda.device_count()))
    assert a.spp % world == 0, '--spp must be divisible by the number of GPUs'
    spp_rank = a.spp // world

    from redner_amd import redner
    # The library parks at most 8 GiB of buffers between calls by default (it may share the device with torch's allocator).  This
    # process owns its GPU: it raises the bound -- stated in the line (`config.pool_cap_mb`) -- so that two sample workers keep
    # 8-sample batches of the 1024 x 1024 frame resident (32 GiB of the 288; RDR_POOL_CAP_MB overrides.  Measured on the final
    # tree, profiles/r6_notes.md: 8 GiB 72.4, 16 GiB 74.1, 32 GiB 75.5, 64 GiB 75.7 Msamples/s when it was measured; 77.7 / 84.3 / 86.0 at the end of round 6).
    if 'RDR_POOL_CAP_MB' not in os.environ:
        redner.set_pool_cap_mb(BENCH_POOL_CAP_MB)
    pool_cap_mb = redner.get_pool_cap_mb()
    prep = Prepared(redner, build_scene(a, dev, a.res), spp_rank, a.spp, rank * spp_rank, a.max_bounces, dev)
    # REDNER_AMD_FORCE_COLLECTIVE=1 under a launcher with ONE rank: the collective runs all the same (RCCL communicator,
    # all_gather_into_tensor on the device bucket, fixed-order sum, unpacking) and its result must equal what went in, bit for
    # bit -- the multi-rank data path executed on a one-GPU box (tests/test_rccl_gpu.py)
    forced = world == 1 and dist.is_initialized() and os.environ.get('REDNER_AMD_FORCE_COLLECTIVE') == '1'
    coll = {'calls': 0, 'seconds': 0.0, 'bytes_per_call': 0, 'bit_identical_to_local': None}

    def reduce_all():
        # the image and every gradient tensor in ONE bucket, ONE collective per step: all_gather + fixed-order sum
        # (bit-reproducible), see distributed.py
        if world == 1 and not forced:
            return
        from redner_amd.distributed import _all_gather_sum_many
        both = [prep.img] + list(prep.grads)
        before = [t.clone() for t in both] if forced else None
        t1 = time.time()
        red = _all_gather_sum_many(both, dist.group.WORLD, force=forced)
        for t, r in zip(both, red):
            t.copy_(r)
        if not share:
            torch.cuda.synchronize(dev)
        coll['seconds'] += time.time() - t1
        coll['calls'] += 1
        coll['bytes_per_call'] = sum(t.numel() * t.element_size() for t in both)
        if forced:
            same = all(torch.equal(x, y) for x, y in zip(before, both))
            coll['bit_identical_to_local'] = same if coll['bit_identical_to_local'] is None else (coll['bit_identical_to_local'] and same)

    from redner_amd import _capi
    lib = _capi.lib()
    for i in range(a.warmup):
        prep.step(i)
        reduce_all()
    lib.rdr_trace_stats_enable(1, 0)
    trace_stats(reset=True)
    dbg0 = _capi.DebugCounters()
    lib.rdr_debug_counters_get(ctypes.byref(dbg0))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.time()
    render_s = [0.0]                  # this rank's time inside its two render() calls (they return synchronised)
    for i in range(a.steps):
        t1 = time.time()
        prep.step(a.warmup + i)
        render_s[0] += time.time() - t1
        reduce_all()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    st = trace_stats()
    per_rank_ms = [dt / a.steps * 1e3]
    if world > 1:
        # every rank's own clock around the same K steps (they end at the same barrier: what differs is where a rank waited --
        # inside its renders or in the collective); the reported time is the maximum
        mine = torch.tensor([dt, render_s[0]], dtype=torch.float64, device=torch.device('cpu') if share else dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(t[0]) / a.steps * 1e3 for t in every]
        per_rank_render_ms = [float(t[1]) / a.steps * 1e3 for t in every]
        dt = max(float(t[0]) for t in every)
    samples = a.res * a.res * a.spp * a.steps
    value = samples / dt / 1e6
    dbg = _capi.DebugCounters()
    lib.rdr_debug_counters_get(ctypes.byref(dbg))
    timed_batch, timed_workers = int(dbg.last_batch_samples), int(dbg.last_workers)      # how the library scheduled the timed steps
    timed_mallocs = int(dbg.device_mallocs) - int(dbg0.device_mallocs)                  # hipMalloc calls inside the timed region (0: everything came from the buffer cache)

    # untimed, world > 1 (or the forced one-rank collective): the north star's rule "bit-identical sum at 1 vs N GPUs", checked
    # on the hardware the job ran on -- a short job of one sample per rank, gathered and summed as in the timed steps, against
    # rank 0 rendering the same `world` sample blocks itself and summing them in rank order (must be equal BIT FOR BIT: image
    # and every gradient tensor), and against one plain call of `world` samples (fp32 summation order: ~1e-7)
    sharded_check = None
    if world > 1 or forced:
        from redner_amd.distributed import _all_gather_sum_many
        part = Prepared(redner, build_scene(a, dev, a.res), 1, world, rank, a.max_bounces, dev)
        part.step(0)
        both = [part.img] + list(part.grads)
        red = [r.clone() for r in _all_gather_sum_many(both, dist.group.WORLD, force=forced)]
        if rank == 0:
            acc = None
            for b in range(world):
                blk = Prepared(redner, build_scene(a, dev, a.res), 1, world, b, a.max_bounces, dev)
                blk.step(0)
                torch.cuda.synchronize(dev)
                mine = [blk.img] + list(blk.grads)
                if acc is None:
                    acc = [t.clone() for t in mine]
                else:
                    for x, y in zip(acc, mine):
                        x += y
                del blk
            one = Prepared(redner, build_scene(a, dev, a.res), world, world, 0, a.max_bounces, dev)
            one.step(0)
            torch.cuda.synchronize(dev)
            whole = [one.img] + list(one.grads)
            sharded_check = {
                'job': '%s %dx%d, %d spp fwd+bwd as %d one-sample blocks, one per rank' % (a.workload, a.res, a.res, world, world),
                'image_bit_identical_to_blocks_on_one_device': bool(torch.equal(red[0], acc[0])),
                'gradients_bit_identical_to_blocks_on_one_device': bool(all(torch.equal(x, y) for x, y in zip(red[1:], acc[1:]))),
                'image_rel_l2_vs_one_call': _rel_l2(red[0], whole[0]),
                'worst_gradient_rel_l2_vs_one_call': max((_rel_l2(x, y) for x, y in zip(red[1:], whole[1:]) if float(y.abs().sum()) > 0),
                                                         default=0.0)}
            del one
        del part

    # everything below is untimed and works on a short job (the counters do not depend on the sample count)
    # (32 spp: enough samples for the library to form the same sample batches as in the timed job, so that launches of the
    #  counted job and of the timed job cover the same number of rays)
    short_spp = min(spp_rank, 32)
    short = Prepared(redner, build_scene(a, dev, a.res), short_spp, short_spp, 0, a.max_bounces, dev)
    scene_build_warm_ms = short.scene_build_s * 1e3      # second Scene of the process: allocator, staging buffer, topology caches warm
    # instrumented traversal variant: node / triangle records per launch
    lib.rdr_trace_stats_enable(0, 1)
    trace_stats(reset=True)
    short.step(a.warmup)
    torch.cuda.synchronize(dev)
    cnt = trace_stats()
    lib.rdr_trace_stats_enable(0, 0)

    # ONE stream: the traversal kernel's launch duration without a neighbour on the GPU (in the timed region the
    # shadow-ray launch of the same bounce runs beside every closest-hit launch)
    alone = one_chain = None
    if not a.no_alone_leg:
        short.u.options.tuning.flags |= _capi.TUNE_NO_OVERLAP          # rdr_tuning: every stage on the calling stream ...
        short.u.options.tuning.workers = 1                              # ... of ONE host thread, the timed job's samples per launch
        short.u.options.tuning.batch_samples = max(timed_batch, 1)
        lib.rdr_trace_stats_enable(1, 0)
        trace_stats(reset=True)
        short.step(a.warmup)
        torch.cuda.synchronize(dev)
        alone = trace_stats()
        lib.rdr_trace_stats_enable(0, 0)
        # ... and ONE worker with the stages overlapped as usual (shadow-ray launch beside every closest-hit launch, picks and adjoints
        # on their side streams): the schedule of round 5, whose bench line reported exactly this quotient (bytes / mean launch
        # duration = bytes / union when one chain of launches is in flight) -- the like-for-like figure across rounds
        short.u.options.tuning.flags &= ~_capi.TUNE_NO_OVERLAP
        short.step(a.warmup)                                            # (the schedule's first step: untimed)
        torch.cuda.synchronize(dev)
        lib.rdr_trace_stats_enable(1, 0)
        trace_stats(reset=True)
        short.step(a.warmup + 1)
        short.step(a.warmup + 2)
        torch.cuda.synchronize(dev)
        one_chain = trace_stats()
        lib.rdr_trace_stats_enable(0, 0)
    del short

    out = None
    if rank == 0:
        # Accounting is per SAMPLE, not per launch: the instrumented pass counts the records of the short job exactly; a sample of
        # the timed job draws the same kind of rays, so the timed region's algorithmic bytes are bytes-per-sample x its samples --
        # whatever the launch shapes were (forward launches cover 16 samples, gradient launches 8, the single-chain leg 8 and 8).
        rays = cnt.closest_rays
        alg_bytes = rays * (RAY_BYTES + HIT_BYTES) + cnt.closest_nodes * NODE_BYTES + cnt.closest_wide_nodes * WIDE_NODE_BYTES + cnt.closest_tris * TRI_BYTES
        alg_bytes_sample = alg_bytes / float(short_spp)
        rays_sample = rays / float(short_spp)
        alg_bytes_timed = alg_bytes_sample * spp_rank * a.steps          # this rank's timed region
        alg_bytes_launch = alg_bytes_timed / max(st.closest_launches, 1)
        mean_launch_ms = st.closest_ms / max(st.closest_launches, 1)
        # Schedule-invariant form (round 6): the algorithmic bytes of ALL closest-hit launches of the timed region over the time
        # during which at least one of them was in flight (union of their [start, end] event intervals).  With one chain of
        # launches that is bytes over mean launch duration; with two sample workers two launches trace side by side -- each
        # takes longer, the same rays are traced at the same total rate -- and the per-launch quotient would halve for no reason.
        busy_ms = st.closest_union_ms if st.closest_union_ms > 0 else st.closest_ms
        achieved = alg_bytes_timed / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0
        achieved_per_launch = alg_bytes_timed / (st.closest_ms * 1e-3) / 1e9 if st.closest_ms > 0 else 0.0
        rays_per_launch = rays_sample * spp_rank * a.steps / max(st.closest_launches, 1)
        prof = None
        if world == 1 and not a.no_profile and not under_profiler():
            try:
                # the passes run in child processes: what this process parks (up to 32 GiB, > 10 % of the device) would make the
                # library size their batches for a shared device (render.cpp: memory_held_by_others)
                redner.trim_cache()
                torch.cuda.empty_cache()
                prof = profile_kernels(a, timed_batch)
            except Exception as e:      # the counters must never take the throughput number down with them
                prof = {'error': repr(e)}
        # counters of the closest-hit kernel that does most of the work: the refilling form on the large incoherent queues of the
        # default job (trace.hip), else the plain one
        tc = None
        if isinstance(prof, dict):
            cands = [prof[k] for k in ('trace_closest_refill', 'trace_closest') if k in prof]
            tc = max(cands, key=lambda v: v['launches_per_sample'] * v['mean_launch_ms_alone']) if cands else None
        out = {
            'metric': 'Msamples/s fwd+bwd', 'value': value, 'unit': 'Msamples/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s %dx%d, max_bounces %d, Sobol, %d spp fwd+bwd per step (%s gradients, primary+secondary '
                                   'edge sampling), sharded by sample index over %d GPU(s): %d spp per GPU'
                                   % (a.workload, a.res, a.res, a.max_bounces, a.spp,
                                      'camera-pose' if a.workload.startswith('living_room_standin') else 'vertex', world, spp_rank),
                       'resolution': [a.res, a.res], 'spp': a.spp, 'spp_per_gpu': spp_rank, 'max_bounces': a.max_bounces,
                       'parallelism': 'sample-sharded x%d' % world, 'world_size': world, 'pool_cap_mb': pool_cap_mb,
                       'samples_per_launch': timed_batch, 'sample_workers': timed_workers, 'device_mallocs_in_timed_region': timed_mallocs,
                       'schedule': os.environ.get('RDR_BENCH_SCHEDULE_NOTE', 'library default: two sample workers, batches as large as the buffer cache holds')},
            # per rank: wall time per step, and the part of it spent inside render() (the rest: the one collective + waiting
            # for the slowest rank) -- so that a scaling curve explains itself
            'per_rank_ms_per_step': per_rank_ms,
            'per_rank_render_ms_per_step': per_rank_render_ms if world > 1 else [render_s[0] / max(a.steps, 1) * 1e3],
            # Scene incl. its edge structures, synchronised (the library builds those beside the caller: a render loop does not
            # wait here); first Scene of the process / a later one with the same connectivity
            'scene_build_ms': prep.scene_build_s * 1e3, 'scene_build_warm_ms': scene_build_warm_ms,
            # the one collective per step (image + every gradient tensor in one bucket): which backend carried it, how long a
            # call took on rank 0 (incl. waiting for the slowest rank); null when no process group exists (plain N = 1 run)
            'sharded_check': sharded_check,
            'collective': ({'backend': dist.get_backend(), 'world_size': world, 'forced_at_world_1': forced, 'calls': coll['calls'],
                            'bytes_per_rank_per_call': coll['bytes_per_call'],
                            'ms_per_call': coll['seconds'] / max(coll['calls'], 1) * 1e3,
                            'bit_identical_to_local': coll['bit_identical_to_local']} if dist.is_initialized() else None),
            'roofline': {'kernel': 'closest-hit traversal (trace_kernel / trace_refill_kernel, all closest-hit launches of the timed region)', 'bound': 'hbm', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': tc['hbm_bytes_per_launch'] if tc else None,
                         'definition': 'algorithmic bytes of all closest-hit launches of the timed region / time with at least one of '
                                       'them in flight (union of HIP-event intervals on the launch streams)',
                         'busy_ms_per_step': busy_ms / max(a.steps, 1), 'sum_of_launch_ms_per_step': st.closest_ms / max(a.steps, 1),
                         'launch_overlap': st.closest_ms / busy_ms if busy_ms > 0 else None,
                         'per_launch': {'achieved': achieved_per_launch, 'frac': achieved_per_launch / HBM_PEAK_GBS,
                                        'note': 'bytes of one launch / its own duration, with whatever shares the GPU beside it'},
                         'mean_launch_ms': mean_launch_ms, 'launches_per_step': st.closest_launches / max(a.steps, 1),
                         'rays_per_launch': rays_per_launch,
                         'rays_per_s': rays_sample * spp_rank * a.steps / (busy_ms * 1e-3) if busy_ms > 0 else None,
                         'nodes_per_ray': cnt.closest_nodes / max(rays, 1), 'wide_nodes_per_ray': cnt.closest_wide_nodes / max(rays, 1), 'tris_per_ray': cnt.closest_tris / max(rays, 1),
                         'algorithmic_bytes_per_launch': alg_bytes_launch, 'algorithmic_bytes_per_sample': alg_bytes_sample,
                         'hbm_frac_measured': tc['hbm_frac_of_peak'] if tc else None,
                         'valu_lane_util': tc['valu_lane_util'] if tc else None,
                         'closest_hit_busy_share_of_step': busy_ms / (dt * 1e3), 'any_hit_busy_share_of_step': (st.any_union_ms or st.any_ms) / (dt * 1e3),
                         'alone': alone_leg(alone, alg_bytes_sample * short_spp) if alone is not None else None,
                         'one_chain': (dict(alone_leg(one_chain, alg_bytes_sample * short_spp * 2),
                                            note='two untimed extra steps, ONE sample worker, stages overlapped as in the timed region: the '
                                                 'schedule and the quotient of the round-5 bench line (0.72 there)')
                                       if one_chain is not None else None),
                         'kernels': prof,
                         'note': 'frac = algorithmic bytes (SURVEY.md 8d) over launch time: the 1 MB hierarchy is L2-resident, '
                                 'so this is an L2-served rate; hbm_frac_measured is what reaches HBM (counters), and the '
                                 'kernels are bound by vector-ALU issue at valu_lane_util (DESIGN.md section 3)'},
        }
        # the same closest-hit kernel on a hierarchy that does NOT fit the L2 (bunny tessellated to 3.7 M triangles, ~250 MB of records):
        # what the design does when traversal really goes to the MALL / HBM
        if world == 1 and not a.no_large_leg and a.workload == 'bunny_box':
            try:
                big = large_hierarchy_leg(LARGE_LEVELS, LARGE_SPP)
                if not a.no_profile and not under_profiler():
                    redner.trim_cache()
                    torch.cuda.empty_cache()
                    big['counters'] = large_hierarchy_counters(a)
                    if big['counters'] and big['counters'].get('launches_counted'):
                        big['traffic'] = big['counters']['hbm_bytes_per_launch']
                        big['hbm_frac_measured'] = big['counters']['hbm_frac_measured']
                big['note'] = ('latency-bound: dependent node fetches miss the L2, three (40-entry stack) or four (32-entry) waves per '
                               'SIMD hide little of it; the HBM traffic the counters see is well under the algorithmic bytes '
                               '(DESIGN.md section 3)')
                out['roofline_large'] = big
            except Exception as e:
                out['roofline_large'] = {'error': repr(e)}
        if world == 1 and not a.no_self_check:
            try:
                out['self_check'] = sharded_self_check(a, redner, dev)
            except Exception as e:
                out['self_check'] = {'error': repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(a, redner, dev)
            except Exception as e:   # the baseline must never take the GPU number down with it
                out['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
