#!/usr/bin/env python3
"""Traversal kernel A/B on realistic ray queues (GPU box): camera rays of bunny_box at RES x RES in pixel order, then two
generations of cosine-distributed bounce rays from the hit points (compacted like the renderer's live-lane lists), and shadow
rays towards the light.  Each VARIANT (a set of environment switches read once per process) runs in its own subprocess, times
rdr_scene_trace per queue with HIP events and writes its hit ids; every variant must return the ids of the first one.

  python tools/trace_ab.py [RES] -- "" "RDR_TRACE_BUDGET=16,32" ..."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]


def make_queues(res, dev, trace):
    import torch
    import scenes
    sc = scenes.bunny_box(dev, resolution=(res, res), vertex_grad=False)
    cam = sc.camera
    if cam.cam_to_world is not None:
        c2w = cam.cam_to_world.to(dev)
    else:                                            # look-at frame (redner: x = up x dir ... left-handed; the sign only mirrors the image)
        pos, la, up = cam.position.to(dev), cam.look_at.to(dev), cam.up.to(dev)
        zd = (la - pos) / (la - pos).norm()
        xd = torch.linalg.cross(up, zd); xd = xd / xd.norm()
        yd = torch.linalg.cross(zd, xd)
        c2w = torch.eye(4, device=dev)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = xd, yd, zd, pos
    K = cam.intrinsic_mat.to(dev)
    yy, xx = torch.meshgrid(torch.arange(res, device=dev), torch.arange(res, device=dev), indexing='ij')
    g = torch.Generator(device=dev); g.manual_seed(3)
    u = (xx.reshape(-1) + torch.rand(res * res, device=dev, generator=g)) / res
    v = (yy.reshape(-1) + torch.rand(res * res, device=dev, generator=g)) / res
    aspect = 1.0
    ndc = torch.stack([(u - 0.5) * 2.0, (v - 0.5) * -2.0 / aspect, torch.ones_like(u)], 1)
    dirs = ndc @ torch.linalg.inv(K).T
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    d_w = dirs @ c2w[:3, :3].T
    o_w = c2w[:3, 3].expand_as(d_w)
    verts = [s.vertices.to(dev) for s in sc.shapes]
    inds = [s.indices.to(dev).long() for s in sc.shapes]
    light = sc.shapes[sc.area_lights[0].shape_id]
    lv = light.vertices.to(dev)[light.indices.to(dev).long()]          # [T,3,3]

    def rays_of(o, d, tmin=1e-3, tmax=float('inf')):
        r = torch.zeros(o.shape[0], 8, device=dev)
        r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = o, tmin, d, tmax
        return r.contiguous()

    def hit_points(o, d, hits):
        p = torch.zeros_like(o); n = torch.zeros_like(o)
        for s in range(len(verts)):
            m = hits[:, 0] == s
            if not m.any():
                continue
            tri = verts[s][inds[s][hits[m, 1].long()]]
            a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
            nn = torch.linalg.cross(b - a, c - a)
            nn = nn / nn.norm(dim=1, keepdim=True)
            t = ((a - o[m]) * nn).sum(1) / (d[m] * nn).sum(1)
            p[m] = o[m] + t[:, None] * d[m]
            nn = torch.where(((d[m] * nn).sum(1) > 0)[:, None], -nn, nn)
            n[m] = nn
        return p, n

    queues = []
    o, d = o_w, d_w
    for gen in range(3):
        q = rays_of(o, d)
        hits = trace(sc, q, 0)
        queues.append(('closest%d' % gen, q, 0))
        live = hits[:, 0] >= 0
        o, d, hits = o[live], d[live], hits[live]
        p, n = hit_points(o, d, hits)
        # shadow rays towards a random point of the light
        k = torch.randint(0, lv.shape[0], (p.shape[0],), device=dev, generator=g)
        b1 = torch.rand(p.shape[0], device=dev, generator=g); b2 = torch.rand(p.shape[0], device=dev, generator=g)
        s1 = b1.sqrt()
        lp = (1 - s1)[:, None] * lv[k, 0] + (s1 * (1 - b2))[:, None] * lv[k, 1] + (s1 * b2)[:, None] * lv[k, 2]
        sd = lp - p
        dist = sd.norm(dim=1)
        sq = rays_of(p, sd / dist[:, None], 1e-3, 1.0)
        sq[:, 7] = dist * (1 - 1e-3)
        sq[(sd * n).sum(1) <= 0, 7] = -1.0          # light behind the surface: not traced (dead slot)
        queues.append(('any%d' % gen, sq.contiguous(), 1))
        # cosine-distributed bounce
        r1 = torch.rand(p.shape[0], device=dev, generator=g); r2 = torch.rand(p.shape[0], device=dev, generator=g) * 6.2831853
        t1 = torch.linalg.cross(n, torch.tensor([0.577, 0.577, 0.577], device=dev).expand_as(n))
        t1 = t1 / t1.norm(dim=1, keepdim=True)
        t2 = torch.linalg.cross(n, t1)
        sr = r1.sqrt()
        nd = (sr * r2.cos())[:, None] * t1 + (sr * r2.sin())[:, None] * t2 + (1 - r1).sqrt()[:, None] * n
        o, d = p, nd / nd.norm(dim=1, keepdim=True)
    return sc, queues


def worker(res, out_path, reps=int(os.environ.get('TRACE_AB_REPS', '20'))):
    import torch
    from redner_amd import _capi, redner
    from redner_amd.render_pytorch import RenderFunction
    dev = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
    if dev.type == 'cpu':
        _capi.load(os.path.join(ROOT, 'tests', 'hostsim', '_build', 'libredner_hostsim.so')); reps = 0
    lib = _capi.lib()
    handle = {}

    def trace(sc, q, any_hit):
        if 'u' not in handle:
            args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=redner.SamplerType.sobol, device=dev)
            handle['u'] = RenderFunction.unpack_args((1, 2), args[0], args[1:])
        hits = torch.zeros(q.shape[0], 2, dtype=torch.int32, device=dev)
        assert lib.rdr_scene_trace(handle['u'].scene._handle, q.data_ptr(), hits.data_ptr(), q.shape[0], any_hit) == 0
        return hits

    sc, queues = make_queues(res, dev, trace)
    result = {}
    line = []
    lib.rdr_trace_stats_enable(1, 0)
    for name, q, any_hit in queues:
        hits = trace(sc, q, any_hit)
        result[name] = hits.cpu().numpy() if not any_hit else (hits[:, 0] >= 0).cpu().numpy()
        for _ in range(3 if reps else 0):
            trace(sc, q, any_hit)
        lib.rdr_trace_stats_reset()
        for _ in range(reps):
            trace(sc, q, any_hit)
        st = _capi.TraceStats()
        lib.rdr_trace_stats_get(ctypes.byref(st))
        ms = (st.any_ms if any_hit else st.closest_ms) / max(reps, 1)
        line.append('%s %7d rays (%d hit) %.4f ms' % (name, q.shape[0], int((hits[:, 0] >= 0).sum()), ms))
    # records per ray (the instrumented kernels), all closest-hit / all any-hit queues together
    lib.rdr_trace_stats_enable(0, 1)
    lib.rdr_trace_stats_reset()
    for name, q, any_hit in queues:
        trace(sc, q, any_hit)
    st = _capi.TraceStats()
    lib.rdr_trace_stats_get(ctypes.byref(st))
    lib.rdr_trace_stats_enable(0, 0)
    line.append('per ray: closest %.2f node records %.2f tris; any %.2f / %.2f' % (
        (st.closest_nodes + st.closest_wide_nodes) / max(st.closest_rays, 1), st.closest_tris / max(st.closest_rays, 1),
        (st.any_nodes + st.any_wide_nodes) / max(st.any_rays, 1), st.any_tris / max(st.any_rays, 1)))
    np.savez(out_path, **result)
    print(' | '.join(line), flush=True)


def main():
    if '--worker' in sys.argv:
        return worker(int(sys.argv[2]), sys.argv[3])
    args = sys.argv[1:]
    res = 1024
    if args and args[0] != '--':
        res = int(args.pop(0))
    variants = args[1:] if args and args[0] == '--' else ['']
    base = None
    for k, v in enumerate(variants):
        env = dict(os.environ)
        for kv in v.split():
            a, b = kv.split('=', 1)
            env[a] = b
        out = '/tmp/trace_ab_%d.npz' % k
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', str(res), out], env=env, capture_output=True, text=True,
                               timeout=float(os.environ.get('TRACE_AB_TIMEOUT', '120')))
        except subprocess.TimeoutExpired:
            print('[%s] TIMED OUT' % v, flush=True)
            continue
        print('[%s]' % (v or 'default'), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED: ' + r.stderr[-600:], flush=True)
        if r.returncode != 0:
            continue
        z = np.load(out)
        if base is None:
            base = {n: z[n] for n in z.files}
        else:
            bad = [n for n in z.files if not np.array_equal(z[n], base[n])]
            print('    hits equal to the first variant:', 'yes' if not bad else 'NO: ' + ', '.join(bad), flush=True)


if __name__ == '__main__':
    main()
