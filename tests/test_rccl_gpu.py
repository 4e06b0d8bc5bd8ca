"""The RCCL path executed on the ONE GPU of the test box.

A one-rank process group has nothing to exchange, so `redner_amd.distributed` returns early at world 1 -- and until round 6 not
one byte had gone through `init_process_group('nccl')` + `all_gather_into_tensor`.  REDNER_AMD_FORCE_COLLECTIVE=1 makes the
collective run all the same: RCCL communicator set-up on cuda:0, the flat device bucket (image + every gradient tensor)
gathered, summed in fixed rank order, unpacked.  The sum of one part is the part: everything must come back bit for bit.

 * bench.py under `torch.distributed.run --nproc-per-node 1` (the `under_launcher` branch the driver's N > 1 runs take);
 * `render_sharded` (the autograd surface) in a one-rank nccl group against the plain RenderFunction in the same process;
 * an 8-rank rehearsal of the driver's 8-GPU command on the one GPU (gloo: RCCL refuses two ranks on one device), 8 x 2 spp of
   the benchmark frame: the self-launch, per-rank device binding, per-rank pool / thread caps at the real world size, and
   bench.py's own `sharded_check` (gathered image == the same 8 blocks rendered on one device, bit for bit).

With RDR_RCCL_LOG=<dir> the NCCL_DEBUG=INFO output of the first two is kept (profiles/r6_rccl_one_rank.log)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR', 'REDNER_AMD_LIBM', 'REDNER_AMD_LIB'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.update(extra)
    return env


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _keep_log(name, text):
    d = os.environ.get('RDR_RCCL_LOG')
    if d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), 'w') as f:
            f.write(text)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_one_rank_under_launcher_runs_the_collective_over_rccl():
    env = _clean_env(REDNER_AMD_FORCE_COLLECTIVE='1', NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,COLL')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--spp', '32', '--steps', '1',
           '--warmup', '0', '--no-profile', '--no-cpu-baseline', '--no-self-check', '--no-alone-leg']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    _keep_log('bench_one_rank_nccl.log', ' '.join(cmd) + '\n--- stdout\n' + r.stdout + '\n--- stderr\n' + r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    c = j['collective']
    assert c['backend'] == 'nccl' and c['world_size'] == 1 and c['forced_at_world_1'] and c['calls'] == 1
    assert c['bytes_per_rank_per_call'] >= 1024 * 1024 * 3 * 4           # the image alone is 12.6 MB
    assert c['bit_identical_to_local'] is True
    s = j['sharded_check']
    assert s['image_bit_identical_to_blocks_on_one_device'] and s['image_rel_l2_vs_one_call'] == 0.0
    assert s['worst_gradient_rel_l2_vs_one_call'] < 1e-6
    assert j['n_gpus'] == 1 and j['value'] > 0
    both = r.stdout + r.stderr
    assert 'NCCL INFO' in both, 'RCCL did not report its initialisation (NCCL_DEBUG=INFO)'
    assert 'AllGather' in both or 'Init COMPLETE' in both or 'comm' in both


WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, %(root)r + '/tests']
import numpy as np, torch, torch.distributed as dist
from redner_amd import _capi
_capi.load()
assert _capi.is_product_library(), _capi.library_path()
from redner_amd import redner
from redner_amd.render_pytorch import RenderFunction
from redner_amd.distributed import render_sharded
import scenes
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
dist.init_process_group('nccl', device_id=dev)
assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1

def run(fn):
    sc = scenes.bunny_box(dev, resolution=(96, 96))
    for l in sc.area_lights:
        l.intensity.requires_grad_(True)
    sc.camera.position.requires_grad_(True)
    args = RenderFunction.serialize_scene(sc, 8, 4, sampler_type=redner.SamplerType.sobol, device=dev)
    img = fn(args)
    img.sum().backward()
    torch.cuda.synchronize()
    out = {'image': img.detach().cpu().numpy(), 'light': sc.area_lights[0].intensity.grad.cpu().numpy(),
           'cam': sc.camera.position.grad.cpu().numpy()}
    out.update({'g%%d' %% i: s.vertices.grad.cpu().numpy() for i, s in enumerate(sc.shapes) if s.vertices.grad is not None})
    return out

calls = []
real = dist.all_gather_into_tensor
def counted(out, inp, group=None, async_op=False):
    calls.append((inp.device.type, inp.numel()))
    return real(out, inp, group=group, async_op=async_op)
dist.all_gather_into_tensor = counted
a = run(lambda args: render_sharded(3, args))
dist.all_gather_into_tensor = real
b = run(lambda args: RenderFunction.apply(3, *args))
np.savez(%(out)r, calls=np.asarray([n for _, n in calls]), on_device=all(d == 'cuda' for d, _ in calls),
         **{'a_' + k: v for k, v in a.items()}, **{'b_' + k: v for k, v in b.items()})
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_render_sharded_in_a_one_rank_nccl_group_equals_the_plain_render(gpu_backend, tmp_path):
    out = str(tmp_path / 'rccl.npz')
    script = tmp_path / 'worker_rccl.py'
    script.write_text(WORKER % {'root': ROOT, 'out': out})
    env = _clean_env(REDNER_AMD_FORCE_COLLECTIVE='1', NCCL_DEBUG='INFO', REDNER_AMD_LIBM=os.environ.get('REDNER_AMD_LIBM', ''))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    _keep_log('render_sharded_one_rank_nccl.log', r.stdout + '\n--- stderr\n' + r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    # two collectives: the image after forward (96 x 96 x 3 floats), one bucket with every gradient tensor after backward
    assert len(z['calls']) == 2 and int(z['calls'][0]) == 96 * 96 * 3 and bool(z['on_device'])
    assert np.array_equal(z['a_image'], z['b_image'])
    for k in z.files:
        if k.startswith('a_') and k != 'a_image':
            g, m = z['b_' + k[2:]].astype(np.float64), z[k].astype(np.float64)
            # two renders of the same samples: the fp64 atomics of the gradient accumulators commute up to their last bit
            assert np.linalg.norm(m - g) <= 1e-6 * max(np.linalg.norm(g), 1e-30), k


@pytest.mark.gpu
def test_eight_rank_rehearsal_on_one_gpu():
    """The driver's 8-GPU command line at the real world size, ranks sharing the one GPU (gloo): 8 x 2 spp of the 1024 x 1024
    frame.  Not a measurement -- a rehearsal of everything around the kernels: self-launch, rendezvous, rank -> sample block,
    per-rank pool and host-thread caps, the gathered sum against the same blocks on one device."""
    free, total = torch.cuda.mem_get_info(0)
    if free < 80 * 2 ** 30:
        pytest.skip('needs ~60 GB of free device memory for 8 ranks on one GPU')
    env = _clean_env(RDR_BENCH_SHARE_GPU='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--spp', '16', '--steps', '1', '--warmup', '0',
                        '--no-profile', '--no-cpu-baseline', '--no-self-check', '--no-alone-leg'], env=env, timeout=1500,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    _keep_log('bench_eight_ranks_shared_gpu.log', r.stdout + '\n--- stderr\n' + r.stderr[-20000:])
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j['n_gpus'] == 8 and j['config']['world_size'] == 8 and j['config']['spp_per_gpu'] == 2
    assert len(j['per_rank_ms_per_step']) == 8 and j['value'] > 0
    assert j['collective']['backend'] == 'gloo' and j['collective']['calls'] == 1
    s = j['sharded_check']
    assert s['image_bit_identical_to_blocks_on_one_device'], s
    assert s['image_rel_l2_vs_one_call'] < 2e-6 and s['worst_gradient_rel_l2_vs_one_call'] < 1e-4, s
