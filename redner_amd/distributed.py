"""Multi-GPU rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm),
the Sobol' samples of one frame split into contiguous blocks, one block per rank.

The reference has no multi-device path; this is new functionality whose
oracle is "sum of the shards == single-device result".  Why sample blocks and not pixel tiles:
a sample's random numbers depend only on (sample index, dimension, per-pixel scramble), live-lane
compaction and the Sobol' dimension bookkeeping are whole-frame properties, so every rank
reproduces exactly the samples a single device would have drawn.
Two exceptions: the PCG sampler, whose state depends on the earlier samples, and scenes with mip-mapped
textures rendered with primary edge sampling, where the reference's scratch carries ray differentials from sample to sample
-- a block that starts at sample k > 0 draws a few of its edge samples' texture levels differently (same estimator).

Communication: ONE all_gather of the partial image after forward and ONE after backward for all gradient
tensors together (a flat bucket; image 12.6 MB at 1024^2, bunny_box gradients < 0.1 MB), each followed by a
sum in FIXED rank order, so the result is bit-identical on every rank and from run to run.  A caller that
has both in hand (bench.py: forward + backward per step) folds them into a single collective
(`_all_gather_sum_many([image] + gradients)`).
`render_blocked` renders the same blocks sequentially on one device with the same summation
order, which is the bit-exact single-GPU counterpart of an R-rank run.  (Against the PLAIN single call --
all samples in one render, added sample by sample -- an R-rank image agrees to ~1e-7 relative L2, the fp32
summation order, not bit for bit.)
"""
import os

import torch
import torch.distributed as dist

from .render_pytorch import RenderFunction


def _collective_forced():
    """REDNER_AMD_FORCE_COLLECTIVE=1: run the gather + fixed-order sum also in a ONE-rank group (the sum of one part is the
    part, bit for bit) -- so that the RCCL path -- communicator set-up, all_gather_into_tensor on device tensors, the unpacking
    -- executes on a box with a single GPU (tests/test_rccl_gpu.py).  Off by default: a one-rank group has nothing to exchange."""
    return os.environ.get('REDNER_AMD_FORCE_COLLECTIVE') == '1'


def _ordered_sum(parts):
    acc = parts[0].clone()
    for p in parts[1:]:
        acc += p
    return acc


def _all_gather_sum(t, group, force=False):
    """Partial results of all ranks summed in FIXED rank order (bit-identical on every rank and from run to run).  One
    collective into one flat [world, n] buffer (all_gather_into_tensor: no list of per-rank tensors, no extra copies)."""
    world = dist.get_world_size(group)
    if world == 1 and not (force or _collective_forced()):
        return t
    flat = t.detach().contiguous().reshape(-1)
    # gloo (CPU tests, rehearsals of the multi-rank path on fewer GPUs than ranks: RDR_BENCH_SHARE_GPU) gathers host tensors
    src = flat.cpu() if flat.is_cuda and dist.get_backend(group) == 'gloo' else flat
    every = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(every, src, group=group)
    every = every.view(world, src.numel()).to(t.device)
    acc = every[0].clone()
    for r in range(1, world):
        acc += every[r]
    return acc.reshape(t.shape)


def _all_gather_sum_many(tensors, group, force=False):
    """The same for a list of tensors in ONE collective: the gradient tensors of a render are many and small (bunny_box:
    ~20 tensors, < 0.1 MB together), and over xGMI a collective this size costs its latency, not its bytes -- so they are
    packed into one flat bucket, gathered once, summed in fixed rank order and unpacked."""
    world = dist.get_world_size(group)
    if (world == 1 and not (force or _collective_forced())) or not tensors:
        return list(tensors)
    out = [None] * len(tensors)
    by_dtype = {}
    for i, t in enumerate(tensors):                 # one bucket per dtype: torch.cat would promote a mixed list
        by_dtype.setdefault(t.dtype, []).append(i)
    for dtype, idx in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):       # the same order on every rank
        red = _all_gather_sum(torch.cat([tensors[i].reshape(-1) for i in idx]), group, force)
        at = 0
        for i in idx:
            n = tensors[i].numel()
            out[i] = red[at:at + n].reshape(tensors[i].shape)
            at += n
    return out


def shard_args(args, rank, world, spp_fwd, spp_bwd):
    """Copy of a serialized argument list restricted to this rank's sample block."""
    meta = dict(args[0])
    assert spp_fwd % world == 0 and spp_bwd % world == 0, 'samples must divide evenly among ranks'
    meta['num_samples'] = (spp_fwd // world, spp_bwd // world)
    meta['sample_offset'] = (rank * (spp_fwd // world), rank * (spp_bwd // world))
    meta['total_samples'] = (spp_fwd, spp_bwd)
    return [meta] + list(args[1:])


class DistributedRenderFunction(torch.autograd.Function):
    """RenderFunction whose forward image and backward gradients are reduced over the process
    group.  Every rank must call it with the same scene and the same upstream gradient."""

    @staticmethod
    def forward(ctx, seed, group, meta, *tensors):
        img = RenderFunction.forward(ctx, seed, meta, *tensors)
        ctx.group = group
        return _all_gather_sum(img, group)

    @staticmethod
    def backward(ctx, grad_img):
        grads = RenderFunction.backward(ctx, grad_img)
        dev = ctx.meta['device']
        live = [g for g in grads[2:] if g is not None]
        reduced = iter(_all_gather_sum_many([g.to(dev) for g in live], ctx.group))      # one bucket, one collective
        out = [None if g is None else next(reduced).to(g.device) for g in grads[2:]]
        return (None, None, None) + tuple(out)


def render_sharded(seed, args, group=None):
    """Render `args` (from RenderFunction.serialize_scene, num_samples = TOTAL samples) with the
    samples split over the ranks of `group`; returns the full image on every rank."""
    group = group if group is not None else dist.group.WORLD
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    spp = args[0]['num_samples']
    local = shard_args(args, rank, world, spp[0], spp[1])
    return DistributedRenderFunction.apply(seed, group, *local)


def render_blocked(seed, args, blocks):
    """Single-process counterpart of a `blocks`-rank run: same sample blocks, same summation
    order (differentiable)."""
    spp = args[0]['num_samples']
    imgs = [RenderFunction.apply(seed, *shard_args(args, b, blocks, spp[0], spp[1])) for b in range(blocks)]
    acc = imgs[0]
    for im in imgs[1:]:
        acc = acc + im
    return acc
