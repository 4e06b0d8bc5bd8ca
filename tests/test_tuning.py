"""rdr_tuning (include/redner_amd.h): the fields that replaced the RDR_* environment switches select kernels and schedules
per call, in process.  Every schedule must reproduce the oracle's fixture (image bit for bit on the harness; gradients to
1e-4), and a zeroed struct is the shipped configuration."""
import os

import numpy as np
import pytest
import torch

from golden.make_golden import CASES, render_case
from parity_util import GOLD, assert_parity, compare
from redner_amd import _capi as K

# name -> rdr_tuning fields
SCHEDULES = {
    'default': {},
    'one_sample_per_launch': {'batch_samples': 1},
    'ragged_batches': {'batch_samples': 3},
    'lane_cap': {'batch_lanes': 2000},
    'general_kernels': {'flags': K.TUNE_FORCE_GENERAL},
    'single_stream': {'flags': K.TUNE_NO_OVERLAP},
    'no_hoist': {'flags': K.TUNE_NO_HOIST},
    'pickn_walk': {'flags': K.TUNE_PICKN_WALK},
    'pickh_fused': {'flags': K.TUNE_PICKH_FUSED},
    'pickh_lazy': {'flags': K.TUNE_PICKH_LAZY},
    'pickh_one_launch': {'flags': K.TUNE_PICKH_ONE_LAUNCH},
    'no_nee_compact': {'flags': K.TUNE_NO_NEE_COMPACT},
    'small_frame_forms': {'flags': K.TUNE_PICKH_ONE_LAUNCH | K.TUNE_NO_NEE_COMPACT},
    'large_frame_forms': {'flags': K.TUNE_LARGE_FORMS},
    'pickh_walk_params': {'pickh_slots_per_lane': 4, 'pickh_idle_lanes': 32, 'pickh_steps': 3},
    'pickh_one_launch_lazy': {'flags': K.TUNE_PICKH_ONE_LAUNCH | K.TUNE_PICKH_LAZY},
    'gather_hand_over': {'gather_budget': 2, 'gather_heavy_cap_plus1': 4, 'gather_work_cap_plus1': 6},
    'gather_no_lists': {'gather_budget': 1, 'gather_heavy_cap_plus1': 1, 'gather_work_cap_plus1': 1},
    'little_memory': {'mem_available_mb': 8},
    'unfused_bounce': {'flags': K.TUNE_NO_FUSED_BOUNCE},
    'trace_every_continuation': {'flags': K.TUNE_TRACE_EVERY_CONTINUATION},
}
GPU_ONLY = {
    'refill_everywhere': {'flags': K.TUNE_REFILL_ALL},
    'refill_params': {'flags': K.TUNE_REFILL_ALL, 'refill_rays_per_lane': 2, 'refill_idle_lanes': 8, 'refill_steps': 1},
    'refill_queue_order': {'flags': K.TUNE_REFILL_ALL, 'refill_order': 1},
    'refill_octant_axis_order': {'flags': K.TUNE_REFILL_ALL, 'refill_order': 3},
    'binary_records': {'flags': K.TUNE_TRACE_BINARY | K.TUNE_TRACE_NO_LDS_TOP},
    'wide_records_everywhere': {'wide_max_rays': 1 << 30},
    'one_worker': {'workers': 1},
    'three_workers': {'workers': 3, 'batch_samples': 2},
}
CASE = 'bunny_box_32x32x4'
GLOSSY = 'glossy_floor_blocker_48x48x4'          # secondary edges at every depth (no hoist), NEE-mode gather with real work


def _check(backend, device, case, fields, exact_image):
    b, res, spp, mb = CASES[case][:4]
    out = render_case(backend, b, res, spp, mb, None, {'tuning': fields}, device=device)
    gold = np.load(os.path.join(GOLD, case + '.npz'))
    if exact_image:
        assert np.array_equal(out['image'], gold['image'])
    assert_parity(compare(out, gold), case)


@pytest.mark.parametrize('name', list(SCHEDULES))
def test_schedules_hostsim(hostsim_backend, name):
    _check(hostsim_backend, torch.device('cpu'), CASE, SCHEDULES[name], True)


@pytest.mark.parametrize('name', ['gather_hand_over', 'gather_no_lists', 'pickn_walk', 'ragged_batches'])
def test_schedules_glossy_hostsim(hostsim_backend, name):
    _check(hostsim_backend, torch.device('cpu'), GLOSSY, SCHEDULES[name], True)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(SCHEDULES) + list(GPU_ONLY))
def test_schedules_gpu(gpu_backend, name):
    fields = SCHEDULES.get(name, GPU_ONLY.get(name))
    _check(gpu_backend, torch.device('cuda:0'), CASE, fields, True)
    _check(gpu_backend, torch.device('cuda:0'), GLOSSY, fields, True)


def test_build_flags_and_pool_cap_round_trip(hostsim_backend):
    rd = hostsim_backend
    rd.set_build_flags(K.BUILD_NO_REFIT | K.BUILD_NO_EDGE_CACHE | K.BUILD_SYNC_EDGES)
    try:
        _check(rd, torch.device('cpu'), CASE, {}, True)
        _check(rd, torch.device('cpu'), CASE, {}, True)          # a second Scene with the same connectivity: nothing is reused
    finally:
        rd.set_build_flags(0)
    rd.set_pool_cap_mb(64)
    rd.set_pool_cap_mb(-1)


@pytest.mark.gpu
def test_caller_stream_gpu(gpu_backend):
    """rdr_set_stream: scene tensors produced on a non-default torch stream are read after their producers, and the result is
    visible to that stream without a device-wide synchronisation by the caller."""
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    dev = torch.device('cuda:0')
    gold = np.load(os.path.join(GOLD, 'two_triangles_64x64x16.npz'))
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(side):
        sc = scenes.two_triangles(dev, resolution=(64, 64))
        # a long producer on the side stream: the vertices pass through a slow chain of kernels before the Scene reads them
        v = sc.shapes[0].vertices.detach()
        big = torch.zeros(1 << 26, device=dev)
        for _ in range(20):
            big = big + 1.0
        v = v + (big[:1].sum() - 20.0)          # == v, but ordered after the chain
        sc.shapes[0].vertices = v.clone().requires_grad_(True)
        args = RenderFunction.serialize_scene(sc, 16, 1, sampler_type=gpu_backend.SamplerType.sobol, device=dev)
        img = RenderFunction.apply(1, *args)
        total = img.sum()
    side.synchronize()
    assert np.array_equal(img.detach().cpu().numpy(), gold['image'])
    assert float(total) == float(torch.from_numpy(gold['image']).to(dev).sum())


def test_unknown_tuning_field_is_an_error(hostsim_backend):
    """A misspelt rdr_tuning field must not silently render with the defaults: ctypes.Structure accepts any name."""
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    dev = torch.device('cpu')
    sc = scenes.single_triangle(dev, resolution=(8, 8))
    args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=hostsim_backend.SamplerType.sobol, device=dev,
                                          backend=hostsim_backend, tuning={'batch_sample': 1})
    with pytest.raises(ValueError, match='batch_sample'):
        RenderFunction.apply(1, *args)


@pytest.mark.parametrize('case', [CASE, GLOSSY, 'bunny_box_96x96x8', 'envmap_sphere_48x48x4'])
def test_adjoint_lists_skip_only_zero_terms_hostsim(hostsim_backend, case):
    """The continuation half of the bounce adjoint runs over the lanes that have something to take over (render.cpp: adj_scatter:
    the next depth's list minus the lanes whose successor record is known to be all zeros, AdjState::carries), the next-event half
    over the lanes whose estimate is not zero for geometric reasons, the edge-derivative stages return on a zero contribution.  What
    is skipped only multiplied by zeros and added zeros -- so on the sequential harness with one sample worker every gradient tensor
    must come out BIT FOR BIT the same with the list forms on (the session's default) and off."""
    b, res, spp, mb = CASES[case][:4]
    on = render_case(hostsim_backend, b, res, spp, mb, None, {'tuning': {'workers': 1}}, device=torch.device('cpu'))
    off = render_case(hostsim_backend, b, res, spp, mb, None, {'tuning': {'workers': 1, 'flags': K.TUNE_NO_NEE_COMPACT}}, device=torch.device('cpu'))
    assert set(on) == set(off)
    for k in on:
        assert np.array_equal(np.asarray(on[k]), np.asarray(off[k])), (case, k, float(np.abs(np.asarray(on[k], np.float64) - np.asarray(off[k], np.float64)).max()))


@pytest.mark.parametrize('case', [CASE, GLOSSY, 'bunny_box_96x96x8'])
def test_last_bounce_emitter_test_changes_nothing_hostsim(hostsim_backend, case):
    """The last bounce's continuation rays that meet no emitter triangle are answered "no hit" without a traversal (stages_fwd.h:
    BounceSample::last_bounce_emitters): the vertex they would reach is never shaded, it matters only as an emitter.  Image and
    every gradient tensor bit for bit as with every ray traced (sequential harness, one sample worker)."""
    b, res, spp, mb = CASES[case][:4]
    for form in (0, K.TUNE_NO_FUSED_BOUNCE):          # the fused bounce stage (small frames) and the two-stage form (large ones)
        on = render_case(hostsim_backend, b, res, spp, mb, None, {'tuning': {'workers': 1, 'flags': form}}, device=torch.device('cpu'))
        off = render_case(hostsim_backend, b, res, spp, mb, None, {'tuning': {'workers': 1, 'flags': form | K.TUNE_TRACE_EVERY_CONTINUATION}},
                          device=torch.device('cpu'))
        for k in on:
            assert np.array_equal(np.asarray(on[k]), np.asarray(off[k])), (case, form, k)
