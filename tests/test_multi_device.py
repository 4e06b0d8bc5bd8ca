"""One process driving TWO devices from two host threads: the API is serialised per device (csrc/capi.cpp), every
cache is per device (edge / topology / gather / merged caches, buffer pool lists), so two renders on two devices may run side by
side -- and must return what they return one after the other.  Needs two GPUs: skipped on the one-GPU test box, run where the
driver has a multi-GPU node.  (The supported multi-GPU deployment is one PROCESS per GPU, redner_amd/distributed.py.)"""
import threading

import numpy as np
import pytest
import torch


def _render(redner, dev, res, spp, seed, out, key, rounds=3):
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    try:
        for r in range(rounds):            # new Scene per round: the per-device caches are exercised, alternating with the other thread
            sc = scenes.bunny_box(dev, resolution=(res, res))
            args = RenderFunction.serialize_scene(sc, spp, 4, sampler_type=redner.SamplerType.sobol, device=dev)
            img = RenderFunction.apply(seed, *args)
            img.sum().backward()
            torch.cuda.synchronize(dev)
            g = [s.vertices.grad.cpu().numpy() for s in sc.shapes if s.vertices.grad is not None]
            out[(key, r)] = (img.detach().cpu().numpy(), g)
    except Exception as e:          # surfaced by the caller
        out[(key, 'error')] = e


@pytest.mark.gpu
def test_two_devices_from_two_threads_equal_serial(gpu_backend):
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (the round-end multi-GPU tier)')
    redner = gpu_backend
    devs = [torch.device('cuda:0'), torch.device('cuda:1')]
    serial, threaded = {}, {}
    for k, dev in enumerate(devs):
        _render(redner, dev, 96, 8, 5 + k, serial, k)
    ts = [threading.Thread(target=_render, args=(redner, dev, 96, 8, 5 + k, threaded, k)) for k, dev in enumerate(devs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k in range(2):
        assert (k, 'error') not in serial and (k, 'error') not in threaded, (serial.get((k, 'error')), threaded.get((k, 'error')))
        for r in range(3):
            a_img, a_g = serial[(k, r)]
            b_img, b_g = threaded[(k, r)]
            assert np.array_equal(a_img, b_img), (k, r)
            for x, y in zip(a_g, b_g):
                n = np.linalg.norm(x.astype(np.float64))
                assert np.linalg.norm(x.astype(np.float64) - y) <= 1e-6 * max(n, 1e-30), (k, r)
    # the same frame on the two devices: same image bit for bit
    assert np.array_equal(serial[(0, 0)][0].shape, serial[(1, 0)][0].shape)
