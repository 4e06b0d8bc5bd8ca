"""The C-ABI library loads and exports every symbol include/redner_amd.h declares; the product
refuses to run without a GPU (no CPU fallback).  No compute calls here (-m "not gpu")."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'redner_amd.h')
LIB = os.path.join(ROOT, 'redner_amd', 'lib', 'libredner_amd.so')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rdr_[a-z_0-9]+)\s*\(', src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for s in ('rdr_scene_create', 'rdr_scene_destroy', 'rdr_render', 'rdr_compute_num_channels', 'rdr_last_error'):
        assert s in syms


@pytest.mark.skipif(not os.path.exists(LIB), reason='libredner_amd.so not built (run __graft_entry__.build())')
def test_product_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), s


@pytest.mark.skipif(not os.path.exists(LIB), reason='libredner_amd.so not built')
def test_no_cpu_fallback():
    """Scene(use_gpu=False) must fail loudly; so must a machine without a HIP device."""
    import torch
    from redner_amd import _capi
    previous = _capi.library_path()
    _capi.load(LIB)
    try:
        from redner_amd import redner
        import scenes
        from redner_amd.render_pytorch import RenderFunction
        sc = scenes.single_triangle(torch.device('cpu'), resolution=(8, 8))
        args = RenderFunction.serialize_scene(sc, 1, 1, sampler_type=redner.SamplerType.sobol, device=torch.device('cpu'))
        with pytest.raises(RuntimeError, match='no CPU fallback|HIP device'):
            RenderFunction.apply(1, *args)
    finally:
        if previous:
            _capi.load(previous)        # other tests in this session use the host harness


def test_compute_num_channels(hostsim_backend):
    rd = hostsim_backend
    assert rd.compute_num_channels([rd.channels.radiance], 0) == 3
    assert rd.compute_num_channels([rd.channels.radiance, rd.channels.alpha, rd.channels.uv, rd.channels.generic_texture], 5) == 11


def test_missing_library_is_an_error(tmp_path):
    from redner_amd import _capi
    with pytest.raises(RuntimeError, match='not found'):
        _capi.load(str(tmp_path / 'nope.so'))
