"""Edge list order, per-edge faces and the topology of both edge hierarchies must equal the
reference's exactly (which edge a sample picks depends on them).  Compared against the oracle
build's own structures (oracle/dump_edges.cpp) when it is available."""
import ctypes
import sys

import pytest
import torch

import oracle_util
import scenes
from redner_amd.render_pytorch import RenderFunction


def _dump(backend, builder, path, is_oracle, device=torch.device('cpu')):
    sc = getattr(scenes, builder)(device, resolution=(32, 32))
    args = RenderFunction.serialize_scene(sc, 1, 2, sampler_type=backend.SamplerType.sobol, device=device,
                                          backend=backend)
    u = RenderFunction.unpack_args((1, 2), args[0], args[1:])
    if is_oracle:
        import glob, importlib.util, os
        p = glob.glob(os.path.join(oracle_util.ROOT, 'oracle', '_ref', 'redner_dbg*.so'))[0]
        spec = importlib.util.spec_from_file_location('redner_dbg', p)
        dbg = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(dbg)
        dbg.dump(u.scene, path)
    else:
        from redner_amd import _capi
        lib = _capi.lib()
        lib.rdr_debug_dump_edges.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        assert lib.rdr_debug_dump_edges(u.scene._handle, path.encode()) == 0


def _compare(backend, builder, tmp_path, device):
    ref = oracle_util.load_oracle()
    sys.modules.setdefault('redner', ref)        # pybind type registry for the dump helper
    a, b = str(tmp_path / 'mine.txt'), str(tmp_path / 'ref.txt')
    _dump(backend, builder, a, False, device)
    _dump(ref, builder, b, True)
    la, lb = open(a).read().split('\n'), open(b).read().split('\n')
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        xs, ys = x.split(), y.split()
        assert len(xs) == len(ys)
        if len(xs) >= 7 and not x.startswith(('edges', 'cs', 'ncs', 'expand')):
            assert xs[:5] == ys[:5], (x, y)                      # links + edge id: exact
            for p, q in zip(xs[5:], ys[5:]):                     # weights / bounds: bit-exact
                assert float(p) == float(q), (x, y)
        else:
            assert xs == ys, (x, y)


@pytest.mark.skipif(not oracle_util.oracle_available(), reason='oracle/_ref not built')
@pytest.mark.parametrize('builder', ['two_triangles', 'bunny_box', 'triangle_soup', 'triangle_soup_large'])
def test_edge_structures_match_reference(hostsim_backend, tmp_path, builder):
    """The host builder (edges.cpp: TreeBuilder), through the CPU harness."""
    _compare(hostsim_backend, builder, tmp_path, torch.device('cpu'))


@pytest.mark.gpu
@pytest.mark.skipif(not oracle_util.oracle_available(), reason='oracle/_ref not built')
@pytest.mark.parametrize('builder', ['two_triangles', 'bunny_box', 'living_room_standin', 'single_triangle', 'triangle_soup', 'triangle_soup_large'])
def test_edge_structures_built_on_the_gpu_match_reference(gpu_backend, tmp_path, builder):
    """The kernels of edges_gpu.cpp (codes, sort, radix tree, bounds, treelets): the node arrays are read back from the
    device and must equal the reference's link for link, their weights, costs and bounds bit for bit."""
    if not hasattr(scenes, builder):
        pytest.skip('no such scene builder')
    _compare(gpu_backend, builder, tmp_path, torch.device('cuda:0'))
