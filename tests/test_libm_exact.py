"""csrc/libm_exact.h against the C library itself, bit for bit.

The reference's CPU path calls glibc's sin / cos / atan2 / atan / acos / log / pow (its Real is double,
/root/reference/src/redner.h:46; fisheye / panorama rays /root/reference/src/camera.h:142-191, BSDF sampling and Phong lobes
/root/reference/src/material.h, environment maps /root/reference/src/envmap.h, the edge estimators /root/reference/src/edge.cpp).
Wherever such a value feeds the chaotic hierarchical edge pick (/root/reference/src/edge.cpp:1115-1237) one differing ulp
draws another sample, so the kernels carry their own restatement of glibc 2.35's routines (the `_fma` variants its x86-64
build selects on every machine with FMA + AVX2).  Checked here: the CPU harness build of those routines (not gpu) and the
kernels' (gpu) against `oracle/_ref/libm_ref.so` = the C library's functions over arrays, on ~10^7 arguments per function
drawn to cover every branch of every routine (interval boundaries, table ends, tiny / huge / special values).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, 'oracle', '_ref', 'libm_ref.so')
FUNCS = {'sin': 0, 'cos': 1, 'atan2': 2, 'atan': 3, 'acos': 4, 'log': 5, 'pow': 6}
TWO_ARGS = ('atan2', 'pow')


def _glibc():
    if not os.path.exists(REF_SO):          # gcc is on both boxes; normally built by __graft_entry__.build()
        os.makedirs(os.path.dirname(REF_SO), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-fno-builtin', '-shared', '-fPIC', os.path.join(ROOT, 'oracle', 'libm_ref.c'),
                               '-o', REF_SO, '-lm'])
    lib = C.CDLL(REF_SO)
    lib.libm_ref_eval.restype = None
    lib.libm_ref_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    return lib


def _specials():
    s = [0.0, 1.0, 2.0, 3.0, 0.5, 0.125, 0.126, 0.0625, 16.0, 0.75, 0.855469, 2.426265, 0.921875, 0.953125, 0.96875, 0.984375,
         1.5707963267948966, 3.141592653589793, 6.283185307179586, 2.0 ** -26, 2.0 ** -27, 2.0 ** -500, 2.0 ** 500, 1e-320,
         5e-324, 2.2250738585072014e-308, 1e308, 1.7976931348623157e308, 105414349.0, float.fromhex('0x1.bb67ap-27'), float.fromhex('0x1.49ff2p+52'),
         1.0 - 2.0 ** -4, 1.0 + float.fromhex('0x1.09p-4'), 2.0 ** -65, 2.0 ** 63, 1074.0, 1075.0, 1024.0, 2.0 ** 53, 2.0 ** 53 + 2, 2.0 ** 52 + 1,
         np.inf, np.nan]
    out = []
    for v in s:
        for w in (v, np.nextafter(v, np.inf), np.nextafter(v, -np.inf)):
            out += [w, -w]
    return np.array(out, dtype=np.float64)


def _log_uniform(rng, n, lo, hi):
    return np.ldexp(1.0 + rng.random(n), rng.integers(lo, hi, n).astype(np.int32)) * rng.choice([-1.0, 1.0], n)


def arguments(name, n, seed=20240925):
    """(x, y) covering the routine's branches; y is None for one-argument functions."""
    rng = np.random.default_rng(seed + FUNCS[name])
    sp = _specials()
    if name in ('sin', 'cos'):
        parts = [(rng.random(n) * 2 - 1) * 7.0, rng.random(n) * 2 - 1, (rng.random(n) * 2 - 1) * 1000.0,
                 (rng.random(n) * 2 - 1) * 1.05e8, _log_uniform(rng, n, -40, 20),
                 _log_uniform(rng, n // 2, 26, 1024), (rng.random(n // 2) * 2 - 1) * 1e9,     # branred.c: |x| >= 105414350
                 np.arange(0, 110 * 64) / (128.0 * 64),                   # every table interval of do_sin / do_cos
                 (rng.integers(-200, 200, n) + (rng.random(n) - 0.5) * 1e-9) * (np.pi / 2), sp]
        return np.concatenate(parts), None
    if name == 'atan':
        parts = [rng.random(n) * 2 - 1, (rng.random(n) * 2 - 1) * 20, _log_uniform(rng, n, -60, 60), _log_uniform(rng, n, -1070, 1020),
                 (16 + np.arange(0, 241 * 16) / 16.0) / 256.0, sp]
        return np.concatenate(parts), None
    if name == 'acos':
        parts = [rng.random(n) * 2 - 1, 1 - np.ldexp(rng.random(n), -rng.integers(0, 54, n).astype(np.int32)),
                 -1 + np.ldexp(rng.random(n), -rng.integers(0, 54, n).astype(np.int32)), _log_uniform(rng, n, -70, 1),
                 np.arange(-4096, 4097) / 4096.0, sp]
        return np.concatenate(parts), None
    if name == 'log':
        parts = [rng.random(n) * 4, 1 + (rng.random(n) - 0.5) * 0.14, np.abs(_log_uniform(rng, n, -1074, 1023)),
                 np.abs(_log_uniform(rng, n, -20, 20)), np.ldexp(rng.random(n), -1040), -rng.random(16), sp]
        return np.concatenate(parts), None
    if name == 'atan2':
        a = rng.random(n) * 2 * np.pi
        x0 = rng.random(n) + 0.1
        ys = [rng.random(n) * 2 - 1, np.sin(a) * rng.random(n), _log_uniform(rng, n, -30, 30), _log_uniform(rng, n, -1070, 1020),
              x0 * (0.0625 + (rng.random(n) - 0.5) * 1e-3), x0 * (1 + (rng.random(n) - 0.5) * 1e-6)]
        xs = [rng.random(n) * 2 - 1, np.cos(a) * rng.random(n), _log_uniform(rng, n, -30, 30), _log_uniform(rng, n, -1070, 1020),
              x0, x0 * rng.choice([-1.0, 1.0], n)]
        gy, gx = np.meshgrid(sp, sp)
        return np.concatenate(ys + [gy.ravel()]), np.concatenate(xs + [gx.ravel()])
    if name == 'pow':
        xs = [rng.random(n), rng.random(n), rng.random(n) * 10, 1 + (rng.random(n) - 0.5) * 1e-3, np.abs(_log_uniform(rng, n, -1070, 1020)),
              -rng.random(n) * 10, rng.random(n), rng.random(n), np.abs(_log_uniform(rng, n, -1070, 1020)), np.ldexp(rng.random(n), -1040)]
        ys = [rng.random(n) * 100, rng.random(n) * 5, (rng.random(n) * 2 - 1) * 10, (rng.random(n) * 2 - 1) * 1e5,
              (rng.random(n) * 2 - 1) * np.ldexp(1.0, rng.integers(-70, 10, n).astype(np.int32)), np.floor(rng.random(n) * 20 - 10),
              np.full(n, 5.0), np.full(n, 4.0), (rng.random(n) * 2 - 1) * 3, rng.random(n) * 2 - 1]
        gx, gy = np.meshgrid(sp, sp)
        return np.concatenate(xs + [gx.ravel()]), np.concatenate(ys + [gy.ravel()])
    raise KeyError(name)


def glibc_values(name, x, y):
    out = np.empty_like(x)
    _glibc().libm_ref_eval(FUNCS[name], x.ctypes.data, y.ctypes.data if y is not None else None, out.ctypes.data, x.size)
    return out


def library_values(name, x, y):
    from redner_amd import _capi
    out = np.empty_like(x)
    step = 1 << 22
    for a in range(0, x.size, step):
        xs = np.ascontiguousarray(x[a:a + step])
        ys = np.ascontiguousarray(y[a:a + step]) if y is not None else None
        o = np.empty_like(xs)
        rc = _capi.lib().rdr_debug_libm(FUNCS[name], xs.ctypes.data, ys.ctypes.data if ys is not None else None, o.ctypes.data, xs.size)
        assert rc == 0, _capi.last_error()
        out[a:a + step] = o
    return out


def assert_bit_equal(name, x, y, want, got):
    wb, gb = want.view(np.uint64), got.view(np.uint64)
    differ = (wb != gb) & ~(np.isnan(want) & np.isnan(got))       # a NaN is a NaN (payloads / signs of invalid results)
    if differ.any():
        i = np.flatnonzero(differ)[:5]
        raise AssertionError('%s: %d of %d arguments differ from glibc, e.g. %s' % (
            name, int(differ.sum()), x.size,
            [(float(x[k]).hex(), None if y is None else float(y[k]).hex(), float(want[k]).hex(), float(got[k]).hex()) for k in i]))


@pytest.mark.parametrize('name', sorted(FUNCS))
def test_libm_exact_cpu_harness(name, hostsim_backend):
    """The routines as g++ compiles them for the CPU harness (tests/hostsim): bit-equal to glibc."""
    x, y = arguments(name, 400000)
    assert_bit_equal(name, x, y, glibc_values(name, x, y), library_values(name, x, y))


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(FUNCS))
def test_libm_exact_gpu(name, gpu_backend):
    """The routines as the kernels run them on gfx950 (one argument per lane through rdr_debug_libm): bit-equal to the glibc
    of the box's host -- the library the oracle calls there."""
    from redner_amd import _capi
    if not _capi.lib().rdr_libm_exact():
        pytest.skip("the default build computes these with the device's own libm (ocml): nothing to hold to glibc")
    x, y = arguments(name, 1500000)
    assert_bit_equal(name, x, y, glibc_values(name, x, y), library_values(name, x, y))


def test_tables_are_the_c_librarys():
    """Every coefficient table of the header, entry by entry, inside this machine's libm.so.6; 2 / pi recomputed from scratch."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'extract_libm_tables.py')], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
