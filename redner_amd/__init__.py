"""redner_amd -- MI355X-native implementation of redner's differentiable path tracer hot path.

    redner_amd.redner            the `redner` module surface (ctypes over the C ABI)
    redner_amd.render_pytorch    RenderFunction (torch.autograd.Function) + minimal scene classes
    redner_amd.install()         register redner_amd.redner as `redner` for the reference's
                                 unmodified pyredner package
"""
import sys


def install():
    """Make `import redner` resolve to the MI355X implementation (drop-in for pyredner)."""
    from . import redner as _redner
    sys.modules['redner'] = _redner
    return _redner


def trim_cache():
    """Release the per-call device buffers the library keeps parked between render() calls (bytes released)."""
    from . import redner as _redner
    return _redner.trim_cache()
