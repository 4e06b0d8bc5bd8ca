"""Load the parity oracle (the reference's C++ core compiled for CPU, oracle/_ref) as a
`redner`-API backend.  Test infrastructure only."""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cached = None


def _oracle_lib():
    return [p for p in glob.glob(os.path.join(ROOT, 'oracle', '_ref', 'redner*.so')) if 'redner_dbg' not in p]


def oracle_available():
    return len(_oracle_lib()) > 0


def load_oracle():
    global _cached
    if _cached is None:
        path = _oracle_lib()[0]
        spec = importlib.util.spec_from_file_location('redner', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cached = mod
    return _cached


def rel_l2(a, b):
    import torch
    a, b = a.double().cpu(), b.double().cpu()
    d = torch.linalg.norm((a - b).flatten())
    n = torch.linalg.norm(b.flatten())
    return float(d / n) if n > 0 else float(d)
