"""TEST INFRASTRUCTURE: a torch-backed stand-in for the part of the TensorFlow 2.x eager API that
`redner_amd/render_tensorflow.py` and `tests/test_tensorflow_frontend.py` call.  This image has no TensorFlow; the stand-in
lets the TensorFlow surface be exercised (serialization, the custom-gradient operator, DLPack hand-over, gradient routing)
on the CPU harness and on the GPU.  It is NOT TensorFlow and is never imported by the product: it is only reachable when a
test puts `tests/tf_standin` on sys.path.

Semantics kept: eager tensors are immutable values with `.shape / .dtype / .numpy()`; `tf.device` scopes place `tf.identity`
results; int32 tensors are held on the host whatever the scope (the TensorFlow behaviour `render_tensorflow.py` works around);
`tf.custom_gradient` / `tf.GradientTape` route gradients through identity, cast, scalar products and reductions -- what the
tests use.
"""
import contextlib

import torch

__version__ = '0-standin'


class DType:
    def __init__(self, name, torch_dtype):
        self.name, self.torch = name, torch_dtype

    def __repr__(self):
        return 'tf.' + self.name

    def __eq__(self, other):
        return isinstance(other, DType) and other.name == self.name

    def __hash__(self):
        return hash(self.name)


float32 = DType('float32', torch.float32)
float64 = DType('float64', torch.float64)
int32 = DType('int32', torch.int32)
int64 = DType('int64', torch.int64)
bool_ = DType('bool', torch.bool)
_BY_TORCH = {d.torch: d for d in (float32, float64, int32, int64, bool_)}

_device_stack = []


def _torch_device(name):
    spec = DeviceSpec.from_string(name)
    if spec.device_type == 'GPU':
        return torch.device('cuda', spec.device_index or 0)
    return torch.device('cpu')


def _current_device():
    return _torch_device(_device_stack[-1]) if _device_stack else None


@contextlib.contextmanager
def device(name):
    _device_stack.append(name)
    try:
        yield
    finally:
        _device_stack.pop()


class DeviceSpec:
    def __init__(self, device_type=None, device_index=None):
        self.device_type, self.device_index = device_type, device_index

    @staticmethod
    def from_string(s):
        parts = [p for p in s.lower().strip('/').split('/') if p]
        for p in parts:
            if p.startswith('device:'):
                p = p[len('device:'):]
            kind, _, idx = p.partition(':')
            if kind in ('gpu', 'cpu'):
                return DeviceSpec(kind.upper(), int(idx) if idx else None)
        return DeviceSpec()


class Tensor:
    """An eager tensor: a value (torch storage underneath) + who produced it, for the tape."""

    def __init__(self, t, inputs=(), vjp=None):
        self._t = t
        self._inputs, self._vjp = tuple(inputs), vjp

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return _BY_TORCH[self._t.dtype]

    @property
    def device(self):
        d = self._t.device
        return '/job:localhost/replica:0/task:0/device:%s:%d' % ('GPU' if d.type == 'cuda' else 'CPU', d.index or 0)

    def numpy(self):
        return self._t.detach().cpu().numpy()

    def __float__(self):
        return float(self._t)

    def __int__(self):
        return int(self._t)

    def __len__(self):
        return self._t.shape[0]

    def __getitem__(self, idx):
        return Tensor(self._t[idx])

    def _bin(self, other, fn, vjp_self=None):
        o = other._t if isinstance(other, Tensor) else other
        if isinstance(o, torch.Tensor) and o.device != self._t.device:
            o = o.to(self._t.device)
        out = fn(self._t, o)
        if vjp_self is not None:
            return Tensor(out, (self,), lambda g, o=o: (vjp_self(g, o),))
        return Tensor(out)

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b, lambda g, b: g * b)          # the other factor is a constant to the tape
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a, lambda g, b: g * b)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a)
    def __neg__(self): return Tensor(-self._t)

    def __repr__(self):
        return 'tf.Tensor(standin, shape=%s, dtype=%s)' % (self.shape, self.dtype.name)


class Variable(Tensor):
    def __init__(self, initial_value, dtype=None, trainable=True):
        super().__init__(convert_to_tensor(initial_value, dtype)._t.clone())
        self.trainable = trainable

    def assign(self, value):
        self._t = convert_to_tensor(value)._t.to(self._t.dtype).clone()
        return self


def is_tensor(x):
    return isinstance(x, Tensor)


def executing_eagerly():
    return True


def _place(t, dtype_is_int=None):
    # int32 stays on the host whatever the scope (TensorFlow's placement rule for int32 kernels)
    dev = _current_device()
    if dev is None or t.dtype == torch.int32:
        return t if t.dtype != torch.int32 else t.cpu()
    if dev.type == 'cuda' and not torch.cuda.is_available():
        raise RuntimeError('standin: no GPU for device scope')
    return t.to(dev)


def convert_to_tensor(x, dtype=None):
    if isinstance(x, Tensor):
        t = x._t
    elif isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(x)
        if t.dtype == torch.float64 and dtype is None:
            t = t.to(torch.float32)              # python floats -> float32, like TensorFlow
        if t.dtype == torch.int64 and dtype is None:
            t = t.to(torch.int32)                # python ints -> int32
    if dtype is not None:
        t = t.to(dtype.torch)
    return x if isinstance(x, Tensor) and t is x._t else Tensor(t)


def constant(x, dtype=None):
    return Tensor(_place(convert_to_tensor(x, dtype)._t.clone()))


def identity(x):
    x = convert_to_tensor(x)
    t = _place(x._t)
    if t is x._t:
        t = t.clone()                            # a new tensor, like TensorFlow's
    return Tensor(t, (x,), lambda g: (g.to(x._t.device),))


def cast(x, dtype):
    x = convert_to_tensor(x)
    t = _place(x._t.to(dtype.torch))
    if x._t.is_floating_point() and t.is_floating_point():
        return Tensor(t, (x,), lambda g: (g.to(x._t.device, x._t.dtype),))
    return Tensor(t)


def reshape(x, shape):
    x = convert_to_tensor(x)
    return Tensor(x._t.reshape(tuple(shape)), (x,), lambda g: (g.reshape(x._t.shape),))


def zeros(shape, dtype=float32):
    return Tensor(_place(torch.zeros(tuple(shape), dtype=dtype.torch)))


def ones(shape, dtype=float32):
    return Tensor(_place(torch.ones(tuple(shape), dtype=dtype.torch)))


def eye(n, m=None, dtype=float32):
    return Tensor(torch.eye(n, m if m is not None else n, dtype=dtype.torch))


def concat(values, axis):
    return Tensor(torch.cat([convert_to_tensor(v)._t for v in values], dim=axis))


def stack(values, axis=0):
    return Tensor(torch.stack([convert_to_tensor(v)._t for v in values], dim=axis))


def tan(x): return Tensor(torch.tan(convert_to_tensor(x)._t))
def sin(x): return Tensor(torch.sin(convert_to_tensor(x)._t))
def cos(x): return Tensor(torch.cos(convert_to_tensor(x)._t))


def cumsum(x, axis=0):
    return Tensor(torch.cumsum(convert_to_tensor(x)._t, dim=axis))


def maximum(a, b):
    a, b = convert_to_tensor(a)._t, convert_to_tensor(b)._t
    return Tensor(torch.maximum(a, b.to(a.device)))


def range(n):                       # noqa: A001 (the TensorFlow name)
    return Tensor(torch.arange(int(n), dtype=torch.int32))


def reduce_sum(x, axis=None):
    x = convert_to_tensor(x)
    if axis is None:
        return Tensor(x._t.sum(), (x,), lambda g: (g.expand(x._t.shape).to(x._t.device),))
    return Tensor(x._t.sum(dim=axis))


def reduce_all(x):
    return Tensor(convert_to_tensor(x)._t.all())


class _Linalg:
    @staticmethod
    def inv(x):
        x = convert_to_tensor(x)

        def vjp(g):
            a = x._t.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                torch.inverse(a).backward(g.to(a.device))
            return (a.grad,)
        return Tensor(torch.inverse(x._t), (x,), vjp)

    @staticmethod
    def diag(x):
        return Tensor(torch.diag(convert_to_tensor(x)._t))


linalg = _Linalg()


class _Math:
    @staticmethod
    def is_finite(x):
        return Tensor(torch.isfinite(convert_to_tensor(x)._t))

    maximum = staticmethod(maximum)


math = _Math()


class _Dlpack:
    @staticmethod
    def to_dlpack(x):
        return torch.utils.dlpack.to_dlpack(x._t)

    @staticmethod
    def from_dlpack(capsule):
        return Tensor(torch.utils.dlpack.from_dlpack(capsule))


class _Experimental:
    dlpack = _Dlpack()


experimental = _Experimental()


def custom_gradient(f):
    def wrapped(*args):
        args = [convert_to_tensor(a) for a in args]
        out, grad_fn = f(*args)

        def vjp(g):
            r = grad_fn(Tensor(g))
            r = list(r) if isinstance(r, (list, tuple)) else [r]
            assert len(r) == len(args), 'custom_gradient: one gradient per input'
            return tuple(None if x is None else x._t for x in r)
        return Tensor(out._t, args, vjp)
    return wrapped


class GradientTape:
    """The stand-in records every op (there is no graph to prune), so the tape is only the entry point of the reverse
    sweep: reverse topological order from `target`, vector-Jacobian products summed per tensor."""

    def __init__(self, persistent=False):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, x):
        pass

    def gradient(self, target, sources):
        single = isinstance(sources, Tensor)
        srcs = [sources] if single else list(sources)
        order, seen = [], set()

        def visit(n):
            if id(n) in seen:
                return
            seen.add(id(n))
            for i in n._inputs:
                visit(i)
            order.append(n)
        visit(target)
        grad = {id(target): torch.ones_like(target._t)}
        for n in reversed(order):
            g = grad.get(id(n))
            if g is None or n._vjp is None:
                continue
            for i, gi in zip(n._inputs, n._vjp(g)):
                if gi is None:
                    continue
                grad[id(i)] = gi if id(i) not in grad else grad[id(i)] + gi
        out = [None if id(s) not in grad else Tensor(grad[id(s)]) for s in srcs]
        return out[0] if single else out
