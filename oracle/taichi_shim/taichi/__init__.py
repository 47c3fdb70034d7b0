"""Build-authored stand-in for the `taichi` package -- FIXTURE GENERATION ONLY (this container).

It lets the reference's *unmodified* SPH/*.py files be imported from /root/reference and executed
serially on tiny scenes, so that golden vectors can be written to tests/golden/ (see
oracle/gen_golden.py).  It is NOT Taichi: results are labelled "reference source under a serial
f32 interpreter", never "Taichi ti.cpu".  Nothing here travels to the GPU box's code path and
nothing in the product imports it.

Semantics implemented (exactly the constructs the reference's hot path uses):
  * fields (0-d / 1-d scalar, vector, matrix) over numpy f32 / i32 storage; f32 reads give
    np.float32 (all arithmetic stays f32 under NumPy-2 weak-scalar promotion), i32 reads give
    python ints (so f32 / i32 stays f32 as in Taichi);
  * ti.Vector / ti.Matrix / ti.Struct value types with Taichi's by-value field reads
    (`pos = field[i]` copies; `field[i][k] = v` writes through);
  * @ti.kernel / @ti.func executed as plain serial Python after three AST rewrites applied on
    first call: (1) `ti.atomic_add/sub(field[idx], v)` on an lvalue, (2) by-reference
    `ti.template()` accumulators of `for_all_neighbors(p_i, task, ret)` (boxed around the call),
    (3) kernel-local float assignments are rounded to f32 (Taichi locals are typed f32);
  * ti.algorithms.PrefixSumExecutor as an inclusive in-place scan (its use at
    base_container.py:513-515/:555-557 fixes that semantics).
"""
from __future__ import annotations

import ast
import inspect
import itertools
import textwrap
import types as _pytypes

import numpy as np

f32 = np.float32
i32 = np.int32
f64 = np.float64
gpu = "gpu"
cpu = "cpu"


def init(*args, **kwargs):
    return None


def _np_dtype(dt):
    if dt in (float, f32, "f32"):
        return np.float32
    if dt in (int, i32, "i32"):
        return np.int32
    return np.dtype(dt).type


def _unbox(v):
    return v.v if isinstance(v, _Box) else v


def _scalar(v):
    v = _unbox(v)
    if isinstance(v, (float, np.floating)):
        return np.float32(v)
    return v


def _loc(v):
    """kernel-local assignment: python / numpy floats become f32 (Taichi default_fp)."""
    if type(v) is float or (isinstance(v, np.floating) and not isinstance(v, np.float32)):
        return np.float32(v)
    if isinstance(v, Vec) and v._f is not None:
        # `pos = field[i]`: Taichi copies the element into the local; element writes on the local must not reach the
        # field (only `field[i][k] = v`, which is not an assignment to a name, writes through)
        return Vec(v.d, v.d.dtype.type)
    return v


# ------------------------------------------------------------------------------- value types
class Vec:
    __slots__ = ("d", "_f", "_i")
    __array_ufunc__ = None  # numpy scalars must defer to our reflected operators

    def __init__(self, data, dtype=None, _field=None, _idx=None):
        if isinstance(data, Vec):
            data = data.d
        arr = np.array([_unbox(x) for x in data] if not isinstance(data, np.ndarray) else data)
        if dtype is None:
            dtype = np.float32 if arr.dtype.kind == "f" else np.int32
        self.d = arr.astype(dtype, copy=True)
        self._f, self._i = _field, _idx

    # --- helpers
    @property
    def n(self):
        return self.d.shape[0]

    def _coerce(self, o):
        o = _unbox(o)
        if isinstance(o, Vec):
            return o.d
        if isinstance(o, (float, np.floating)):
            return np.float32(o)
        return o

    def _res(self, arr):
        if arr.dtype == np.float64:
            arr = arr.astype(np.float32)
        return Vec(arr, arr.dtype.type)

    def __len__(self):
        return self.n

    def __iter__(self):
        return (self[i] for i in range(self.n))

    def __getitem__(self, k):
        v = self.d[int(_unbox(k))]
        return int(v) if self.d.dtype.kind == "i" else np.float32(v)

    def __setitem__(self, k, val):
        k = int(_unbox(k))
        self.d[k] = _unbox(val)
        if self._f is not None:  # field[i][k] = val
            self._f._data[self._i][k] = self.d[k]

    # --- arithmetic (elementwise, f32)
    def __add__(self, o): return self._res(self.d + self._coerce(o))
    def __radd__(self, o): return self._res(self._coerce(o) + self.d)
    def __sub__(self, o): return self._res(self.d - self._coerce(o))
    def __rsub__(self, o): return self._res(self._coerce(o) - self.d)
    def __mul__(self, o): return self._res(self.d * self._coerce(o))
    def __rmul__(self, o): return self._res(self._coerce(o) * self.d)
    def __truediv__(self, o): return self._res(self.d / self._coerce(o))
    def __neg__(self): return self._res(-self.d)

    def _inplace(self, arr):
        if arr.dtype == np.float64:
            arr = arr.astype(np.float32)
        self.d = arr.astype(self.d.dtype) if arr.dtype != self.d.dtype and self.d.dtype.kind == "f" else arr
        return self

    def __iadd__(self, o): return self._inplace(self.d + self._coerce(o))
    def __isub__(self, o): return self._inplace(self.d - self._coerce(o))
    def __imul__(self, o): return self._inplace(self.d * self._coerce(o))
    def __itruediv__(self, o): return self._inplace(self.d / self._coerce(o))

    # --- Taichi vector API
    def norm_sqr(self):
        s = self.d[0] * self.d[0]
        for k in range(1, self.n):
            s = s + self.d[k] * self.d[k]
        return np.float32(s)

    def norm(self):
        return np.float32(np.sqrt(self.norm_sqr()))

    def dot(self, o):
        o = self._coerce(o)
        s = self.d[0] * o[0]
        for k in range(1, self.n):
            s = s + self.d[k] * o[k]
        return np.float32(s) if self.d.dtype.kind == "f" else int(s)

    def cross(self, o):
        a, b = self.d, self._coerce(o)
        return Vec(np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]],
                            dtype=np.float32))

    def outer_product(self, o):
        b = self._coerce(o)
        return Mat(np.array([[self.d[a] * b[c] for c in range(len(b))] for a in range(self.n)], dtype=np.float32))

    def cast(self, dt):
        if _np_dtype(dt) == np.int32:
            return Vec(np.trunc(self.d).astype(np.int32), np.int32)
        return Vec(self.d.astype(np.float32), np.float32)

    def to_numpy(self):
        return self.d.copy()

    def __repr__(self):
        return f"Vec({self.d})"


class Mat:
    __slots__ = ("d",)
    __array_ufunc__ = None

    def __init__(self, data):
        self.d = np.array(data.d if isinstance(data, Mat) else data, dtype=np.float32)

    def _coerce(self, o):
        o = _unbox(o)
        if isinstance(o, Mat):
            return o.d
        if isinstance(o, (float, np.floating)):
            return np.float32(o)
        return o

    def __getitem__(self, k):
        return np.float32(self.d[k])

    def __setitem__(self, k, v):
        self.d[k] = _unbox(v)

    def __add__(self, o): return Mat(self.d + self._coerce(o))
    def __sub__(self, o): return Mat(self.d - self._coerce(o))
    def __mul__(self, o): return Mat(self.d * self._coerce(o))
    def __rmul__(self, o): return Mat(self._coerce(o) * self.d)
    def __truediv__(self, o): return Mat(self.d / self._coerce(o))
    def __neg__(self): return Mat(-self.d)

    def __iadd__(self, o): self.d = (self.d + self._coerce(o)).astype(np.float32); return self
    def __isub__(self, o): self.d = (self.d - self._coerce(o)).astype(np.float32); return self

    def __matmul__(self, o):
        o = _unbox(o)
        n = self.d.shape[1]
        if isinstance(o, Vec):
            out = []
            for i in range(self.d.shape[0]):
                s = self.d[i, 0] * o.d[0]
                for k in range(1, n):
                    s = s + self.d[i, k] * o.d[k]
                out.append(s)
            return Vec(np.array(out, dtype=np.float32))
        b = o.d
        out = np.zeros((self.d.shape[0], b.shape[1]), np.float32)
        for i in range(out.shape[0]):
            for j in range(out.shape[1]):
                s = self.d[i, 0] * b[0, j]
                for k in range(1, n):
                    s = s + self.d[i, k] * b[k, j]
                out[i, j] = s
        return Mat(out)

    def determinant(self):
        a = self.d
        return np.float32(a[0, 0] * (a[1, 1] * a[2, 2] - a[2, 1] * a[1, 2]) - a[1, 0] * (a[0, 1] * a[2, 2] - a[2, 1] * a[0, 2])
                          + a[2, 0] * (a[0, 1] * a[1, 2] - a[1, 1] * a[0, 2]))

    def inverse(self):
        # cofactor form of Taichi's matrix_ops.inverse for n = 3 (restated from memory of Taichi >= 1.6)
        a = self.d
        n = 3
        inv_det = np.float32(1.0) / self.determinant()
        E = lambda x, y: a[x % n, y % n]
        out = np.zeros((3, 3), np.float32)
        for i in range(n):
            for j in range(n):
                out[j, i] = inv_det * (E(i + 1, j + 1) * E(i + 2, j + 2) - E(i + 2, j + 1) * E(i + 1, j + 2))
        return Mat(out)

    def to_numpy(self):
        return self.d.copy()


class _StructVal:
    def __init__(self, **kw):
        for k, v in kw.items():
            object.__setattr__(self, k, _loc(v))

    def __setattr__(self, k, v):
        v = _unbox(v)
        old = self.__dict__.get(k)
        if isinstance(old, (np.floating, float)) and not isinstance(v, Vec):
            v = np.float32(v)
        object.__setattr__(self, k, v)


def Struct(**kw):
    return _StructVal(**kw)


class _Box:
    """by-reference `ti.template()` argument around for_all_neighbors (see module docstring)."""
    __slots__ = ("v",)
    __array_ufunc__ = None

    def __init__(self, v):
        self.v = v.v if isinstance(v, _Box) else v

    def get(self):
        return self.v

    def _mut(self):
        return isinstance(self.v, (Vec, Mat, _StructVal))

    def __iadd__(self, o):
        if self._mut(): self.v += o
        else: self.v = _loc(_scalar(self.v) + _unbox(o))
        return self

    def __isub__(self, o):
        if self._mut(): self.v -= o
        else: self.v = _loc(_scalar(self.v) - _unbox(o))
        return self

    def __getattr__(self, k):
        return getattr(object.__getattribute__(self, "v"), k)

    def __setattr__(self, k, val):
        if k == "v":
            object.__setattr__(self, k, val)
        else:
            setattr(self.v, k, val)

    def __getitem__(self, k): return self.v[k]
    def __setitem__(self, k, val): self.v[k] = val
    def __add__(self, o): return self.v + _unbox(o)
    def __radd__(self, o): return _unbox(o) + self.v
    def __sub__(self, o): return self.v - _unbox(o)
    def __rsub__(self, o): return _unbox(o) - self.v
    def __mul__(self, o): return self.v * _unbox(o)
    def __rmul__(self, o): return _unbox(o) * self.v
    def __truediv__(self, o): return self.v / _unbox(o)
    def __rtruediv__(self, o): return _unbox(o) / self.v
    def __neg__(self): return -self.v
    def __float__(self): return float(self.v)
    def __lt__(self, o): return self.v < _unbox(o)
    def __le__(self, o): return self.v <= _unbox(o)
    def __gt__(self, o): return self.v > _unbox(o)
    def __ge__(self, o): return self.v >= _unbox(o)


# ------------------------------------------------------------------------------- fields
class _ScalarField:
    def __init__(self, dtype, shape):
        self.dtype = _np_dtype(dtype)
        self.shape = () if shape == () else ((shape,) if isinstance(shape, (int, np.integer)) else tuple(shape))
        self._data = np.zeros(self.shape, dtype=self.dtype)

    def _get(self, v):
        return int(v) if self.dtype == np.int32 else np.float32(v)

    def __getitem__(self, k):
        if k is None:
            return self._get(self._data[()])
        k = _unbox(k)
        if isinstance(k, Vec):
            k = tuple(int(x) for x in k.d)
        elif k < 0:
            raise IndexError(f"taichi shim: negative field index {k} (the reference has no bounds check here)")
        return self._get(self._data[k])

    def __setitem__(self, k, v):
        v = _unbox(v)
        if k is None:
            self._data[()] = v
        else:
            k = _unbox(k)
            if isinstance(k, Vec):
                k = tuple(int(x) for x in k.d)
            self._data[k] = v

    def fill(self, v):
        self._data[...] = v

    def to_numpy(self):
        return self._data.copy()

    def from_numpy(self, a):
        self._data[...] = a


class _VectorField:
    def __init__(self, n, dtype, shape):
        self.n = n
        self.dtype = _np_dtype(dtype)
        self.shape = ((shape,) if isinstance(shape, (int, np.integer)) else tuple(shape))
        self._data = np.zeros(self.shape + (n,), dtype=self.dtype)

    def __getitem__(self, k):
        k = int(_unbox(k))
        return Vec(self._data[k], self.dtype, _field=self, _idx=k)

    def __setitem__(self, k, v):
        k = int(_unbox(k))
        v = _unbox(v)
        self._data[k] = v.d if isinstance(v, Vec) else np.asarray(v)

    def fill(self, v):
        self._data[...] = v

    def to_numpy(self):
        return self._data.copy()

    def from_numpy(self, a):
        self._data[...] = a


class _MatrixField:
    def __init__(self, n, m, dtype, shape):
        self.shape = ((shape,) if isinstance(shape, (int, np.integer)) else tuple(shape))
        self._data = np.zeros(self.shape + (n, m), dtype=np.float32)

    def __getitem__(self, k):
        return Mat(self._data[int(_unbox(k))])

    def __setitem__(self, k, v):
        v = _unbox(v)
        self._data[int(_unbox(k))] = v.d if isinstance(v, Mat) else np.asarray(v)

    def fill(self, v):
        self._data[...] = v

    def to_numpy(self):
        return self._data.copy()


def field(dtype, shape=()):
    return _ScalarField(dtype, shape)


class _VectorNS:
    def __call__(self, data, dt=None):
        return Vec(list(data) if not isinstance(data, (Vec, np.ndarray)) else data,
                   None if dt is None else _np_dtype(dt))

    @staticmethod
    def field(n, dtype, shape):
        return _VectorField(n, dtype, shape)

    @staticmethod
    def zero(dt, n):
        return Vec(np.zeros(n), _np_dtype(dt))


class _MatrixNS:
    def __call__(self, data, dt=None):
        return Mat(data)

    @staticmethod
    def field(n, m, dtype, shape):
        return _MatrixField(n, m, dtype, shape)

    @staticmethod
    def zero(dt, n, m):
        return Mat(np.zeros((n, m), np.float32))

    @staticmethod
    def identity(dt, n):
        return Mat(np.eye(n, dtype=np.float32))


Vector = _VectorNS()
Matrix = _MatrixNS()


# ------------------------------------------------------------------------------- functions
def template():
    return "template"


class _Types:
    @staticmethod
    def ndarray(*a, **k):
        return "ndarray"

    @staticmethod
    def vector(*a, **k):
        return "vector"


types = _Types()


def static(x):
    return x


def ndrange(*ranges):
    rs = [range(r[0], r[1]) if isinstance(r, tuple) else range(r) for r in ranges]
    return itertools.product(*rs)


def grouped(x):
    if isinstance(x, (_ScalarField, _VectorField)):
        return (Vec(np.array(idx), np.int32) if len(idx) > 1 else idx[0] for idx in np.ndindex(*x.shape))
    return (Vec(np.array(t), np.int32) for t in x)


def _atomic_add_at(fld, idx, v):
    old = fld[idx]
    fld[idx] = old + _unbox(v)
    return old


def _atomic_sub_at(fld, idx, v):
    old = fld[idx]
    fld[idx] = old - _unbox(v)
    return old


def atomic_add(*a):  # only reachable if the AST rewrite missed a call site
    raise RuntimeError("taichi shim: atomic_add on a non-lvalue")


atomic_sub = atomic_add


def cast(v, dt):
    v = _unbox(v)
    return np.float32(v) if _np_dtype(dt) == np.float32 else int(v)


def pow(a, b):  # noqa: A001
    return np.float32(np.power(_scalar(a), _scalar(b)))


def max(a, b):  # noqa: A001
    a, b = _scalar(a), _scalar(b)
    return a if a >= b else (np.float32(b) if isinstance(a, np.floating) else b)


def min(a, b):  # noqa: A001
    a, b = _scalar(a), _scalar(b)
    return a if a <= b else (np.float32(b) if isinstance(a, np.floating) else b)


def abs(a):  # noqa: A001
    return np.abs(_scalar(a))


def sqrt(a):
    return np.float32(np.sqrt(_scalar(a)))


class _Math:
    @staticmethod
    def cross(a, b):
        return _unbox(a).cross(b)

    @staticmethod
    def dot(a, b):
        return _unbox(a).dot(b)

    @staticmethod
    def inverse(m):
        return _unbox(m).inverse()


math = _Math()


class _PrefixSumExecutor:
    def __init__(self, n):
        self.n = n

    def run(self, fld):
        fld._data[...] = np.cumsum(fld._data, dtype=np.int64).astype(np.int32)


class _Algorithms:
    PrefixSumExecutor = _PrefixSumExecutor


algorithms = _Algorithms()


def data_oriented(cls):
    return cls


# ------------------------------------------------------------------------------- AST rewriting
class _Rewriter(ast.NodeTransformer):
    def __init__(self):
        self.counter = 0

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if (isinstance(f, ast.Attribute) and f.attr in ("atomic_add", "atomic_sub") and isinstance(f.value, ast.Name)
                and f.value.id == "ti" and len(node.args) == 2 and isinstance(node.args[0], ast.Subscript)):
            sub = node.args[0]
            new = ast.Call(func=ast.Attribute(value=ast.Name(id="ti", ctx=ast.Load()), attr="_" + f.attr + "_at", ctx=ast.Load()),
                           args=[sub.value, sub.slice, node.args[1]], keywords=[])
            return ast.copy_location(new, node)
        return node

    def visit_Expr(self, node):
        self.generic_visit(node)
        c = node.value
        if (isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute) and c.func.attr == "for_all_neighbors"
                and len(c.args) == 3 and isinstance(c.args[2], ast.Name)):
            name = c.args[2].id
            box = f"__box{self.counter}"
            self.counter += 1
            pre = ast.parse(f"{box} = ti._Box({name})").body[0]
            c.args[2] = ast.Name(id=box, ctx=ast.Load())
            post = ast.parse(f"{name} = {box}.get()").body[0]
            return [ast.copy_location(pre, node), node, ast.copy_location(post, node)]
        return node

    def visit_Assign(self, node):
        self.generic_visit(node)
        if len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            node.value = ast.Call(func=ast.Attribute(value=ast.Name(id="ti", ctx=ast.Load()), attr="_loc", ctx=ast.Load()),
                                  args=[node.value], keywords=[])
        return node


def _compile(fn):
    src = textwrap.dedent(inspect.getsource(fn))
    tree = ast.parse(src)
    fdef = tree.body[0]
    fdef.decorator_list = []
    tree = _Rewriter().visit(tree)
    ast.fix_missing_locations(tree)
    glb = fn.__globals__
    ns = {}
    exec(compile(tree, filename=f"<taichi-shim:{fn.__qualname__}>", mode="exec"), glb, ns)
    return ns[fdef.name]


def _lazy(fn):
    cache = {}

    def wrapper(*args, **kwargs):
        g = cache.get("f")
        if g is None:
            g = cache["f"] = _compile(fn)
        return g(*args, **kwargs)

    wrapper.__name__ = fn.__name__
    wrapper.__qualname__ = fn.__qualname__
    wrapper.__wrapped__ = fn
    return wrapper


def kernel(fn):
    return _lazy(fn)


def func(fn):
    return _lazy(fn)


class _UI:
    def __getattr__(self, k):
        raise RuntimeError("taichi shim has no GGUI")


ui = _UI()
tools = _UI()
