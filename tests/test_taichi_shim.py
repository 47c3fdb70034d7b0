"""CPU: known-answer tests of oracle/taichi_shim itself -- the build-authored serial f32 interpreter under which
oracle/gen_golden.py executes the reference's unmodified source to produce tests/golden/*.npz.  Every Taichi semantic the
fixtures (and through them the oracle's pin) rely on is checked here against what Taichi documents: by-value reads of
field elements, write-through of `field[i][k] = v`, by-reference `ti.template()` accumulators across
for_all_neighbors, truncation toward zero in cast(int), inclusive in-place prefix sum, atomic_add/sub returning the old
value, f32-typed locals and f32/i32 arithmetic, the 3x3 inverse.  (The shim is still not Taichi: see DESIGN.md 2.)"""
import importlib.util
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "taichi_shim", "taichi", "__init__.py")
_spec = importlib.util.spec_from_file_location("taichi_shim_under_test", _PATH)
ti = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ti)


_vf = ti.Vector.field(3, ti.f32, shape=4)


@ti.kernel
def _by_value():
    p = _vf[1]            # Taichi: a copy of the element
    p[0] = 50.0           # ... so this stays in the local
    p += ti.Vector([1.0, 1.0, 1.0])
    _vf[1][2] = 9.0       # writes the field
    _vf[2] = p            # whole-element store


def test_field_reads_copy_and_element_writes_go_through():
    _vf[1] = ti.Vector([1.0, 2.0, 3.0])
    _by_value()
    out = _vf.to_numpy()
    assert out[1].tolist() == [1.0, 2.0, 9.0], "a local copy of a field element must not alias the field"
    assert out[2].tolist() == [51.0, 3.0, 4.0]
    assert out.dtype == np.float32


@ti.data_oriented
class _Acc:
    def __init__(self):
        self.out = ti.field(ti.f32, shape=3)
        self.vout = ti.Vector.field(3, ti.f32, shape=3)
        self.cnt = ti.field(ti.i32, shape=4)
        self.seen = ti.field(ti.i32, shape=8)

    @ti.func
    def for_all_neighbors(self, p_i, task: ti.template(), ret: ti.template()):
        for j in range(4):
            if j != p_i:
                task(p_i, j, ret)

    @ti.func
    def task_scalar(self, p_i, p_j, ret: ti.template()):
        ret += 0.5 * p_j

    @ti.func
    def task_vec(self, p_i, p_j, ret: ti.template()):
        ret += ti.Vector([1.0, 0.0, float(p_j)])

    @ti.kernel
    def run(self):
        for i in range(3):
            s = 0.0
            self.for_all_neighbors(i, self.task_scalar, s)
            self.out[i] = s
            v = ti.Vector([0.0, 0.0, 0.0])
            self.for_all_neighbors(i, self.task_vec, v)
            self.vout[i] = v

    @ti.kernel
    def atomics(self):
        for i in range(6):
            old = ti.atomic_add(self.cnt[i % 2], 1)
            self.seen[i] = old
        a = ti.atomic_sub(self.cnt[0], 2)
        self.seen[6] = a
        self.seen[7] = self.cnt[0]


def test_template_accumulators_are_by_reference():
    a = _Acc()
    a.run()
    # neighbours of i are {0,1,2,3} minus i
    assert a.out.to_numpy().tolist() == [0.5 * (1 + 2 + 3), 0.5 * (0 + 2 + 3), 0.5 * (0 + 1 + 3)]
    assert a.vout.to_numpy().tolist() == [[3.0, 0.0, 6.0], [3.0, 0.0, 5.0], [3.0, 0.0, 4.0]]
    assert a.out.to_numpy().dtype == np.float32


def test_atomic_add_sub_return_the_old_value():
    a = _Acc()
    a.atomics()
    assert a.seen.to_numpy().tolist() == [0, 0, 1, 1, 2, 2, 3, 1]
    assert a.cnt.to_numpy().tolist()[:2] == [1, 3]


def test_cast_int_truncates_toward_zero():
    v = ti.Vector([-1.7, 2.9, -0.2, 5.0]).cast(int)
    assert [int(x) for x in v] == [-1, 2, 0, 5]
    assert ti.cast(-1.7, int) == -1 and ti.cast(np.float32(2.999), ti.i32) == 2
    assert isinstance(ti.cast(3, ti.f32), np.float32)


def test_prefix_sum_is_inclusive_and_in_place():
    f = ti.field(ti.i32, shape=6)
    f.from_numpy(np.array([3, 0, 2, 0, 0, 5], np.int32))
    ti.algorithms.PrefixSumExecutor(6).run(f)
    assert f.to_numpy().tolist() == [3, 3, 5, 5, 5, 10] and f.to_numpy().dtype == np.int32


_sf = ti.field(ti.f32, shape=2)
_si = ti.field(ti.i32, shape=2)
_res = ti.field(ti.f32, shape=6)


@ti.kernel
def _locals():
    a = 0.1 + 0.2                      # a Taichi local is f32: the f64 sum is rounded on assignment
    _res[0] = a
    b = _sf[0] / _si[0]                # f32 / i32 -> f32
    _res[1] = b
    c = _sf[0] * 0.1                   # f32 * python float -> f32 arithmetic (weak scalar)
    _res[2] = c
    q = _sf[1]
    _res[3] = ti.pow(1.0 - q, 3.0)
    _res[4] = ti.max(q, 0.75)
    _res[5] = ti.sqrt(_sf[0])


def test_locals_and_mixed_arithmetic_stay_f32():
    _sf.from_numpy(np.array([7.0, 0.3], np.float32))
    _si.from_numpy(np.array([3, 1], np.int32))
    _locals()
    r = _res.to_numpy()
    f = np.float32
    assert r[0] == f(0.1 + 0.2)
    assert r[1] == f(7.0) / f(3.0)
    assert r[2] == f(7.0) * f(0.1)
    assert r[3] == f(np.power(f(1.0) - f(0.3), f(3.0)))
    assert r[4] == f(0.75) and r[5] == f(np.sqrt(f(7.0)))


def test_vector_and_matrix_ops():
    a, b = ti.Vector([1.0, 2.0, 3.0]), ti.Vector([-2.0, 0.5, 4.0])
    assert a.dot(b) == np.float32(11.0) and a.norm_sqr() == np.float32(14.0)
    assert a.norm() == np.float32(np.sqrt(np.float32(14.0)))
    assert [float(x) for x in ti.math.cross(a, b)] == [6.5, -10.0, 4.5]
    op = a.outer_product(b).to_numpy()
    np.testing.assert_array_equal(op, np.outer([1, 2, 3], [-2, 0.5, 4]).astype(np.float32))
    M = ti.Matrix([[4.0, 1.0, 0.0], [1.0, 3.0, 0.5], [0.0, 0.5, 2.0]])
    inv = ti.math.inverse(M).to_numpy()
    np.testing.assert_allclose(inv, np.linalg.inv(M.to_numpy().astype(np.float64)), rtol=2e-6)
    assert inv.dtype == np.float32
    mv = M @ a
    assert [float(x) for x in mv] == [6.0, 8.5, 7.0]
    np.testing.assert_array_equal(ti.Matrix.identity(ti.f32, 3).to_numpy(), np.eye(3, dtype=np.float32))
