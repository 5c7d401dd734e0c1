"""Pins the oracle's leaf functions (3x3 algebra, material models).

Golden vectors: the reference's only unit test, solver-large/tests.c:17-24.
Reference code: oracle/_ref/libfearef.so is the reference's own
dense_matrix.c / fea_model.c / tests.c compiled in place (oracle/Makefile);
the restatement must agree with it BIT FOR BIT on random inputs.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob

A = [[1, 2, 0], [2, 0, 3], [0, 2, 3]]
B = [[0, 2, 1], [1, 1, 1], [3, 2, -1]]
AB = [[2, 4, 3], [9, 10, -1], [11, 8, -1]]        # tests.c:20
AtB = [[2, 4, 3], [6, 8, 0], [12, 9, 0]]          # tests.c:21
ABt = [[4, 3, 7], [3, 5, 3], [7, 5, 1]]           # tests.c:22

needs_ref = pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/libfearef.so not built (needs /root/reference)")


def _call3(fn, a, b):
    r = ob.M33()
    fn(ob.m33(a), ob.m33(b), r)
    return ob.m33_np(r)


def test_matmul_golden_vectors():
    L = ob.lib()
    assert np.array_equal(_call3(L.orc_mul3x3, A, B), np.array(AB, float))
    assert np.array_equal(_call3(L.orc_tmul3x3, A, B), np.array(AtB, float))
    assert np.array_equal(_call3(L.orc_mult3x3, A, B), np.array(ABt, float))


@needs_ref
def test_reference_self_test_passes():
    assert ob.ref().do_tests() == 1       # tests.c:53-56, run at fea_solver.c:76


@needs_ref
def test_dense_matrix_bit_exact_vs_reference():
    L, R = ob.lib(), ob.ref()
    rng = np.random.default_rng(20251004)
    for _ in range(200):
        a, b = rng.normal(size=(3, 3)), rng.normal(size=(3, 3))
        for mine, theirs in ((L.orc_mul3x3, R.matrix_mul3x3), (L.orc_tmul3x3, R.matrix_transpose_mul3x3),
                             (L.orc_mult3x3, R.matrix_transpose2_mul3x3)):
            assert np.array_equal(_call3(mine, a, b), _call3(theirs, a, b))
        assert L.orc_det3x3(ob.m33(a)) == R.det3x3(ob.m33(a))
        m1, m2 = ob.m33(a), ob.m33(a)
        d1, d2 = C.c_double(), C.c_double()
        assert L.orc_inv3x3(m1, C.byref(d1)) == R.inv3x3(m2, C.byref(d2))
        assert d1.value == d2.value
        assert np.array_equal(ob.m33_np(m1), ob.m33_np(m2))
        v1, v2 = rng.normal(size=37), rng.normal(size=37)
        p = lambda v: v.ctypes.data_as(C.POINTER(C.c_double))
        assert L.orc_cdot(p(v1), p(v2), 37) == R.cdot(p(v1), p(v2), 37)


@needs_ref
def test_singular_matrix_is_left_alone():
    L, R = ob.lib(), ob.ref()
    z = [[1, 2, 3], [2, 4, 6], [0, 1, 1]]
    m1, m2 = ob.m33(z), ob.m33(z)
    d1, d2 = C.c_double(), C.c_double()
    assert L.orc_inv3x3(m1, C.byref(d1)) == 0 == R.inv3x3(m2, C.byref(d2))
    assert np.array_equal(ob.m33_np(m1), np.array(z, float))


@needs_ref
@pytest.mark.parametrize("model", [0, 1])
def test_material_models_bit_exact_vs_reference(model):
    L, R = ob.lib(), ob.ref()
    fm = ob.RefModel()
    fm.model = model
    fm.parameters[0], fm.parameters[1] = 100.0, 37.5      # lambda, mu
    fm.parameters_count = 2
    R.fea_model_init(C.byref(fm), model)                  # binds stress / ctensor (fea_model.c:7-23)
    stress_t = C.CFUNCTYPE(None, C.POINTER(ob.RefModel), ob.M33, ob.M33)
    ctens_t = C.CFUNCTYPE(None, C.POINTER(ob.RefModel), ob.M33, ob.C4)
    ref_stress, ref_ctens = stress_t(fm.stress), ctens_t(fm.ctensor)
    par = (C.c_double * 10)(100.0, 37.5)
    rng = np.random.default_rng(7 + model)
    for _ in range(200):
        F = np.eye(3) + 0.3 * rng.normal(size=(3, 3))
        if np.linalg.det(F) <= 0.05:
            continue
        s1, s2 = ob.M33(), ob.M33()
        L.orc_stress(model, par, ob.m33(F), s1)
        ref_stress(C.byref(fm), ob.m33(F), s2)
        assert np.array_equal(ob.m33_np(s1), ob.m33_np(s2))
        c1, c2 = ob.C4(), ob.C4()
        L.orc_ctensor(model, par, ob.m33(F), c1)
        ref_ctens(C.byref(fm), ob.m33(F), c2)
        assert bytes(c1) == bytes(c2)


def test_neohookean_stress_known_values():
    """sigma = mu(B-I)/J + lambda ln J I/J on F = diag(k2,k1,k2), the
    closed form of exact-solutions/uniaxial/uniaxial_neohookean_bonet.m:20-37."""
    import math
    L = ob.lib()
    par = (C.c_double * 10)(100.0, 100.0)
    k1, k2 = 1.1, 0.97
    J = k1 * k2 * k2
    s = ob.M33()
    L.orc_stress(1, par, ob.m33(np.diag([k2, k1, k2])), s)
    S = ob.m33_np(s)
    assert S[1, 1] == pytest.approx((100 * (k1 ** 2 - 1) + 100 * math.log(J)) / J, rel=1e-14)
    assert S[0, 0] == pytest.approx((100 * (k2 ** 2 - 1) + 100 * math.log(J)) / J, rel=1e-13)
    assert S[0, 1] == 0 and S[1, 2] == 0
