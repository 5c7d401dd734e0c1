"""ctypes binding of the CHECKER (oracle/liboracle.so, oracle/_ref/libfearef.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing in the product imports this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libfearef.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
M33 = (C.c_double * 3) * 3
C4 = (((C.c_double * 3) * 3) * 3) * 3

TET10, TET4, HEX8 = 0, 1, 2


class BcNode(C.Structure):
    _fields_ = [("node", C.c_int), ("values", C.c_double * 3), ("type", C.c_int)]


class ElemTable(C.Structure):
    _fields_ = [("npe", C.c_int), ("ngauss", C.c_int), ("weight", C.c_double * 27),
                ("forms", (C.c_double * 10) * 27), ("dforms", ((C.c_double * 10) * 3) * 27)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_LIB)
        L.orc_cdot.restype = C.c_double
        L.orc_cdot.argtypes = [_dp, _dp, C.c_int]
        L.orc_det3x3.restype = C.c_double
        L.orc_inv3x3.restype = C.c_int
        L.orc_elem_table_init.argtypes = [C.POINTER(ElemTable), C.c_int, C.c_int]
        L.orc_solver_create.restype = C.c_void_p
        L.orc_solver_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _dp, C.c_int, _dp, C.c_int,
                                        C.POINTER(BcNode)]
        for name in ("orc_grads", "orc_detj", "orc_graddefs", "orc_stresses", "orc_values", "orc_forces",
                     "orc_solution"):
            getattr(L, name).restype = _dp
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("orc_offsets", "orc_indexes"):
            getattr(L, name).restype = _ip
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_nnz.argtypes = [C.c_void_p]
        L.orc_solver_free.argtypes = [C.c_void_p]
        L.orc_set_nodes.argtypes = [C.c_void_p, _dp]
        L.orc_get_nodes.argtypes = [C.c_void_p, _dp]
        L.orc_update_state.argtypes = [C.c_void_p]
        L.orc_element_stiffness.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_element_residual.argtypes = [C.c_void_p, C.c_int, _dp]
        for name in ("orc_create_stiffness", "orc_create_residual_forces", "orc_stash_stiffness",
                     "orc_restore_stiffness"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.orc_assemble_coloured.argtypes = [C.c_void_p, C.c_int]
        L.orc_assemble_coloured.restype = C.c_int
        L.orc_update_nodes_with_bc.argtypes = [C.c_void_p, C.c_double]
        L.orc_apply_prescribed_bc.argtypes = [C.c_void_p, C.c_double]
        L.orc_update_nodes_with_solution.argtypes = [C.c_void_p, _dp]
        L.orc_solve_slae.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, _dp]
        L.orc_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                _dp, C.c_int, _ip]
        L.orc_spmv.argtypes = [C.c_void_p, _dp, _dp]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_LIB)


class RefModel(C.Structure):
    """struct fea_model of the reference (fea_model.h:46-57) for calling
    oracle/_ref/libfearef.so -- the reference's own compiled leaf code."""
    _fields_ = [("model", C.c_int), ("parameters", C.c_double * 10), ("parameters_count", C.c_int),
                ("stress", C.c_void_p), ("ctensor", C.c_void_p)]


_ref = None


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_LIB)
        R.det3x3.restype = C.c_double
        R.cdot.restype = C.c_double
        R.cdot.argtypes = [_dp, _dp, C.c_int]
        R.inv3x3.restype = C.c_int
        R.do_tests.restype = C.c_int
        _ref = R
    return _ref


def m33(a):
    m = M33()
    for i in range(3):
        for j in range(3):
            m[i][j] = float(a[i][j])
    return m


def m33_np(m):
    return np.array([[m[i][j] for j in range(3)] for i in range(3)])


def elem_table(kind, ngauss):
    t = ElemTable()
    if lib().orc_elem_table_init(C.byref(t), kind, ngauss) != 0:
        raise ValueError("unsupported element table")
    npe = t.npe
    w = np.array([t.weight[g] for g in range(ngauss)])
    forms = np.array([[t.forms[g][i] for i in range(npe)] for g in range(ngauss)])
    dforms = np.array([[[t.dforms[g][d][i] for i in range(npe)] for d in range(3)] for g in range(ngauss)])
    return w, forms, dforms


class OracleSolver:
    """orc_solver with the reference's method names."""

    def __init__(self, deck):
        L = lib()
        self.deck = deck
        self.N, self.E = len(deck.nodes), len(deck.elements)
        self.npe, self.G = deck.nodes_per_element, deck.gauss_nodes_count
        self.ndof = 3 * self.N
        kind = {10: TET10, 4: TET4, 8: HEX8}[self.npe]
        nb = len(deck.presc_node)
        bc = (BcNode * max(nb, 1))()
        for i in range(nb):
            bc[i].node = int(deck.presc_node[i])
            bc[i].type = int(deck.presc_type[i])
            for j in range(3):
                bc[i].values[j] = float(deck.presc_values[i, j])
        par = np.zeros(10)
        par[:2] = deck.parameters[:2]
        conn = np.ascontiguousarray(deck.elements, dtype=np.int32)
        X0 = np.ascontiguousarray(deck.nodes, dtype=np.float64)
        self._p = L.orc_solver_create(self.N, self.E, kind, self.G, conn.ctypes.data_as(_ip),
                                      X0.ctypes.data_as(_dp), deck.model, par.ctypes.data_as(_dp), nb, bc)
        if not self._p:
            raise ValueError("oracle rejected the element type / Gauss rule")
        self._L = L

    def close(self):
        if self._p:
            self._L.orc_solver_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, fn, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(fn(self._p), (n,)).reshape(shape)

    def set_nodes(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._L.orc_set_nodes(self._p, x.ctypes.data_as(_dp))

    def nodes(self):
        x = np.zeros((self.N, 3))
        self._L.orc_get_nodes(self._p, x.ctypes.data_as(_dp))
        return x

    def update_state(self):
        return self._L.orc_update_state(self._p)

    def grads(self):
        return self._view(self._L.orc_grads, (self.E, self.G, 3, self.npe))

    def detj(self):
        return self._view(self._L.orc_detj, (self.E, self.G))

    def graddefs(self):
        return self._view(self._L.orc_graddefs, (self.E, self.G, 3, 3))

    def stresses(self):
        return self._view(self._L.orc_stresses, (self.E, self.G, 3, 3))

    def element_stiffness(self, e):
        n3 = 3 * self.npe
        kc, ks = np.zeros((n3, n3)), np.zeros((n3, n3))
        self._L.orc_element_stiffness(self._p, e, kc.ctypes.data_as(_dp), ks.ctypes.data_as(_dp))
        return kc, ks

    def element_residual(self, e):
        fe = np.zeros(3 * self.npe)
        self._L.orc_element_residual(self._p, e, fe.ctypes.data_as(_dp))
        return fe

    def nnz(self):
        return self._L.orc_nnz(self._p)

    def offsets(self):
        return np.ctypeslib.as_array(self._L.orc_offsets(self._p), (self.ndof + 1,))

    def indexes(self):
        return np.ctypeslib.as_array(self._L.orc_indexes(self._p), (self.nnz(),))

    def values(self):
        return self._view(self._L.orc_values, (self.nnz(),))

    def forces(self):
        return self._view(self._L.orc_forces, (self.ndof,))

    def solution(self):
        return self._view(self._L.orc_solution, (self.ndof,))

    def assemble_coloured(self, nthreads=0):
        """State + stiffness + residual with the elements coloured and every colour a parallel loop over the host's
        threads (OpenMP): the all-cores CPU baseline.  Returns the number of colours."""
        nc = self._L.orc_assemble_coloured(self._p, nthreads)
        if nc <= 0:
            raise RuntimeError(f"orc_assemble_coloured failed ({nc})")
        return nc

    def create_stiffness(self):
        self._L.orc_create_stiffness(self._p)

    def create_residual_forces(self):
        self._L.orc_create_residual_forces(self._p)

    def update_nodes_with_bc(self, lam):
        self._L.orc_update_nodes_with_bc(self._p, lam)

    def apply_prescribed_bc(self, lam):
        self._L.orc_apply_prescribed_bc(self._p, lam)

    def update_nodes_with_solution(self, u=None):
        u = self.solution().copy() if u is None else np.ascontiguousarray(u, dtype=np.float64)
        self._L.orc_update_nodes_with_solution(self._p, u.ctypes.data_as(_dp))

    def stash_stiffness(self):
        self._L.orc_stash_stiffness(self._p)

    def restore_stiffness(self):
        self._L.orc_restore_stiffness(self._p)

    def solve_slae(self, solver_type, tol=1e-14, max_iter=20000):
        res = C.c_double(0)
        it = self._L.orc_solve_slae(self._p, solver_type, tol, max_iter, C.byref(res))
        return it, res.value

    def energy(self):
        return float(self._L.orc_cdot(self._L.orc_forces(self._p), self._L.orc_solution(self._p), self.ndof))

    def solve(self, load_increments, max_newton, modified_newton, desired_tolerance, solver_type,
              solver_tolerance=1e-14, solver_max_iter=20000):
        cap = load_increments * max_newton
        tol_log = np.zeros(cap)
        its = np.zeros(load_increments, dtype=np.int32)
        done = self._L.orc_solve(self._p, load_increments, max_newton, int(modified_newton), desired_tolerance,
                                 solver_type, solver_tolerance, solver_max_iter, tol_log.ctypes.data_as(_dp), cap,
                                 its.ctypes.data_as(_ip))
        n = int(its.sum())
        return done, its, tol_log[:n]

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.ndof)
        self._L.orc_spmv(self._p, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp))
        return y
