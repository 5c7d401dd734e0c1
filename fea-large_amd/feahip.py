"""ctypes mirror of include/fea_hip.h and host/fea_host.h.

Python is plumbing here (tests, bench): every call goes straight through the
C ABI of libfeahip.so -- the same symbols a C host (the reference's solve(),
solver-large/fea_solver.c:130-242) binds.  There is no Python or CPU
implementation behind these names: if the shared library is missing, or no
HIP device is visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEAHIP_LIB: measurement tooling only (the diagnostic build libfeahip_dbg.so); still a HIP library, never a fallback
LIB_PATH = os.environ.get("FEAHIP_LIB") or os.path.join(_HERE, "libfeahip.so")
HOST_LIB_PATH = os.path.join(_HERE, "libfeahost.so")

MODEL_A5, MODEL_COMPRESSIBLE_NEOHOOKEAN = 0, 1
CG, PCG_ILU, CHOLESKY = 0, 1, 2
ASM_AUTO, ASM_ROWOWNER, ASM_ATOMIC, ASM_PATCH, ASM_STAGED, ASM_PAIRED, ASM_PIPELINED, ASM_SHARED, ASM_GATHER = 0, 1, 2, 3, 4, 5, 6, 7, 8
TETRAHEDRA10, TETRAHEDRA4, HEXAHEDRA8 = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/fea_hip.h declares, with its argument types
ABI = {
    "feahip_create": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _ip, _dp,
                      C.c_int, _dp, C.c_int, C.c_int, _ip, _ip, _dp],
    "feahip_destroy": [C.c_void_p],
    "feahip_last_error": [C.c_void_p],
    "feahip_create_error": [],
    "feahip_update_nodes_with_bc": [C.c_void_p, C.c_double],
    "feahip_update_state": [C.c_void_p, _ip],
    "feahip_create_stiffness": [C.c_void_p],
    "feahip_create_residual_forces": [C.c_void_p],
    "feahip_create_stiffness_and_residual": [C.c_void_p],
    "feahip_stash_stiffness": [C.c_void_p],
    "feahip_restore_stiffness": [C.c_void_p],
    "feahip_apply_prescribed_bc": [C.c_void_p, C.c_double],
    "feahip_solve_slae": [C.c_void_p, C.c_int, C.c_double, C.c_int, _ip, _dp],
    "feahip_energy": [C.c_void_p, _dp],
    "feahip_update_nodes_with_solution": [C.c_void_p, _dp],
    "feahip_solve": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, _dp,
                     C.c_int, _ip, _ip],
    "feahip_set_nodes": [C.c_void_p, _dp],
    "feahip_get_nodes": [C.c_void_p, _dp],
    "feahip_get_forces": [C.c_void_p, _dp],
    "feahip_set_forces": [C.c_void_p, _dp],
    "feahip_get_solution": [C.c_void_p, _dp],
    "feahip_get_graddefs": [C.c_void_p, _dp],
    "feahip_get_stresses": [C.c_void_p, _dp],
    "feahip_get_shape_gradients": [C.c_void_p, _dp, _dp],
    "feahip_matrix_nnz": [C.c_void_p, C.POINTER(C.c_longlong)],
    "feahip_get_matrix_yale": [C.c_void_p, _ip, _ip, _dp],
    "feahip_get_matrix_yale64": [C.c_void_p, C.POINTER(C.c_longlong), _ip, _dp],
    "feahip_spmv": [C.c_void_p, _dp, _dp],
    "feahip_set_assembly": [C.c_void_p, C.c_int],
    "feahip_set_preconditioner": [C.c_void_p, C.c_int],
    "feahip_set_line_search": [C.c_void_p, C.c_int],
    "feahip_set_pcg_variant": [C.c_void_p, C.c_int],
    "feahip_set_row_shard": [C.c_void_p, C.c_int, C.c_int],
    "feahip_comm_unique_id": [C.c_void_p, C.c_int],
    "feahip_comm_init": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "feahip_owned_rows": [C.c_void_p, _ip, _ip],
    "feahip_group_init": [C.POINTER(C.c_void_p), C.c_int],
    "feahip_group_solve_slae": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_double, C.c_int, _ip, _dp],
    "feahip_group_energy": [C.POINTER(C.c_void_p), C.c_int, _dp],
    "feahip_group_update_nodes_with_solution": [C.POINTER(C.c_void_p), C.c_int],
    "feahip_group_solve": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                           C.c_double, C.c_int, _dp, C.c_int, _ip, _ip],
    "feahip_shard_plan": [C.c_int, C.c_int, C.c_int, _ip, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _ip, _ip],
    "feahip_sync": [C.c_void_p],
    "feahip_time_kernel": [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp],
    "feahip_sizes": [C.c_void_p, C.POINTER(C.c_longlong)],
    "feahip_host_gather_stats": [C.c_int, C.c_int, C.c_int, _ip, C.POINTER(C.c_longlong), _ip],
    "feahip_host_assembly_digest": [C.c_int, C.c_int, C.c_int, _ip, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), _ip],
    "feahip_assembly_in_use": [C.c_void_p, _ip],
    "feahip_node_numbering": [C.c_void_p, _ip],
    "feahip_create_rank": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _ip, _dp,
                           C.c_int, _dp, C.c_int, C.c_int, _ip, _ip, _dp],
    "feahip_rank_counts": [C.c_void_p, C.POINTER(C.c_longlong)],
    "feahip_rank_maps": [C.c_void_p, _ip, _ip],
    "feahip_host_rank_plan": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _dp, _ip, _ip, _ip, _ip, _ip, _ip],
    "feahip_host_rank_mesh": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _dp, C.POINTER(C.c_longlong), _ip, _ip,
                              C.POINTER(C.c_longlong), _ip],
    "feahip_copy_bandwidth": [C.c_void_p, C.c_longlong, _dp],
    "feahip_copy_bandwidth_detail": [C.c_void_p, C.c_longlong, _dp],
    "feahip_assembly_stats": [C.c_void_p, _dp],
    "feahip_device_layout": [C.c_void_p, C.POINTER(C.c_longlong)],
    "feahip_host_numbering": [C.c_int, C.c_int, C.c_int, _ip, _dp, _ip],
}

_lib = None
_host = None


class FeaHipError(RuntimeError):
    pass


def load_library():
    """dlopen libfeahip.so and type every ABI symbol (no device is touched)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FeaHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no fallback implementation")
        lib = C.CDLL(LIB_PATH)
        for name, args in ABI.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is absent
            fn.argtypes = args
            fn.restype = C.c_char_p if name in ("feahip_last_error", "feahip_create_error") else (
                None if name == "feahip_destroy" else C.c_int)
        _lib = lib
    return _lib


class FeaDeck(C.Structure):
    """struct fea_deck of host/fea_host.h."""
    _fields_ = [
        ("model", C.c_int), ("parameters", C.c_double * 10), ("parameters_count", C.c_int),
        ("solver_type", C.c_int), ("solver_tolerance", C.c_double), ("solver_max_iter", C.c_int),
        ("ele_type", C.c_int), ("load_increments_count", C.c_int), ("desired_tolerance", C.c_double),
        ("max_newton_count", C.c_int), ("linesearch_max", C.c_int), ("arclength_max", C.c_int),
        ("modified_newton", C.c_int), ("nodes_per_element", C.c_int), ("gauss_nodes_count", C.c_int),
        ("nodes_count", C.c_int), ("nodes", _dp), ("elements_count", C.c_int), ("elements", _ip),
        ("prescribed_nodes_count", C.c_int), ("presc_node", _ip), ("presc_type", _ip), ("presc_values", _dp),
    ]


class StepSnapshot(C.Structure):
    """struct fea_step_snapshot of host/fea_host.h."""
    _fields_ = [("nodes", _dp), ("stress0", _dp)]


def export_gmsh(path, deck, nodes_steps, stress0_steps):
    """fea_export_gmsh: nodes_steps[k] is [N][3], stress0_steps[k] is [E][3][3] after load step k+1."""
    h = load_host_library()
    n = len(nodes_steps)
    keep = [(np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64))
            for a, b in zip(nodes_steps, stress0_steps)]
    arr = (StepSnapshot * max(n, 1))()
    for k, (a, b) in enumerate(keep):
        arr[k].nodes, arr[k].stress0 = _d(a), _d(b)
    fd = deck.to_struct()
    if h.fea_export_gmsh(os.fsencode(path), C.byref(fd), arr, n) != 0:
        raise FeaHipError(f"could not write {path}")


def load_host_library():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise FeaHipError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
        load_library()   # libfeahost.so links against libfeahip.so
        h = C.CDLL(HOST_LIB_PATH)
        h.fea_deck_load.argtypes = [C.c_char_p, C.POINTER(FeaDeck), C.c_char_p, C.c_int]
        h.fea_deck_load.restype = C.c_int
        h.fea_deck_free.argtypes = [C.POINTER(FeaDeck)]
        h.fea_deck_free.restype = None
        h.fea_deck_save.argtypes = [C.c_char_p, C.POINTER(FeaDeck)]
        h.fea_deck_save.restype = C.c_int
        h.fea_element_tables.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        h.fea_element_tables.restype = C.c_int
        h.fea_deck_create_solver.argtypes = [C.POINTER(FeaDeck), C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        h.fea_deck_create_solver.restype = C.c_int
        h.fea_solve.argtypes = [C.POINTER(FeaDeck), C.c_void_p, C.c_void_p, _dp, C.c_int]
        h.fea_solve.restype = C.c_int
        h.fea_export_gmsh.argtypes = [C.c_char_p, C.POINTER(FeaDeck), C.POINTER(StepSnapshot), C.c_int]
        h.fea_export_gmsh.restype = C.c_int
        h.fea_export_name.argtypes = [C.c_char_p, C.c_char_p]
        h.fea_export_name.restype = None
        _host = h
    return _host


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def element_tables(ele_type, gauss_count):
    """(weights[G], forms[G][npe], dforms[G][3][npe]) from the host plug-in."""
    h = load_host_library()
    npe = {TETRAHEDRA10: 10, TETRAHEDRA4: 4, HEXAHEDRA8: 8}[ele_type]
    w = np.zeros(gauss_count)
    forms = np.zeros((gauss_count, npe))
    dforms = np.zeros((gauss_count, 3, npe))
    rc = h.fea_element_tables(ele_type, gauss_count, _d(w), _d(forms), _d(dforms))
    if rc < 0:
        raise FeaHipError(f"unsupported element type {ele_type} with {gauss_count} Gauss points")
    return w, forms, dforms


class Deck:
    """A task deck as numpy arrays (fields named as the reference's structs)."""

    def __init__(self, **kw):
        self.model = kw.get("model", MODEL_COMPRESSIBLE_NEOHOOKEAN)
        self.parameters = np.array(kw.get("parameters", [100.0, 100.0]), dtype=np.float64)
        self.solver_type = kw.get("solver_type", CG)
        self.solver_tolerance = kw.get("solver_tolerance", 1e-14)
        self.solver_max_iter = kw.get("solver_max_iter", 20000)
        self.ele_type = kw.get("ele_type", TETRAHEDRA10)
        self.load_increments_count = kw.get("load_increments_count", 1)
        self.desired_tolerance = kw.get("desired_tolerance", 1e-8)
        self.max_newton_count = kw.get("max_newton_count", 20)
        self.modified_newton = kw.get("modified_newton", True)
        self.gauss_nodes_count = kw.get("gauss_nodes_count", 5)
        self.nodes = np.ascontiguousarray(kw["nodes"], dtype=np.float64)
        self.elements = np.ascontiguousarray(kw["elements"], dtype=np.int32)
        self.nodes_per_element = self.elements.shape[1]
        self.presc_node = np.ascontiguousarray(kw.get("presc_node", []), dtype=np.int32)
        self.presc_type = np.ascontiguousarray(kw.get("presc_type", []), dtype=np.int32)
        self.presc_values = np.ascontiguousarray(kw.get("presc_values", np.zeros((0, 3))), dtype=np.float64).reshape(-1, 3)

    @staticmethod
    def load(path):
        """Reads a .sexp deck through the product's C reader."""
        h = load_host_library()
        fd = FeaDeck()
        err = C.create_string_buffer(512)
        if h.fea_deck_load(os.fsencode(path), C.byref(fd), err, 512) != 0:
            raise FeaHipError(f"{path}: {err.value.decode()}")
        try:
            n, e, npe, nb = fd.nodes_count, fd.elements_count, fd.nodes_per_element, fd.prescribed_nodes_count
            deck = Deck(
                model=fd.model, parameters=[fd.parameters[0], fd.parameters[1]], solver_type=fd.solver_type,
                solver_tolerance=fd.solver_tolerance, solver_max_iter=fd.solver_max_iter, ele_type=fd.ele_type,
                load_increments_count=fd.load_increments_count, desired_tolerance=fd.desired_tolerance,
                max_newton_count=fd.max_newton_count, modified_newton=bool(fd.modified_newton),
                gauss_nodes_count=fd.gauss_nodes_count,
                nodes=np.ctypeslib.as_array(fd.nodes, (n, 3)).copy(),
                elements=np.ctypeslib.as_array(fd.elements, (e, npe)).copy(),
                presc_node=np.ctypeslib.as_array(fd.presc_node, (nb,)).copy() if nb else [],
                presc_type=np.ctypeslib.as_array(fd.presc_type, (nb,)).copy() if nb else [],
                presc_values=np.ctypeslib.as_array(fd.presc_values, (nb, 3)).copy() if nb else np.zeros((0, 3)))
            deck.linesearch_max, deck.arclength_max = fd.linesearch_max, fd.arclength_max
            return deck
        finally:
            h.fea_deck_free(C.byref(fd))

    def to_struct(self):
        fd = FeaDeck()
        fd.model = self.model
        fd.parameters[0], fd.parameters[1] = float(self.parameters[0]), float(self.parameters[1])
        fd.parameters_count = 2
        fd.solver_type, fd.solver_tolerance, fd.solver_max_iter = self.solver_type, self.solver_tolerance, self.solver_max_iter
        fd.ele_type = self.ele_type
        fd.load_increments_count, fd.desired_tolerance = self.load_increments_count, self.desired_tolerance
        fd.max_newton_count, fd.modified_newton = self.max_newton_count, int(self.modified_newton)
        fd.nodes_per_element, fd.gauss_nodes_count = self.nodes_per_element, self.gauss_nodes_count
        fd.nodes_count, fd.nodes = len(self.nodes), _d(self.nodes)
        fd.elements_count, fd.elements = len(self.elements), _i(self.elements)
        fd.prescribed_nodes_count = len(self.presc_node)
        fd.presc_node, fd.presc_type, fd.presc_values = _i(self.presc_node), _i(self.presc_type), _d(self.presc_values)
        return fd

    def save(self, path):
        h = load_host_library()
        fd = self.to_struct()
        if h.fea_deck_save(os.fsencode(path), C.byref(fd)) != 0:
            raise FeaHipError(f"could not write {path}")


class FeaSolver:
    """The `fea_solver` object of the reference, backed by the HIP context.

    Method names are the reference's solver_* functions minus the prefix
    (fea_solver.h:497-610)."""

    def __init__(self, deck, device=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self.deck = deck
        w, _, dforms = element_tables(deck.ele_type, deck.gauss_nodes_count)
        self.N, self.E = len(deck.nodes), len(deck.elements)
        self.npe, self.G = deck.nodes_per_element, deck.gauss_nodes_count
        self.ndof = 3 * self.N
        par = np.zeros(10)
        par[:2] = deck.parameters[:2]
        rc = self._lib.feahip_create(
            C.byref(self._ctx), device, self.N, self.E, self.npe, self.G, _d(w), _d(dforms), _i(deck.elements),
            _d(deck.nodes), deck.model, _d(par), 2, len(deck.presc_node), _i(deck.presc_node), _i(deck.presc_type),
            _d(deck.presc_values))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise FeaHipError(f"feahip_create failed ({rc}): {self._lib.feahip_create_error().decode()}")

    def close(self):
        if self._ctx:
            self._lib.feahip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise FeaHipError(f"libfeahip error {rc}: {self._lib.feahip_last_error(self._ctx).decode()}")

    # ---- the calls of solve() ------------------------------------------
    def update_nodes_with_bc(self, lam):
        self._chk(self._lib.feahip_update_nodes_with_bc(self._ctx, lam))

    def update_state(self):
        bad = C.c_int(0)
        self._chk(self._lib.feahip_update_state(self._ctx, C.byref(bad)))
        return bad.value

    def create_stiffness(self):
        self._chk(self._lib.feahip_create_stiffness(self._ctx))

    def create_residual_forces(self):
        self._chk(self._lib.feahip_create_residual_forces(self._ctx))

    def create_stiffness_and_residual(self):
        self._chk(self._lib.feahip_create_stiffness_and_residual(self._ctx))

    def stash_stiffness(self):
        self._chk(self._lib.feahip_stash_stiffness(self._ctx))

    def restore_stiffness(self):
        self._chk(self._lib.feahip_restore_stiffness(self._ctx))

    def apply_prescribed_bc(self, lam):
        self._chk(self._lib.feahip_apply_prescribed_bc(self._ctx, lam))

    def solve_slae(self, solver_type=None, tolerance=None, max_iterations=None):
        it, res = C.c_int(0), C.c_double(0)
        d = self.deck
        self._chk(self._lib.feahip_solve_slae(
            self._ctx, d.solver_type if solver_type is None else solver_type,
            d.solver_tolerance if tolerance is None else tolerance,
            d.solver_max_iter if max_iterations is None else max_iterations, C.byref(it), C.byref(res)))
        return it.value, res.value

    def energy(self):
        t = C.c_double(0)
        self._chk(self._lib.feahip_energy(self._ctx, C.byref(t)))
        return t.value

    def update_nodes_with_solution(self, u=None):
        if u is None:
            self._chk(self._lib.feahip_update_nodes_with_solution(self._ctx, None))
        else:
            u = np.ascontiguousarray(u, dtype=np.float64)
            self._chk(self._lib.feahip_update_nodes_with_solution(self._ctx, _d(u)))

    def solve(self, load_increments=None, max_newton=None, modified_newton=None, desired_tolerance=None,
              solver_type=None, solver_tolerance=None, solver_max_iter=None, line_search=None):
        d = self.deck
        if line_search is not None:
            self.set_line_search(line_search)
        li = d.load_increments_count if load_increments is None else load_increments
        mn = d.max_newton_count if max_newton is None else max_newton
        cap = li * mn
        tol_log = np.zeros(cap)
        its = np.zeros(li, dtype=np.int32)
        done = C.c_int(0)
        self._chk(self._lib.feahip_solve(
            self._ctx, li, mn, int(d.modified_newton if modified_newton is None else modified_newton),
            d.desired_tolerance if desired_tolerance is None else desired_tolerance,
            d.solver_type if solver_type is None else solver_type,
            d.solver_tolerance if solver_tolerance is None else solver_tolerance,
            d.solver_max_iter if solver_max_iter is None else solver_max_iter,
            _d(tol_log), cap, _i(its), C.byref(done)))
        n = int(its[:max(done.value, 0) + (1 if done.value < li else 0)].sum())
        return done.value, its, tol_log[:n]

    # ---- views -----------------------------------------------------------
    def set_nodes(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.N, 3)
        self._chk(self._lib.feahip_set_nodes(self._ctx, _d(x)))

    def nodes(self):
        x = np.zeros((self.N, 3))
        self._chk(self._lib.feahip_get_nodes(self._ctx, _d(x)))
        return x

    def shape_gradients(self):
        """(grads[E][G][3][npe], detJ[E][G]) of the current configuration."""
        g = np.zeros((self.E, self.G, 3, self.npe))
        d = np.zeros((self.E, self.G))
        self._chk(self._lib.feahip_get_shape_gradients(self._ctx, _d(g), _d(d)))
        return g, d

    def forces(self):
        f = np.zeros(self.ndof)
        self._chk(self._lib.feahip_get_forces(self._ctx, _d(f)))
        return f

    def set_forces(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        assert f.shape == (self.ndof,)
        self._chk(self._lib.feahip_set_forces(self._ctx, _d(f)))

    def solution(self):
        u = np.zeros(self.ndof)
        self._chk(self._lib.feahip_get_solution(self._ctx, _d(u)))
        return u

    def graddefs(self):
        F = np.zeros((self.E, self.G, 3, 3))
        self._chk(self._lib.feahip_get_graddefs(self._ctx, _d(F)))
        return F

    def stresses(self):
        S = np.zeros((self.E, self.G, 3, 3))
        self._chk(self._lib.feahip_get_stresses(self._ctx, _d(S)))
        return S

    def matrix_yale(self):
        nnz = C.c_longlong(0)
        self._chk(self._lib.feahip_matrix_nnz(self._ctx, C.byref(nnz)))
        off = np.zeros(self.ndof + 1, dtype=np.int32)
        idx = np.zeros(nnz.value, dtype=np.int32)
        val = np.zeros(nnz.value)
        self._chk(self._lib.feahip_get_matrix_yale(self._ctx, _i(off), _i(idx), _d(val)))
        return off, idx, val

    def matrix_yale64(self):
        """The same with 64-bit offsets (feahip_get_matrix_yale refuses 2^31 or more scalar non-zeros)."""
        nnz = C.c_longlong(0)
        self._chk(self._lib.feahip_matrix_nnz(self._ctx, C.byref(nnz)))
        off = np.zeros(self.ndof + 1, dtype=np.int64)
        idx = np.zeros(nnz.value, dtype=np.int32)
        val = np.zeros(nnz.value)
        self._chk(self._lib.feahip_get_matrix_yale64(self._ctx, off.ctypes.data_as(C.POINTER(C.c_longlong)), _i(idx), _d(val)))
        return off, idx, val

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.ndof)
        self._chk(self._lib.feahip_spmv(self._ctx, _d(x), _d(y)))
        return y

    # ---- tuning / measurement ------------------------------------------
    def set_preconditioner(self, kind):
        self._chk(self._lib.feahip_set_preconditioner(self._ctx, kind))

    def set_pcg_variant(self, variant):
        self._chk(self._lib.feahip_set_pcg_variant(self._ctx, variant))

    def set_line_search(self, max_iterations):
        self._chk(self._lib.feahip_set_line_search(self._ctx, max_iterations))

    def set_assembly(self, strategy):
        self._chk(self._lib.feahip_set_assembly(self._ctx, strategy))

    def set_row_shard(self, rank, nranks):
        self._chk(self._lib.feahip_set_row_shard(self._ctx, rank, nranks))

    def owned_rows(self):
        """[row0, row1) in LIBRARY node ids (see node_numbering / owned_nodes)."""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self._lib.feahip_owned_rows(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def node_numbering(self):
        """library id of every caller's node (the identity when the library kept the caller's numbering)."""
        out = np.empty(self.N, dtype=np.int32)
        self._chk(self._lib.feahip_node_numbering(self._ctx, _i(out)))
        return out

    def owned_nodes(self):
        """the caller's ids of the nodes this rank owns, ascending."""
        r0, r1 = self.owned_rows()
        lib = self.node_numbering()
        return np.nonzero((lib >= r0) & (lib < r1))[0]

    def owned_dofs(self):
        """the caller's dof indices (node * 3 + axis) of the nodes this rank owns, ascending."""
        return (3 * self.owned_nodes()[:, None] + np.arange(3)[None, :]).ravel()

    def comm_init(self, rank, nranks, unique_id):
        """unique_id: the 128 bytes rank 0 obtained from comm_unique_id()."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._lib.feahip_comm_init(self._ctx, rank, nranks, buf))

    def sync(self):
        self._chk(self._lib.feahip_sync(self._ctx))

    def time_kernel(self, what, warmup=2, iters=10):
        ms = C.c_double(0)
        self._chk(self._lib.feahip_time_kernel(self._ctx, what, warmup, iters, C.byref(ms)))
        return ms.value

    def sizes(self):
        o = (C.c_longlong * 8)()
        self._chk(self._lib.feahip_sizes(self._ctx, o))
        keys = ["N", "E", "npe", "G", "nnzb", "nchunks", "aux_bytes", "max_rowlen"]
        return dict(zip(keys, [int(v) for v in o]))

    def copy_bandwidth(self, nbytes=1 << 30):
        """GB/s of a plain device copy (read + written) on this box."""
        v = C.c_double(0)
        self._chk(self._lib.feahip_copy_bandwidth(self._ctx, nbytes, C.byref(v)))
        return v.value

    def assembly_stats(self):
        """dict: element evaluations per element of the gather chunks, chunks, chunks with their predecessor's map
        words, map bytes (zeros when another strategy runs)."""
        v = (C.c_double * 4)()
        self._chk(self._lib.feahip_assembly_stats(self._ctx, v))
        return {"evals_per_element": float(v[0]), "chunks": int(v[1]), "chunks_with_predecessors_words": int(v[2]), "map_bytes": int(v[3])}

    def copy_bandwidth_detail(self, nbytes=1 << 30):
        """The same four ways: [one 16-byte load in flight per lane, four in flight per lane, hipMemcpyDtoDAsync,
        four in flight non-temporal]."""
        v = (C.c_double * 4)()
        self._chk(self._lib.feahip_copy_bandwidth_detail(self._ctx, nbytes, v))
        return [float(x) for x in v]

    def device_layout(self):
        o = (C.c_longlong * 4)()
        self._chk(self._lib.feahip_device_layout(self._ctx, o))
        return {"K": int(o[0]), "colidx": int(o[1]), "p": int(o[2]), "q": int(o[3])}

    def assembly_in_use(self):
        """The strategy the most recent assembly launch ran (what ASM_AUTO resolved to)."""
        v = C.c_int(0)
        self._chk(self._lib.feahip_assembly_in_use(self._ctx, C.byref(v)))
        return v.value


def host_numbering(elements, nodes):
    """(library id of every node, renumbered?) that feahip_create would choose for this mesh; host only."""
    el = np.ascontiguousarray(elements, dtype=np.int32)
    x = np.ascontiguousarray(nodes, dtype=np.float64)
    out = np.empty(len(x), dtype=np.int32)
    rc = load_library().feahip_host_numbering(len(x), el.shape[0], el.shape[1], _i(el), _d(x), _i(out))
    if rc < 0:
        raise FeaHipError(f"feahip_host_numbering failed ({rc})")
    return out, bool(rc)


def host_gather_stats(elements, n_nodes):
    """Shape of the GATHER maps of a mesh (4-, 10- or 8-node elements) in the numbering given (host only): dict + chunks by row count."""
    el = np.ascontiguousarray(elements, dtype=np.int32)
    st = np.zeros(8, dtype=np.int64)
    hist = np.zeros(65, dtype=np.int32)
    rc = load_library().feahip_host_gather_stats(n_nodes, el.shape[0], el.shape[1], _i(el), st.ctypes.data_as(C.POINTER(C.c_longlong)), _i(hist))
    if rc:
        raise FeaHipError(f"feahip_host_gather_stats failed ({rc})")
    return {"chunks": int(st[0]), "evals": int(st[1]), "elements": int(st[2]), "rows": int(st[3]),
            "evals_per_element": float(st[1]) / float(st[2]), "rows_per_chunk": float(st[3]) / float(st[0]),
            "chunks_with_predecessors_words": int(st[4]), "map_bytes": int(st[5]),
            "chunks_with_long_block_lists": int(st[6]), "chunks_with_long_diagonal_lists": int(st[7])}, hist


def host_assembly_digest(elements, n_nodes, rank=0, nranks=1):
    """(rowhash[N] uint64, (row0, row1)) of the assembly maps `rank` of `nranks` builds on the host (no device)."""
    el = np.ascontiguousarray(elements, dtype=np.int32)
    h = np.zeros(n_nodes, dtype=np.uint64)
    rows = np.zeros(2, dtype=np.int32)
    rc = load_library().feahip_host_assembly_digest(n_nodes, el.shape[0], el.shape[1], _i(el), rank, nranks,
                                                    h.ctypes.data_as(C.POINTER(C.c_ulonglong)), _i(rows))
    if rc:
        raise FeaHipError(f"feahip_host_assembly_digest failed ({rc})")
    return h, (int(rows[0]), int(rows[1]))


def comm_unique_id():
    buf = C.create_string_buffer(128)
    n = load_library().feahip_comm_unique_id(buf, 128)
    if n <= 0:
        raise FeaHipError(f"feahip_comm_unique_id failed ({n})")
    return bytes(buf.raw[:128])


def shard_plan(deck, rank, nranks):
    """Host-only halo plan of one rank: dict(row0,row1,peers,send,recv) with
    per-peer node-id arrays."""
    lib = load_library()
    el = np.ascontiguousarray(deck.elements, dtype=np.int32)
    counts = np.zeros(5, dtype=np.int32)
    args = (len(deck.nodes), len(el), el.shape[1], _i(el), rank, nranks)
    rc = lib.feahip_shard_plan(*args, _i(counts), None, None, None, None, None)
    if rc != 0:
        raise FeaHipError(f"feahip_shard_plan failed ({rc})")
    npeer, nsend, nrecv = int(counts[0]), int(counts[1]), int(counts[2])
    peers = np.zeros(max(npeer, 1), dtype=np.int32)
    soff, roff = np.zeros(npeer + 1, dtype=np.int32), np.zeros(npeer + 1, dtype=np.int32)
    sidx, ridx = np.zeros(max(nsend, 1), dtype=np.int32), np.zeros(max(nrecv, 1), dtype=np.int32)
    rc = lib.feahip_shard_plan(*args, _i(counts), _i(peers), _i(soff), _i(roff), _i(sidx), _i(ridx))
    if rc != 0:
        raise FeaHipError(f"feahip_shard_plan failed ({rc})")
    return {"row0": int(counts[3]), "row1": int(counts[4]), "peers": [int(p) for p in peers[:npeer]],
            "send": [sidx[soff[k]:soff[k + 1]].copy() for k in range(npeer)],
            "recv": [ridx[roff[k]:roff[k + 1]].copy() for k in range(npeer)]}


class RankSolver(FeaSolver):
    """One rank's context of a sharded run (feahip_create_rank): the sub-mesh this rank holds, locally indexed (owned
    nodes first).  node_global / elem_global say which nodes and elements of the deck the local ones are; every
    node- or element-indexed method speaks local indices."""

    def __init__(self, deck, rank, nranks, device=0):            # noqa: super().__init__ not called: another constructor of the ABI
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self.deck = deck
        w, _, dforms = element_tables(deck.ele_type, deck.gauss_nodes_count)
        self.npe, self.G = deck.nodes_per_element, deck.gauss_nodes_count
        par = np.zeros(10)
        par[:2] = deck.parameters[:2]
        rc = self._lib.feahip_create_rank(
            C.byref(self._ctx), device, rank, nranks, len(deck.nodes), len(deck.elements), self.npe, self.G, _d(w), _d(dforms),
            _i(deck.elements), _d(deck.nodes), deck.model, _d(par), 2, len(deck.presc_node), _i(deck.presc_node),
            _i(deck.presc_type), _d(deck.presc_values))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise FeaHipError(f"feahip_create_rank failed ({rc}): {self._lib.feahip_create_error().decode()}")
        o = (C.c_longlong * 8)()
        self._chk(self._lib.feahip_rank_counts(self._ctx, o))
        self.N, self.n_own, self.E, self.N_global = int(o[0]), int(o[1]), int(o[2]), int(o[3])
        self.nnzb_local, self.nnzb_owned, self.rows_sent, self.rows_received = int(o[4]), int(o[5]), int(o[6]), int(o[7])
        self.ndof = 3 * self.N
        self.node_global = np.empty(self.N, dtype=np.int32)
        self.elem_global = np.empty(self.E, dtype=np.int32)
        self._chk(self._lib.feahip_rank_maps(self._ctx, _i(self.node_global), _i(self.elem_global)))


def host_rank_mesh(deck, rank, nranks, pattern=False):
    """Host only: what rank `rank` of `nranks` would hold.  dict of counts; with pattern=True also node_global, elem_global
    and the block rows of the owned nodes (rowptr, colidx in the deck's node ids)."""
    lib = load_library()
    el = np.ascontiguousarray(deck.elements, dtype=np.int32)
    x = np.ascontiguousarray(deck.nodes, dtype=np.float64)
    cnt = (C.c_longlong * 8)()
    args = (rank, nranks, len(x), el.shape[0], el.shape[1], _i(el), _d(x))
    rc = lib.feahip_host_rank_mesh(*args, cnt, None, None, None, None)
    if rc != 0:
        raise FeaHipError(f"feahip_host_rank_mesh failed ({rc})")
    names = ("local_nodes", "owned_nodes", "local_elements", "blocks_owned_rows", "blocks_local_rows", "peers", "rows_sent", "rows_received")
    out = {k: int(v) for k, v in zip(names, cnt)}
    if pattern:
        ng = np.empty(out["local_nodes"], dtype=np.int32); eg = np.empty(out["local_elements"], dtype=np.int32)
        rp = np.empty(out["owned_nodes"] + 1, dtype=np.int64); ci = np.empty(max(out["blocks_owned_rows"], 1), dtype=np.int32)
        rc = lib.feahip_host_rank_mesh(*args, cnt, _i(ng), _i(eg), rp.ctypes.data_as(C.POINTER(C.c_longlong)), _i(ci))
        if rc != 0:
            raise FeaHipError(f"feahip_host_rank_mesh failed ({rc})")
        out.update(node_global=ng, elem_global=eg, rowptr=rp, colidx=ci[:out["blocks_owned_rows"]])
    return out


def host_rank_plan(deck, rank, nranks):
    """Host only: halo plan of one rank's sub-mesh in the deck's node ids: dict(peers, send, recv)."""
    lib = load_library()
    el = np.ascontiguousarray(deck.elements, dtype=np.int32)
    x = np.ascontiguousarray(deck.nodes, dtype=np.float64)
    cnt = np.zeros(3, dtype=np.int32)
    args = (rank, nranks, len(x), el.shape[0], el.shape[1], _i(el), _d(x))
    if lib.feahip_host_rank_plan(*args, _i(cnt), None, None, None, None, None) != 0:
        raise FeaHipError("feahip_host_rank_plan failed")
    npeer, nsend, nrecv = (int(v) for v in cnt)
    peers = np.zeros(max(npeer, 1), dtype=np.int32)
    soff, roff = np.zeros(npeer + 1, dtype=np.int32), np.zeros(npeer + 1, dtype=np.int32)
    sidx, ridx = np.zeros(max(nsend, 1), dtype=np.int32), np.zeros(max(nrecv, 1), dtype=np.int32)
    if lib.feahip_host_rank_plan(*args, _i(cnt), _i(peers), _i(soff), _i(roff), _i(sidx), _i(ridx)) != 0:
        raise FeaHipError("feahip_host_rank_plan failed")
    return {"peers": [int(p) for p in peers[:npeer]],
            "send": [sidx[soff[k]:soff[k + 1]].copy() for k in range(npeer)],
            "recv": [ridx[roff[k]:roff[k + 1]].copy() for k in range(npeer)]}


class FeaGroup:
    """n contexts of one mesh sharded by rows and driven from this process
    (feahip_group_* entries)."""

    def __init__(self, deck, n, device=0, rank_contexts=False):
        """rank_contexts: every rank holds only its sub-mesh (RankSolver) instead of the whole mesh with a row shard."""
        self.deck, self.n = deck, n
        self.rank_contexts = rank_contexts
        if rank_contexts:
            self.ranks = [RankSolver(deck, r, n, device=device) for r in range(n)]
        else:
            self.ranks = [FeaSolver(deck, device=device) for _ in range(n)]
        self._lib = load_library()
        self._arr = (C.c_void_p * n)(*[r._ctx for r in self.ranks])
        self._chk(self._lib.feahip_group_init(self._arr, n))
        self.rows = [r.owned_rows() for r in self.ranks]        # library ids (local ids for rank contexts)
        if rank_contexts:
            self.nodes = [r.node_global[:r.n_own].astype(np.int64) for r in self.ranks]
        else:
            self.nodes = [r.owned_nodes() for r in self.ranks]  # the caller's ids of every rank's nodes

    def _chk(self, rc):
        if rc != 0:
            msgs = [self._lib.feahip_last_error(r._ctx).decode() for r in self.ranks]
            raise FeaHipError(f"libfeahip group error {rc}: {msgs}")

    def each(self, name, *args):
        return [getattr(r, name)(*args) for r in self.ranks]

    def solve_slae(self, solver_type, tolerance, max_iterations):
        it, res = C.c_int(0), C.c_double(0)
        self._chk(self._lib.feahip_group_solve_slae(self._arr, self.n, solver_type, tolerance, max_iterations,
                                                    C.byref(it), C.byref(res)))
        return it.value, res.value

    def energy(self):
        t = C.c_double(0)
        self._chk(self._lib.feahip_group_energy(self._arr, self.n, C.byref(t)))
        return t.value

    def update_nodes_with_solution(self):
        self._chk(self._lib.feahip_group_update_nodes_with_solution(self._arr, self.n))

    def solve(self, load_increments, max_newton, modified_newton, desired_tolerance, solver_type,
              solver_tolerance=1e-14, solver_max_iter=20000):
        cap = load_increments * max_newton
        tol_log = np.zeros(cap)
        its = np.zeros(load_increments, dtype=np.int32)
        done = C.c_int(0)
        self._chk(self._lib.feahip_group_solve(self._arr, self.n, load_increments, max_newton, int(modified_newton),
                                               desired_tolerance, solver_type, solver_tolerance, solver_max_iter,
                                               _d(tol_log), cap, _i(its), C.byref(done)))
        return done.value, its, tol_log[:int(its.sum())]

    def gather(self, name):
        """Owned rows of a per-node ([N][3]) or per-dof ([3N]) getter, stitched together."""
        parts = self.each(name)
        if self.rank_contexts:                                  # local arrays: owned rows are the first n_own
            first = parts[0]
            out = np.zeros((len(self.deck.nodes), 3)) if first.ndim == 2 else np.zeros(3 * len(self.deck.nodes))
            for r, p in zip(self.ranks, parts):
                nd = r.node_global[:r.n_own].astype(np.int64)
                if out.ndim == 2:
                    out[nd] = p[:r.n_own]
                else:
                    out.reshape(-1, 3)[nd] = p.reshape(-1, 3)[:r.n_own]
            return out
        out = parts[0].copy()
        for nd, p in zip(self.nodes, parts):
            if out.ndim == 2:
                out[nd] = p[nd]
            else:
                d = (3 * nd[:, None] + np.arange(3)[None, :]).ravel()
                out[d] = p[d]
        return out

    def close(self):
        for r in self.ranks:
            r.close()
