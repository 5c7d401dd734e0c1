"""Parity of the HIP path with the oracle, through the C ABI, on an MI355X.

Bars: integer / index work (pattern, connectivity-derived offsets) is
bit-exact; floating-point results agree to the tolerance written at each
assert -- K_e and f to 1e-12 of their scale (re-association and FMA
contraction only), displacements to 1e-10 relative (BASELINE.json).
"""
import gzip
import os
import shutil

import numpy as np
import pytest
import scipy.sparse as sp

import feahip
import mesh
from oracle_binding import OracleSolver

pytestmark = pytest.mark.gpu

K_TOL = 1e-12     # relative to max|K| (resp. max|f|)
U_TOL = 1e-10     # BASELINE.json: displacements within 1e-10 relative


def rel(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def make_pair(deck, x=None):
    s, o = feahip.FeaSolver(deck), OracleSolver(deck)
    if x is not None:
        s.set_nodes(x)
        o.set_nodes(x)
    return s, o


def check_assembly(s, o, strategies=(feahip.ASM_ROWOWNER, feahip.ASM_ATOMIC)):
    if s.npe == 4 and s.G == 1:
        strategies = tuple(strategies) + (feahip.ASM_STAGED, feahip.ASM_GATHER)
    if s.npe == 10:
        strategies = tuple(strategies) + (feahip.ASM_SHARED, feahip.ASM_GATHER)
    o.update_state()
    o.create_stiffness()
    o.create_residual_forces()
    for strat in strategies:
        s.set_assembly(strat)
        s.create_stiffness_and_residual()
        off, idx, val = s.matrix_yale()
        assert np.array_equal(off, o.offsets())       # bit-exact indexing
        assert np.array_equal(idx, o.indexes())
        assert rel(val, o.values()) < K_TOL, strat
        assert rel(s.forces(), o.forces()) < K_TOL, strat
        # the separate entry points give the same bits as the fused one
        s.create_stiffness()
        assert rel(s.matrix_yale()[2], val) < 1e-15
        s.create_residual_forces()
        assert rel(s.forces(), o.forces()) < K_TOL
    s.set_assembly(feahip.ASM_AUTO)


@pytest.mark.parametrize("model", [feahip.MODEL_COMPRESSIBLE_NEOHOOKEAN, feahip.MODEL_A5])
@pytest.mark.parametrize("dims", [(1, 1, 1), (3, 5, 2)])
def test_tet4_assembly(model, dims):
    deck = mesh.bar_deck(dims=dims, model=model)
    s, o = make_pair(deck, mesh.deformed_state(deck.nodes, k1=1.08, wiggle=5e-3))
    check_assembly(s, o)
    assert rel(s.graddefs(), o.graddefs()) < 1e-13
    assert rel(s.stresses(), o.stresses()) < 1e-12
    g, d = s.shape_gradients()                          # a3: shape_gradients[e][g] (fea_solver.h:200-205)
    assert rel(g, o.grads()) < 1e-13 and rel(d, o.detj()) < 1e-13
    s.close()


@pytest.mark.parametrize("name", ["neohook_brick", "a5_brick"])
def test_tet10_reference_deck_assembly(decks_dir, name):
    deck = feahip.Deck.load(os.path.join(decks_dir, name + ".sexp"))
    s, o = make_pair(deck, mesh.deformed_state(deck.nodes, k1=1.05, wiggle=2e-3))
    check_assembly(s, o)
    assert rel(s.graddefs(), o.graddefs()) < 1e-12
    assert rel(s.stresses(), o.stresses()) < 1e-11
    g, d = s.shape_gradients()
    assert rel(g, o.grads()) < 1e-12 and rel(d, o.detj()) < 1e-12
    s.close()


def test_tet10_27_point_rule():
    """BASELINE.json config 5's rule (27 Gauss points per quadratic tet)."""
    deck = mesh.bar_deck(dims=(2, 3, 2), quadratic=True, gauss=27)
    s, o = make_pair(deck, mesh.deformed_state(deck.nodes, k1=1.06, wiggle=3e-3))
    check_assembly(s, o, strategies=(feahip.ASM_ROWOWNER, feahip.ASM_ATOMIC))
    assert rel(s.stresses(), o.stresses()) < 1e-11
    s.close()


def test_tet10_four_point_rule(decks_dir):
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    deck.gauss_nodes_count = 4
    s, o = make_pair(deck, mesh.deformed_state(deck.nodes, k1=1.03))
    check_assembly(s, o, strategies=(feahip.ASM_ROWOWNER,))
    s.close()


def test_gather_assembly_is_bitwise_reproducible():
    deck = mesh.bar_deck(dims=(7, 13, 5))
    s = feahip.FeaSolver(deck)
    s.set_nodes(mesh.deformed_state(deck.nodes))
    s.set_assembly(feahip.ASM_GATHER)
    s.create_stiffness_and_residual()
    v1, f1 = s.matrix_yale()[2], s.forces()
    s.create_stiffness_and_residual()
    assert np.array_equal(v1, s.matrix_yale()[2]) and np.array_equal(f1, s.forces())
    s.create_stiffness()                                  # K alone: same bits
    assert np.array_equal(v1, s.matrix_yale()[2])
    s.create_residual_forces()                            # f alone: a different record, same sums to rounding
    assert rel(s.forces(), f1) < 1e-14
    s.close()


def test_rowowner_is_deterministic():
    deck = mesh.bar_deck(dims=(4, 8, 4))
    s = feahip.FeaSolver(deck)
    s.set_nodes(mesh.deformed_state(deck.nodes))
    s.set_assembly(feahip.ASM_ROWOWNER)
    s.create_stiffness_and_residual()
    v1, f1 = s.matrix_yale()[2], s.forces()
    s.create_stiffness_and_residual()
    v2, f2 = s.matrix_yale()[2], s.forces()
    # LDS adds of one wave may commute between runs; everything else is fixed
    assert rel(v1, v2) < 1e-15 and rel(f1, f2) < 1e-15
    s.close()


def test_undeformed_state_is_stress_free_and_bump_moves_face():
    deck = mesh.bar_deck(dims=(2, 6, 2), dy=0.05)
    s, o = make_pair(deck)
    s.create_residual_forces()
    assert np.abs(s.forces()).max() < 1e-12
    s.update_nodes_with_bc(1.0)
    o.update_nodes_with_bc(1.0)
    assert np.array_equal(s.nodes(), o.nodes())
    assert s.update_state() == 0
    s.close()


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_prescribed_bc_application(lam, decks_dir):
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick_analytical.sexp"))
    s, o = make_pair(deck, mesh.deformed_state(deck.nodes, k1=1.02))
    o.update_state(); o.create_stiffness(); o.create_residual_forces()
    s.create_stiffness_and_residual()
    o.apply_prescribed_bc(lam)
    s.apply_prescribed_bc(lam)
    off, idx, val = s.matrix_yale()
    cmask = np.zeros(s.ndof, dtype=bool)
    for n, t in zip(deck.presc_node, deck.presc_type):
        for j in range(3):
            if t & (1 << j):
                cmask[3 * n + j] = True
    rows = np.repeat(np.arange(s.ndof), np.diff(off))
    cancelled = (cmask[rows] | cmask[idx]) & (rows != idx)
    assert cancelled.sum() > 0 and np.all(val[cancelled] == 0)       # rows and columns zeroed exactly
    assert np.all(o.values()[cancelled] == 0)
    assert np.all(val[(rows == idx) & cmask[rows]] != 0)               # diagonal kept (fea_solver.c:1255)
    assert rel(val, o.values()) < K_TOL
    assert rel(s.forces(), o.forces()) < 1e-11     # f is a sum of cancelling element forces
    s.close()


def test_spmv_and_energy():
    deck = mesh.bar_deck(dims=(3, 7, 3))
    s, o = make_pair(deck, mesh.deformed_state(deck.nodes))
    o.update_state(); o.create_stiffness()
    s.create_stiffness()
    rng = np.random.default_rng(3)
    x = rng.normal(size=s.ndof)
    assert rel(s.spmv(x), o.spmv(x)) < 1e-13
    s.close()


@pytest.mark.parametrize("solver", [feahip.CG, feahip.PCG_ILU, feahip.CHOLESKY])
def test_linear_solve_matches_direct_solver(solver):
    deck = mesh.bar_deck(dims=(3, 8, 3))
    s, o = make_pair(deck)
    for obj in (s, o):
        obj.update_nodes_with_bc(1.0)
    o.update_state(); o.create_stiffness(); o.create_residual_forces(); o.apply_prescribed_bc(0.0)
    s.create_stiffness_and_residual(); s.apply_prescribed_bc(0.0)
    o.solve_slae(feahip.CHOLESKY)
    it, res = s.solve_slae(solver, 1e-15, 20000)
    assert it > 0 and res < 1e-14
    assert rel(s.solution(), o.solution()) < U_TOL
    assert s.energy() == pytest.approx(o.energy(), rel=1e-11)
    cd = np.concatenate([[3 * n + j for j in range(3) if t & (1 << j)] for n, t in zip(deck.presc_node, deck.presc_type)])
    assert np.all(s.solution()[cd] == 0)              # increments at prescribed dofs are exactly 0
    s.close()


def test_multigrid_preconditioner_same_solution_fewer_iterations():
    """feahip_set_preconditioner(1): PCG with the aggregation-multigrid W-cycle
    (rigid-body coarse space).  Same linear system, same stop test: the
    solution is the direct solver's within 1e-10 and the iteration count drops
    well below block-Jacobi's; a whole Newton solve lands on the same nodes."""
    deck = mesh.bar_deck(dims=(6, 36, 6))
    s, o = make_pair(deck)
    for obj in (s, o):
        obj.update_nodes_with_bc(1.0)
    o.update_state(); o.create_stiffness(); o.create_residual_forces(); o.apply_prescribed_bc(0.0)
    s.create_stiffness_and_residual(); s.apply_prescribed_bc(0.0)
    o.solve_slae(feahip.CHOLESKY)
    it_bj, _ = s.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
    s.set_preconditioner(1)
    it_mg, res = s.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
    assert res < 1e-14 and 0 < it_mg < 0.6 * it_bj
    assert rel(s.solution(), o.solution()) < U_TOL
    cd = np.concatenate([[3 * n + j for j in range(3) if t & (1 << j)] for n, t in zip(deck.presc_node, deck.presc_type)])
    assert np.all(s.solution()[cd] == 0)
    # plain CG ignores the setting (the reference's CG is unpreconditioned)
    it_cg, _ = s.solve_slae(feahip.CG, 1e-15, 20000)
    assert it_cg > it_bj
    s.close()
    deck = mesh.bar_deck(dims=(6, 36, 6), load_increments_count=1, max_newton_count=30)
    a, b = feahip.FeaSolver(deck), feahip.FeaSolver(deck)
    b.set_preconditioner(1)
    ra = a.solve(solver_type=feahip.PCG_ILU, solver_tolerance=1e-15)
    rb = b.solve(solver_type=feahip.PCG_ILU, solver_tolerance=1e-15)
    assert ra[0] == rb[0] == 1 and list(ra[1]) == list(rb[1])
    assert rel(b.nodes() - deck.nodes, a.nodes() - deck.nodes) < U_TOL
    with pytest.raises(feahip.FeaHipError):
        a.set_preconditioner(7)
    a.close(); b.close()
    tiny = feahip.FeaSolver(mesh.bar_deck(dims=(2, 2, 2)))
    with pytest.raises(feahip.FeaHipError, match="multigrid"):     # nothing to coarsen: refused, not silently Jacobi
        tiny.set_preconditioner(1)
    tiny.close()


def test_multigrid_variants_are_the_same_preconditioner(monkeypatch):
    """The measurement knobs of the cycle change how it is computed, not what: the small levels in one launch
    (k_amg_tail) or kernel by kernel, the post-smoothing sweep fused with its product or in two launches, the smoother's
    level-0 matrix in bfloat16 / float / double, a shallower hierarchy, single corrections below the finest level.
    Every variant must solve the same system to the same answer; the one-launch tail repeats the kernel-by-kernel
    iteration count exactly (same steps, same order), the matrix precision may move it by a few iterations."""
    deck = mesh.bar_deck(dims=(8, 48, 8))
    s, o = make_pair(deck)
    for obj in (s, o):
        obj.update_nodes_with_bc(1.0)
    o.update_state(); o.create_stiffness(); o.create_residual_forces(); o.apply_prescribed_bc(0.0)
    o.solve_slae(feahip.CHOLESKY)
    s.close()
    its = {}
    for name, env in (("default", {}), ("no_tail", {"FEAHIP_AMG_TAIL": "0"}), ("fused_post", {"FEAHIP_AMG_FUSED_POST": "1"}),
                      ("v_below_1", {"FEAHIP_AMG_GAMMA_UNTIL": "1"}),
                      ("tail_csr", {"FEAHIP_AMG_TAIL_ELL": "0", "FEAHIP_AMG_TAIL_BLOB": "0", "FEAHIP_AMG_TAIL_COP": "0"}),
                      ("f32", {"FEAHIP_AMG_FINE_BITS": "32"}),
                      ("f64", {"FEAHIP_AMG_FINE_BITS": "64"}), ("shallow", {"FEAHIP_AMG_COARSEST": "1500", "FEAHIP_AMG_SWEEPS": "12"})):
        for k in ("FEAHIP_AMG_TAIL", "FEAHIP_AMG_FINE_BITS", "FEAHIP_AMG_COARSEST", "FEAHIP_AMG_SWEEPS", "FEAHIP_AMG_FUSED_POST",
                  "FEAHIP_AMG_GAMMA_UNTIL", "FEAHIP_AMG_TAIL_ELL", "FEAHIP_AMG_TAIL_BLOB", "FEAHIP_AMG_TAIL_COP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        t = feahip.FeaSolver(deck)
        t.update_nodes_with_bc(1.0)
        t.create_stiffness_and_residual(); t.apply_prescribed_bc(0.0)
        t.set_preconditioner(1)
        its[name], res = t.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
        assert res < 1e-14, name
        assert rel(t.solution(), o.solution()) < U_TOL, name
        t.close()
    # the one-launch tail against the same steps kernel by kernel: the same operator (its coarsest level as one dense
    # product: a different association), iteration counts within one
    assert abs(its["default"] - its["no_tail"]) <= 1 and its["no_tail"] == its["tail_csr"], its
    assert abs(its["fused_post"] - its["default"]) <= 1, its          # same sums, the update in the same launch
    assert abs(its["f32"] - its["default"]) <= 6 and abs(its["f64"] - its["default"]) <= 6, its


@pytest.mark.parametrize("case", ["cylinder_tet10_a5", "bar_tet10_neohookean"])
def test_multigrid_on_quadratic_elements_and_curved_geometry(case):
    """The hierarchy is built from the node graph and the node positions only:
    10-node elements (mid-side nodes are nodes like any other), the periodic
    cylinder, model A5.  Same solution as block-Jacobi PCG, fewer iterations."""
    deck = (mesh.cylinder_deck(5, 30, 6, quadratic=True) if case == "cylinder_tet10_a5"
            else mesh.bar_deck(dims=(5, 30, 5), quadratic=True))
    s = feahip.FeaSolver(deck)
    s.update_nodes_with_bc(1.0); s.create_stiffness_and_residual(); s.apply_prescribed_bc(0.0)
    it0, _ = s.solve_slae(feahip.PCG_ILU, 1e-14, 50000)
    u0 = s.solution()
    s.set_preconditioner(1)
    it1, res = s.solve_slae(feahip.PCG_ILU, 1e-14, 50000)
    assert res < 1e-13 and 0 < it1 < 0.8 * it0
    assert rel(s.solution(), u0) < U_TOL
    s.close()


def test_zero_rhs_solves_to_zero():
    deck = mesh.bar_deck(dims=(2, 2, 2))
    s = feahip.FeaSolver(deck)
    s.create_stiffness_and_residual()
    s.apply_prescribed_bc(0.0)
    it, res = s.solve_slae(feahip.PCG_ILU, 1e-14, 100)
    assert it == 0 and np.all(s.solution() == 0)
    s.close()


def run_newton_both(deck, steps, modified, solver=feahip.CHOLESKY):
    s, o = make_pair(deck)
    od, oits, otol = o.solve(steps, deck.max_newton_count, modified, deck.desired_tolerance, feahip.CHOLESKY)
    sd, sits, stol = s.solve(load_increments=steps, modified_newton=modified, solver_type=solver)
    return s, o, (sd, sits, stol), (od, oits, otol)


def test_newton_on_reference_deck_matches_oracle(decks_dir):
    """config 1: the shipped clamped deck with its own settings (modified
    Newton, tolerance 1e-6 on <u,f>, direct solver).  Same iteration
    sequence, same <u,f> per iteration, displacements within 1e-10."""
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    s, o, (sd, sits, stol), (od, oits, otol) = run_newton_both(deck, 2, True)
    assert sd == od == 2 and list(sits[:2]) == list(oits[:2]) == [13, 12]
    assert np.abs(stol - otol).max() < 1e-10 * np.abs(otol).max()
    du_s, du_o = s.nodes() - deck.nodes, o.nodes() - deck.nodes
    assert rel(du_s, du_o) < U_TOL
    assert rel(s.stresses(), o.stresses()) < 1e-9
    s.close()


def test_newton_a5_deck_matches_oracle(decks_dir):
    deck = feahip.Deck.load(os.path.join(decks_dir, "a5_brick.sexp"))
    s, o, (sd, sits, stol), (od, oits, otol) = run_newton_both(deck, 1, True)
    assert sd == od == 1 and list(sits[:1]) == list(oits[:1]) == [7]
    assert rel(s.nodes() - deck.nodes, o.nodes() - deck.nodes) < U_TOL
    s.close()


@pytest.mark.parametrize("quadratic", [False, True])
def test_lame_cylinder_a5(quadratic):
    """BASELINE configs[3]: mapped, theta-periodic Kuhn block of the Lame
    cylinder, model A5, mixed boundary types (3, 4, 7).  Assembly in the
    bumped state and one load increment of Newton against the oracle; the
    small-strain Lame answer itself is pinned on the oracle in
    test_oracle_closed_form.py."""
    deck = mesh.cylinder_deck(2, 12, 2, quadratic=quadratic, zlo=0.0, zhi=1.0, du=0.01, load_increments_count=1,
                              max_newton_count=20, desired_tolerance=1e-16, modified_newton=False)
    s, o = make_pair(deck)
    for obj in (s, o):
        obj.update_nodes_with_bc(1.0)
    check_assembly(s, o)
    s.close(); o.close()
    s, o, (sd, sits, stol), (od, oits, otol) = run_newton_both(deck, 1, False)
    assert sd == od == 1 and list(sits[:1]) == list(oits[:1])
    du_s, du_o = s.nodes() - deck.nodes, o.nodes() - deck.nodes
    assert rel(du_s, du_o) < U_TOL
    r = np.hypot(deck.nodes[:, 0], deck.nodes[:, 1])
    ur = (du_s[:, 0] * deck.nodes[:, 0] + du_s[:, 1] * deck.nodes[:, 1]) / r
    assert np.abs(ur[np.abs(r - 1.0) < 1e-9] - 0.01).max() < 1e-15          # prescribed exactly
    assert 0.5 * 0.01 < ur[np.abs(r - 2.0) < 1e-9].mean() < 0.8 * 0.01      # Lame: 2/3 at small strain
    s.close(); o.close()


def test_line_search_matches_the_prototype_restated_on_the_oracle():
    """feahip_set_line_search: golden-section search of the step length in
    [1/2, 1] for min |eta <u, R(x + eta u)>| (cartesian3d_large.m:85-119; the C
    solver parses `line-search :max` and leaves it unused).  The same loop
    restated with the oracle's primitives gives the same iteration count,
    the same <u,f> sequence and the same displacements."""
    LS = 5
    deck = mesh.bar_deck(dims=(3, 9, 3), dy=0.3, load_increments_count=1, max_newton_count=25,
                         desired_tolerance=1e-16, modified_newton=False)
    s = feahip.FeaSolver(deck)
    done, its, tol = s.solve(solver_type=feahip.CHOLESKY, line_search=LS)
    o = OracleSolver(deck)
    o.update_nodes_with_bc(1.0); o.update_state(); o.create_stiffness()
    tau, olog, it = (np.sqrt(5.0) - 1.0) / 2.0, [], 0
    while True:
        it += 1
        if it > 1:
            o.create_stiffness()
        o.create_residual_forces(); o.apply_prescribed_bc(0.0); o.solve_slae(feahip.CHOLESKY)
        t = o.energy(); olog.append(t)
        base, u = o.nodes().copy(), o.solution().copy()
        a, b, eta = 0.5, 1.0, 1.0
        for _ in range(LS):
            x1, x2 = b - tau * (b - a), a + tau * (b - a)
            f = []
            for xk in (x1, x2):
                o.set_nodes(base + xk * u.reshape(-1, 3)); o.update_state(); o.create_residual_forces()
                f.append(abs(xk * float(o.forces() @ u)))
            if f[0] > f[1]:
                a = x1
            else:
                b = x2
            if abs(t) < f[0] and abs(t) < f[1]:
                eta = 1.0
                break
            eta = 0.5 * (x1 + x2)
        o.set_nodes(base + eta * u.reshape(-1, 3)); o.update_state()
        if not (abs(t) > deck.desired_tolerance and it < deck.max_newton_count):
            break
    assert done == 1 and its[0] == it
    assert np.abs(tol - np.array(olog)).max() < 1e-9 * np.abs(olog).max()
    assert rel(s.nodes() - deck.nodes, o.nodes() - deck.nodes) < U_TOL
    # and it is a different path from the plain Newton loop on this step
    p = feahip.FeaSolver(deck)
    pd, pits, ptol = p.solve(solver_type=feahip.CHOLESKY)
    assert pd != 1 or pits[0] != its[0] or np.abs(ptol[1] - tol[1]) > 1e-6 * abs(tol[1])
    with pytest.raises(feahip.FeaHipError):
        p.set_line_search(-1)
    s.close(); p.close(); o.close()


def test_full_newton_tet4_patch_test_closed_form():
    """Rotation-free uniaxial recipe on linear tets: the Neo-Hookean closed
    form of exact-solutions/uniaxial/uniaxial_neohookean_bonet.m is the exact
    FE answer, at any mesh size."""
    from test_oracle_closed_form import nh_closed_form
    deck = mesh.bar_deck(dims=(4, 24, 4), recipe="uniaxial", dy=0.05, max_newton_count=8, desired_tolerance=1e-22,
                         modified_newton=False, load_increments_count=2)
    s = feahip.FeaSolver(deck)
    done, its, tol = s.solve(solver_type=feahip.PCG_ILU, solver_tolerance=1e-15)
    k1 = 1 + 2 * 0.05 / 6
    k2, syy = nh_closed_form(k1)
    S = s.stresses()
    assert np.abs(S[:, 0, 1, 1] - syy).max() < 1e-9
    A = deck.nodes.min(axis=0)
    expect = A + (deck.nodes - A) * np.array([k2, k1, k2])
    assert np.abs(s.nodes() - expect).max() < 1e-11
    s.close()


def test_host_driver_runs_the_deck(decks_dir, tmp_path):
    """fea_solve() in libfeahost.so (the C mirror of solve()) drives the same
    ABI and logs the reference's convergence lines."""
    import ctypes as C
    h = feahip.load_host_library()
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    deck.load_increments_count = 1
    fd = deck.to_struct()
    ctx = C.c_void_p()
    err = C.create_string_buffer(256)
    assert h.fea_deck_create_solver(C.byref(fd), 0, C.byref(ctx), err, 256) == 0, err.value
    xs = np.zeros((1, len(deck.nodes), 3))
    steps = h.fea_solve(C.byref(fd), ctx, None, xs.ctypes.data_as(C.POINTER(C.c_double)), 1)
    assert steps == 1
    o = OracleSolver(deck)
    o.solve(1, deck.max_newton_count, True, deck.desired_tolerance, feahip.CHOLESKY)
    assert rel(xs[0] - deck.nodes, o.nodes() - deck.nodes) < U_TOL
    feahip.load_library().feahip_destroy(ctx)


def test_brick_fine_one_assembly(decks_dir, tmp_path):
    """The largest shipped deck (22 934 TET10) with its BC ids shifted to
    0-based (build-side correction): one assembly + BC."""
    p = tmp_path / "brick_fine.sexp"
    with gzip.open(os.path.join(decks_dir, "brick_fine.sexp.gz"), "rb") as src, open(p, "wb") as dst:
        shutil.copyfileobj(src, dst)
    deck = feahip.Deck.load(str(p))
    deck.presc_node = (deck.presc_node - 1).astype(np.int32)
    s = feahip.FeaSolver(deck)
    s.update_nodes_with_bc(1.0)
    s.create_stiffness_and_residual()
    assert s.update_state() == 0                         # no inverted Gauss point with corrected ids
    assert s.assembly_in_use() == feahip.ASM_GATHER      # the TetGen deck, numbered by the library: state + gather kernels
    assert s.assembly_stats()["evals_per_element"] < 12.0
    off, idx, val = s.matrix_yale()
    assert len(val) == 8300196                           # scalar nnz of the deck
    # spot-check the block rows of 40 nodes against the sum of the oracle's element matrices of the elements around
    # them (fea_solver.c:964-969: the rows an element adds to are 3 conn[e][a] + i)
    o = OracleSolver(deck)
    o.update_nodes_with_bc(1.0)
    o.update_state()
    import scipy.sparse as sp
    K = sp.csr_matrix((val, idx, off), shape=(s.ndof, s.ndof))
    rng = np.random.default_rng(5)
    picked = rng.choice(len(deck.nodes), 40, replace=False)
    kscale = np.abs(val).max()
    for a in picked:
        es, las = np.nonzero(deck.elements == a)
        want = np.zeros((3, s.ndof))
        for e, la in zip(es, las):
            kc, ks = o.element_stiffness(int(e))
            cols = (3 * deck.elements[e][:, None] + np.arange(3)[None, :]).ravel()
            np.add.at(want, (np.arange(3)[:, None], cols[None, :]), (kc + ks)[3 * la:3 * la + 3, :])
        got = K[3 * a:3 * a + 3, :].toarray()
        assert np.abs(got - want).max() < 1e-12 * kscale, int(a)
    sym = abs(K - K.T).max() / abs(K).max()
    assert sym < 1e-12
    t = np.zeros(s.ndof); t[1::3] = 1.0
    assert np.abs(K @ t).max() < 1e-10 * abs(K).max()
    s.close()


def test_isolated_node_and_single_element_edge_cases():
    """Ragged inputs.  A node no element refers to: its row is just its
    diagonal block; pinned, it must not disturb anything -- the run equals
    the run without it (which the oracle checks), bit for bit in the
    iteration counts.  (The reference has no such deck: its sparse matrix has
    no entry to put the 1 of a prescribed dof on.)  And a mesh of one element."""
    base = mesh.bar_deck(dims=(2, 3, 2), load_increments_count=1, max_newton_count=20, desired_tolerance=1e-18)
    s0, o, (sd, sits, stol), (od, oits, otol) = run_newton_both(base, 1, True)
    assert sd == od == 1 and list(sits[:1]) == list(oits[:1])
    deck = mesh.bar_deck(dims=(2, 3, 2), load_increments_count=1, max_newton_count=20, desired_tolerance=1e-18)
    N = len(deck.nodes)
    deck.nodes = np.vstack([deck.nodes, [[5.0, 5.0, 5.0]]])
    deck.presc_node = np.append(deck.presc_node, N).astype(np.int32)
    deck.presc_type = np.append(deck.presc_type, 7).astype(np.int32)
    deck.presc_values = np.vstack([deck.presc_values, [[0.0, 0.0, 0.0]]])
    s = feahip.FeaSolver(deck)
    d1, its1, tol1 = s.solve(solver_type=feahip.CHOLESKY, load_increments=1, modified_newton=True)
    assert d1 == 1 and its1[0] == sits[0]
    assert rel(s.nodes()[:N] - base.nodes, o.nodes() - base.nodes) < U_TOL
    assert np.array_equal(s.nodes()[N], deck.nodes[N])
    off, idx, val = s.matrix_yale()
    assert off[-1] == o.offsets()[-1] + 9 and np.all(np.isfinite(val))     # one more 3x3 block: the diagonal of the lone node
    s.close(); s0.close(); o.close()
    one = mesh.bar_deck(dims=(1, 1, 1))
    keep = np.unique(one.elements[0])
    remap = -np.ones(len(one.nodes), dtype=np.int64); remap[keep] = np.arange(4)
    one.nodes = one.nodes[keep]
    one.elements = remap[one.elements[:1]].astype(np.int32)     # one tetrahedron
    sel = np.isin(one.presc_node, keep)
    one.presc_node = remap[one.presc_node[sel]].astype(np.int32)
    one.presc_type, one.presc_values = one.presc_type[sel], one.presc_values[sel]
    s, o = make_pair(one, mesh.deformed_state(one.nodes))
    check_assembly(s, o)
    s.close(); o.close()


def test_error_paths():
    deck = mesh.bar_deck(dims=(1, 1, 1))
    s = feahip.FeaSolver(deck)
    with pytest.raises(feahip.FeaHipError, match="before stash"):
        s.restore_stiffness()
    with pytest.raises(feahip.FeaHipError, match="solver type"):
        s.solve_slae(9, 1e-10, 10)
    with pytest.raises(feahip.FeaHipError, match="strategy"):
        s.set_assembly(9)
    s.set_assembly(feahip.ASM_SHARED)                   # a 10-node strategy on linear tets: refused at the launch
    with pytest.raises(feahip.FeaHipError, match="10-node"):
        s.create_stiffness()
    s.close()
    bad = mesh.bar_deck(dims=(1, 1, 1))
    bad.elements = bad.elements.copy()
    bad.elements[0, 0] = 99
    with pytest.raises(feahip.FeaHipError, match="outside"):
        feahip.FeaSolver(bad)
    bad = mesh.bar_deck(dims=(1, 1, 1))
    bad.presc_node = bad.presc_node.copy()
    bad.presc_node[0] = -4
    with pytest.raises(feahip.FeaHipError, match="out of range"):
        feahip.FeaSolver(bad)


def test_inverted_elements_are_reported():
    deck = mesh.bar_deck(dims=(2, 2, 2))
    x = deck.nodes.copy()
    x[deck.elements[0, 1]] = x[deck.elements[0, 0]] - (x[deck.elements[0, 1]] - x[deck.elements[0, 0]])
    s = feahip.FeaSolver(deck)
    s.set_nodes(x)
    s.create_stiffness()
    assert s.update_state() > 0
    s.close()


def test_feasolver_hip_command_line(decks_dir, tmp_path):
    """`feasolver_hip deck.sexp` = the reference's command line: loads the deck,
    logs the convergence lines of fea_solver.c:212-224, writes <base>.msh."""
    import re
    import subprocess
    src = open(os.path.join(decks_dir, "neohook_brick.sexp")).read()
    deckfile = tmp_path / "two_steps.sexp"
    deckfile.write_text(src.replace(":load-increments-count 120", ":load-increments-count 2"))
    exe = os.path.join(os.path.dirname(feahip.LIB_PATH), "feasolver_hip")
    res = subprocess.run([exe, str(deckfile)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    out = res.stdout
    assert out.count("Newton iteration") == 25 and "Load increment 2 finished" in out      # 13 + 12
    tol = [float(v) for v in re.findall(r"Tolerance <X,R> = (\S+)", out)]
    assert tol[0] == pytest.approx(5.392131, rel=1e-6) and abs(tol[12]) < 1e-6 <= abs(tol[11])
    msh = (tmp_path / "two_steps.msh").read_text().splitlines()
    assert msh.count("$NodeData") == 3 and msh.count("$ElementData") == 3
    # displacement of a top-face node after two increments: 2 x 0.05 along y
    deck = feahip.Deck.load(str(deckfile))
    top = int(deck.presc_node[deck.presc_values[:, 1] != 0][0])
    last = [i for i, l in enumerate(msh) if l == "$NodeData"][-1]
    row = msh[last + 9 + top].split()
    assert int(row[0]) == top + 1 and float(row[2]) == pytest.approx(0.1, abs=1e-6)
    # the one extra option: the same run with the multigrid preconditioner (737 nodes: two levels) -- same Newton path
    res2 = subprocess.run([exe, str(deckfile), "--multigrid"], capture_output=True, text=True, timeout=300)
    assert res2.returncode == 0, res2.stderr
    assert res2.stdout.count("Newton iteration") == 25
    tol2 = [float(v) for v in re.findall(r"Tolerance <X,R> = (\S+)", res2.stdout)]
    assert tol2[0] == pytest.approx(tol[0], rel=1e-9) and abs(tol2[12]) < 1e-6 <= abs(tol2[11])
    # on a 117-node bar there is nothing to coarsen -- asked for and unavailable is an error (exit code 1, message),
    # never a quiet run with another preconditioner
    tiny = tmp_path / "tiny.sexp"
    mesh.bar_deck(n=2).save(str(tiny))
    res3 = subprocess.run([exe, str(tiny), "--multigrid"], capture_output=True, text=True, timeout=300)
    assert res3.returncode == 1 and "--multigrid" in res3.stderr and "multigrid" in res3.stderr
    assert res3.stdout.count("Newton iteration") == 0                  # no quiet run with another preconditioner


def test_node_with_more_neighbours_than_the_spmv_tile():
    """A fan of 140 tetrahedra pairs around one node: its block row has 143 blocks, more than the 128-block LDS
    tile of the SpMV / PCG kernels (and of the row-owner assembly).  Assembly must still be right, and product and solve walk the long row with the lanes striding over its blocks: the
    reference's solvers have no row-length limit (fea_solver.c:300-321), so neither has this path."""
    m = 140
    ang = 2 * np.pi * np.arange(m) / m
    ring = np.stack([np.cos(ang), np.sin(ang), np.zeros(m)], axis=1)
    nodes = np.vstack([[0.0, 0.0, 0.0], ring, [0.0, 0.0, 0.7], [0.0, 0.0, -0.7]])
    top, bot = m + 1, m + 2
    el = []
    for i in range(m):
        a, b = 1 + i, 1 + (i + 1) % m
        el.append([0, a, b, top])
        el.append([0, b, a, bot])
    # both apexes clamped, one ring node held in y: no rigid motion left (the rotation about the apex axis would be)
    deck = feahip.Deck(nodes=nodes, elements=np.array(el, dtype=np.int32), ele_type=feahip.TETRAHEDRA4, gauss_nodes_count=1,
                       presc_node=[top, bot, 1], presc_type=[7, 7, 2], presc_values=np.zeros((3, 3)))
    x = nodes * np.array([1.02, 0.99, 1.05]) + 1e-3 * np.sin(3 * nodes[:, [1, 2, 0]])
    s, o = make_pair(deck, x)
    o.update_state(); o.create_stiffness(); o.create_residual_forces()
    s.create_stiffness_and_residual()
    assert s.assembly_in_use() == feahip.ASM_GATHER            # (its 142 off-diagonal blocks fit the gather chunk's 768 block threads)
    off, idx, val = s.matrix_yale()
    assert np.array_equal(off, o.offsets()) and np.array_equal(idx, o.indexes())
    assert off[3] - off[0] == 3 * 3 * (m + 3)                  # the centre's three rows: 143 blocks
    assert rel(val, o.values()) < K_TOL and rel(s.forces(), o.forces()) < K_TOL
    K = sp.csr_matrix((val, idx, off), shape=(s.ndof, s.ndof))
    xs = np.random.default_rng(3).normal(size=s.ndof)
    assert rel(s.spmv(xs), K @ xs) < 1e-13                     # the 143-block row included
    s.apply_prescribed_bc(0.0); o.apply_prescribed_bc(0.0)
    o.solve_slae(feahip.CHOLESKY)
    for solver in (feahip.PCG_ILU, feahip.CG):
        it, res = s.solve_slae(solver, 1e-15, 5000)
        assert res < 1e-13
        assert rel(s.solution(), o.solution()) < 1e-10
    assert s.energy() == pytest.approx(o.energy(), rel=1e-10)
    s.close(); o.close()


def test_tet10_hub_node_beyond_every_chunk_limit():
    """A quadratic fan: 140 ten-node tetrahedra around one node (more than the 127 element records of a gather chunk,
    its block row longer than the K tile).  The gather and shared-state maps do not build for this mesh; AUTO must end
    at a kernel that assembles it correctly, and an explicit GATHER request must be refused, not replaced."""
    m = 70
    ang = 2 * np.pi * np.arange(m) / m
    ring = np.stack([np.cos(ang), np.sin(ang), np.zeros(m)], axis=1)
    corners = np.vstack([[0.0, 0.0, 0.0], ring, [0.0, 0.0, 0.7], [0.0, 0.0, -0.7]])
    top, bot = m + 1, m + 2
    el4 = []
    for i in range(m):
        a, b = 1 + i, 1 + (i + 1) % m
        el4.append([0, a, b, top])
        el4.append([0, b, a, bot])
    nodes = [tuple(c) for c in corners]
    mid = {}
    el = []
    for e in el4:
        row = list(e)
        for (i, j) in mesh._EDGES:
            key = (min(e[i], e[j]), max(e[i], e[j]))
            if key not in mid:
                mid[key] = len(nodes)
                nodes.append(tuple(0.5 * (corners[key[0]] + corners[key[1]])))
            row.append(mid[key])
        el.append(row)
    nodes = np.array(nodes)
    deck = feahip.Deck(nodes=nodes, elements=np.array(el, dtype=np.int32), ele_type=feahip.TETRAHEDRA10, gauss_nodes_count=5,
                       presc_node=[top, bot, 1], presc_type=[7, 7, 2], presc_values=np.zeros((3, 3)))
    x = nodes * np.array([1.02, 0.99, 1.05]) + 1e-3 * np.sin(3 * nodes[:, [1, 2, 0]])
    s, o = make_pair(deck, x)
    o.update_state(); o.create_stiffness(); o.create_residual_forces()
    s.create_stiffness_and_residual()
    assert s.assembly_in_use() not in (feahip.ASM_GATHER, feahip.ASM_SHARED)
    off, idx, val = s.matrix_yale()
    assert np.array_equal(off, o.offsets()) and np.array_equal(idx, o.indexes())
    assert rel(val, o.values()) < K_TOL and rel(s.forces(), o.forces()) < K_TOL
    s.create_residual_forces()
    assert rel(s.forces(), o.forces()) < K_TOL
    s.set_assembly(feahip.ASM_GATHER)
    with pytest.raises(feahip.FeaHipError, match="gather assembly"):
        s.create_stiffness_and_residual()
    # the hub's block row (about 430 blocks: itself, the ring, the apexes and every mid-side node of its 140 elements)
    # through the product and the solve: displacement increment of the first Newton step against the oracle's direct solve
    s.set_assembly(feahip.ASM_AUTO)
    s.create_stiffness_and_residual()
    assert np.diff(off)[0] // 3 > 128
    s.apply_prescribed_bc(0.0); o.apply_prescribed_bc(0.0)
    o.solve_slae(feahip.CHOLESKY)
    it, res = s.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
    assert res < 1e-13
    assert rel(s.solution(), o.solution()) < 1e-10
    s.close(); o.close()
