"""Structural invariants of the restated element loops (SURVEY.md 8c iii):
checkable without any reference run, they catch index / sign / order slips."""
import os

import numpy as np
import pytest

import feahip
import mesh
import oracle_binding as ob
from oracle_binding import OracleSolver


@pytest.mark.parametrize("kind,G", [(ob.TET10, 4), (ob.TET10, 5), (ob.TET10, 27), (ob.TET4, 1)])
def test_gauss_tables(kind, G):
    w, forms, dforms = ob.elem_table(kind, G)
    assert w.sum() == pytest.approx(1.0 / 6.0, abs=1e-16)          # fea_solver.c:32-54
    assert np.abs(forms.sum(axis=1) - 1).max() < 1e-15            # partition of unity
    assert np.abs(dforms.sum(axis=2)).max() < 1e-15


def test_four_point_rule_uses_the_reference_literals():
    w, forms, dforms = ob.elem_table(ob.TET10, 4)
    a, b = 0.58541020, 0.13819660                                  # fea_solver.c:33-35
    assert dforms[0, 0, 1] == 4 * a - 1 and dforms[0, 1, 2] == 4 * b - 1


def test_27_point_rule_integrates_cubics():
    """config 5's rule: exact for total degree 3 on the unit tetrahedron
    (int r^a s^b t^c = a! b! c! / (a+b+c+3)!)."""
    from math import factorial as f
    L = ob.lib()
    t = ob.ElemTable()
    assert L.orc_elem_table_init(t, ob.TET10, 27) == 0
    # recover the points from the linear parts of the corner shape functions: N1+N4/2+N5/2+N8/2 = r etc.
    w, forms, _ = ob.elem_table(ob.TET10, 27)
    r = forms[:, 1] + 0.5 * (forms[:, 4] + forms[:, 5] + forms[:, 8])
    s = forms[:, 2] + 0.5 * (forms[:, 5] + forms[:, 6] + forms[:, 9])
    tt = forms[:, 3] + 0.5 * (forms[:, 7] + forms[:, 8] + forms[:, 9])
    assert (r > 0).all() and (s > 0).all() and (tt > 0).all() and (r + s + tt < 1).all()
    for a in range(4):
        for b in range(4 - a):
            for c in range(4 - a - b):
                exact = f(a) * f(b) * f(c) / f(a + b + c + 3)
                assert (w * r ** a * s ** b * tt ** c).sum() == pytest.approx(exact, rel=1e-13)


def test_unsupported_rule_rejected():
    with pytest.raises(ValueError):
        ob.elem_table(ob.TET10, 3)                                 # fea_solver.c:1503


@pytest.fixture(scope="module")
def brick(decks_dir):
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    o = OracleSolver(deck)
    o.update_state()
    return deck, o


def test_block_volume(brick):
    deck, o = brick
    w, _, _ = ob.elem_table(ob.TET10, 5)
    assert (np.abs(o.detj()) * w).sum() == pytest.approx(6.0, abs=1e-12)   # bar 1 x 6 x 1
    assert (o.detj() > 0).all()


def test_reference_configuration_is_stress_free(brick):
    deck, o = brick
    assert np.abs(o.graddefs() - np.eye(3)).max() < 1e-13
    assert np.abs(o.stresses()).max() < 1e-11
    o.create_residual_forces()
    assert np.abs(o.forces()).max() < 1e-12


def test_stiffness_symmetric_and_translation_free(brick):
    deck, o = brick
    x = mesh.deformed_state(deck.nodes, k1=1.04)
    o.set_nodes(x)
    o.update_state()
    o.create_stiffness()
    import scipy.sparse as sp
    K = sp.csr_matrix((o.values().copy(), o.indexes().copy(), o.offsets().copy()), shape=(o.ndof, o.ndof))
    scale = abs(K).max()
    assert abs(K - K.T).max() < 1e-12 * scale
    for ax in range(3):
        t = np.zeros(o.ndof)
        t[ax::3] = 1.0
        assert np.abs(K @ t).max() < 1e-11 * scale
        assert np.abs(o.spmv(t)).max() < 1e-11 * scale
    # assembled matrix = sum of the element matrices
    dense = np.zeros((o.ndof, o.ndof))
    for e in range(o.E):
        kc, ks = o.element_stiffness(e)
        dofs = (3 * deck.elements[e][:, None] + np.arange(3)).ravel()
        dense[np.ix_(dofs, dofs)] += kc + ks
    assert np.abs(K.toarray() - dense).max() < 1e-12 * scale
    o.set_nodes(deck.nodes)
    o.update_state()


def test_element_matrix_matches_compact_form(brick):
    """The 2 x 81-term loops of fea_solver.c:944-952 / 1035-1041 equal
    l1 g_a(x)g_b + m1 (g_a.g_b I + g_b(x)g_a) and (g_a.sigma.g_b) I --
    the identity the HIP kernels rely on."""
    deck, o = brick
    x = mesh.deformed_state(deck.nodes, k1=1.07)
    o.set_nodes(x)
    o.update_state()
    w, _, _ = ob.elem_table(ob.TET10, 5)
    g, dJ, F, S = o.grads(), o.detj(), o.graddefs(), o.stresses()
    lam, mu = deck.parameters[:2]
    for e in (0, 17, 345):
        kc, ks = o.element_stiffness(e)
        kc2, ks2 = np.zeros_like(kc), np.zeros_like(ks)
        for q in range(5):
            J = np.linalg.det(F[e, q])
            l1, m1 = lam / J, (mu - lam * np.log(J)) / J
            G = g[e, q]                                   # [3][npe]
            vol = w[q] * abs(dJ[e, q])
            gg = G.T @ G
            gsg = G.T @ S[e, q] @ G
            for a in range(10):
                for b in range(10):
                    blk = l1 * np.outer(G[:, a], G[:, b]) + m1 * (gg[a, b] * np.eye(3) + np.outer(G[:, b], G[:, a]))
                    kc2[3 * a:3 * a + 3, 3 * b:3 * b + 3] += vol * blk
                    ks2[3 * a:3 * a + 3, 3 * b:3 * b + 3] += vol * gsg[a, b] * np.eye(3)
        assert np.abs(kc - kc2).max() < 1e-12 * np.abs(kc).max()
        assert np.abs(ks - ks2).max() < 1e-12 * max(np.abs(ks).max(), 1e-30)
    o.set_nodes(deck.nodes)
    o.update_state()


def test_bc_application(brick):
    deck, o = brick
    o.update_nodes_with_bc(1.0)
    moved = o.nodes() - deck.nodes
    top = deck.presc_values[:, 1] != 0
    assert np.allclose(moved[deck.presc_node[top], 1], 0.05) and np.abs(moved).sum() == pytest.approx(0.05 * top.sum())
    o.update_state(); o.create_stiffness(); o.create_residual_forces()
    diag_before = {}
    off, idx, val = o.offsets(), o.indexes(), o.values()
    cd = np.concatenate([[3 * n, 3 * n + 1, 3 * n + 2] for n in deck.presc_node])
    for c in cd:
        row = slice(off[c], off[c + 1])
        diag_before[c] = val[row][idx[row] == c][0]
    o.apply_prescribed_bc(0.0)
    val = o.values()
    for c in cd:
        row = slice(off[c], off[c + 1])
        offd = val[row][idx[row] != c]
        assert np.all(offd == 0) and val[row][idx[row] == c][0] == diag_before[c]
    import scipy.sparse as sp
    K = sp.csr_matrix((val, idx, off), shape=(o.ndof, o.ndof)).tocsc()
    assert np.all(o.forces()[cd] == 0)
    for c in cd[:20]:
        col = K.getcol(int(c)).toarray().ravel()
        col[c] = 0
        assert np.all(col == 0)
    o.set_nodes(deck.nodes)
    o.update_state()


def test_tet4_extension_patch_test():
    """Linear tets on a Kuhn block with the rotation-free uniaxial recipe
    reproduce the Neo-Hookean closed form (SURVEY.md 4, TET4 probe)."""
    from test_oracle_closed_form import nh_closed_form
    deck = mesh.bar_deck(dims=(2, 6, 2), recipe="uniaxial", dy=0.05)
    o = OracleSolver(deck)
    done, its, tol = o.solve(1, 8, False, 1e-24, feahip.CHOLESKY)
    k1 = 1 + 0.05 / 6
    k2, syy = nh_closed_form(k1)
    assert np.abs(o.stresses()[:, 0, 1, 1] - syy).max() < 1e-10
    A = deck.nodes.min(axis=0)
    expect = A + (deck.nodes - A) * np.array([k2, k1, k2])
    assert np.abs(o.nodes() - expect).max() < 1e-12


@pytest.mark.parametrize("quadratic", [False, True])
def test_coloured_parallel_assembly_is_the_serial_one(quadratic):
    """bench.py's all-cores CPU baseline (the oracle's loops with the elements coloured and every colour an OpenMP
    parallel loop) against the oracle's serial loops: the same K and f to rounding (the order of the
    additions inside an entry follows the colours), on 1, 3 and all threads."""
    import mesh
    deck = mesh.bar_deck(dims=(3, 8, 4), quadratic=quadratic)
    x = mesh.deformed_state(deck.nodes, k1=1.06)
    o = OracleSolver(deck)
    o.set_nodes(x)
    o.update_state(); o.create_stiffness(); o.create_residual_forces()
    K, f = o.values().copy(), o.forces().copy()
    for nt in (1, 3, 0):
        p = OracleSolver(deck)
        p.set_nodes(x)
        ncol = p.assemble_coloured(nt)
        assert 4 <= ncol <= 64
        assert np.abs(p.values() - K).max() < 1e-13 * np.abs(K).max()
        assert np.abs(p.forces() - f).max() < 1e-13 * max(np.abs(f).max(), 1e-300)
