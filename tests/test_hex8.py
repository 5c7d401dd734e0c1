"""HEXAHEDRA8: the 8-node trilinear brick of BASELINE.json's "synthetic hex/tet meshes".

The reference has tetrahedra only (fea_solver.h:68-71), so like TETRAHEDRA4 the brick is a build extension that
goes through the reference's generic loops (they are generic in nodes_per_element, gauss_nodes_count, isoform and
disoform: fea_solver.c:503-535, 690-718, 932-971) with its own table: unit-cube parent element, 2 x 2 x 2 Gauss
points.  What pins it: the structural identities of any isoparametric element, and the same closed forms the
reference holds for the uniaxial state (exact-solutions/uniaxial) -- a trilinear brick reproduces a homogeneous
deformation exactly, so on a block of bricks they are exact finite-element answers.
"""
import numpy as np
import pytest

import feahip
import mesh
import oracle_binding as ob
from oracle_binding import OracleSolver
from test_oracle_closed_form import a5_closed_form, nh_closed_form


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def test_hex8_table_identities():
    w, N, dN = ob.elem_table(ob.HEX8, 8)
    assert w.sum() == pytest.approx(1.0, abs=1e-15)                   # volume of the parent cube
    assert np.abs(N.sum(axis=1) - 1).max() < 1e-15                    # partition of unity
    assert np.abs(dN.sum(axis=2)).max() < 1e-15
    # the rule integrates the trilinear mass matrix exactly: int N_a = 1/8
    assert np.abs((w[:, None] * N).sum(axis=0) - 1 / 8).max() < 1e-15
    with pytest.raises(ValueError):
        ob.elem_table(ob.HEX8, 5)


@pytest.mark.parametrize("model,closed", [(feahip.MODEL_COMPRESSIBLE_NEOHOOKEAN, nh_closed_form), (feahip.MODEL_A5, a5_closed_form)])
@pytest.mark.parametrize("k1", [1.05, 1.5])
def test_oracle_hex8_homogeneous_state(model, closed, k1):
    deck = mesh.bar_deck(dims=(3, 7, 2), hexa=True, model=model)
    k2, syy = closed(k1)
    A = deck.nodes.min(axis=0)
    x = A + (deck.nodes - A) * np.array([k2, k1, k2])
    o = OracleSolver(deck)
    o.update_state()
    assert (o.detj() * ob.elem_table(ob.HEX8, 8)[0][None, :]).sum() == pytest.approx(6.0, abs=1e-12)      # volume of the bar
    o.set_nodes(x)
    o.update_state()
    S, F = o.stresses(), o.graddefs()
    assert np.abs(S[:, :, 1, 1] - syy).max() < 1e-11 * abs(syy)
    assert max(np.abs(S[:, :, 0, 0]).max(), np.abs(S[:, :, 2, 2]).max(), np.abs(S[:, :, 0, 1]).max()) < 1e-10 * abs(syy)
    assert np.abs(F - np.diag([k2, k1, k2])).max() < 1e-12
    o.create_residual_forces()
    f = o.forces().reshape(-1, 3)
    y = deck.nodes[:, 1]
    inner = (y > y.min() + 1e-9) & (y < y.max() - 1e-9)
    assert np.abs(f[inner]).max() < 1e-11 * abs(syy)
    assert np.abs(f[~inner][:, 1]).sum() == pytest.approx(2 * abs(syy) * k2 * k2, rel=1e-11)
    if model == feahip.MODEL_COMPRESSIBLE_NEOHOOKEAN:                # tangent = derivative of the residual
        o.create_stiffness()
        d = np.random.default_rng(3).standard_normal(x.shape)
        Kd = o.spmv(d.ravel())
        fs = []
        for sgn in (+1, -1):
            o.set_nodes(x + sgn * 1e-6 * d); o.update_state(); o.create_residual_forces()
            fs.append(o.forces().copy())
        assert np.abs(Kd - (fs[1] - fs[0]) / 2e-6).max() < 2e-7 * np.abs(Kd).max()
    o.close()


def test_hex8_deck_round_trip(tmp_path):
    deck = mesh.bar_deck(dims=(2, 3, 2), hexa=True)
    p = tmp_path / "hex.sexp"
    deck.save(str(p))
    assert "HEXAHEDRA8" in p.read_text()
    back = feahip.Deck.load(str(p))
    assert back.ele_type == feahip.HEXAHEDRA8 and back.nodes_per_element == 8 and back.gauss_nodes_count == 8
    assert np.array_equal(back.elements, deck.elements) and np.array_equal(back.nodes, deck.nodes)


@pytest.mark.gpu
@pytest.mark.parametrize("model", [feahip.MODEL_COMPRESSIBLE_NEOHOOKEAN, feahip.MODEL_A5])
def test_hex8_assembly_matches_oracle(model):
    deck = mesh.bar_deck(dims=(4, 9, 3), hexa=True, model=model)
    x = mesh.deformed_state(deck.nodes, k1=1.08, wiggle=5e-3)
    s, o = feahip.FeaSolver(deck), OracleSolver(deck)
    s.set_nodes(x); o.set_nodes(x)
    o.update_state(); o.create_stiffness(); o.create_residual_forces()
    for strat in (feahip.ASM_AUTO, feahip.ASM_GATHER, feahip.ASM_ROWOWNER, feahip.ASM_ATOMIC):
        s.set_assembly(strat)
        s.create_stiffness_and_residual()
        off, idx, val = s.matrix_yale()
        assert np.array_equal(off, o.offsets()) and np.array_equal(idx, o.indexes())      # bit-exact indexing
        assert rel(val, o.values()) < 1e-12 and rel(s.forces(), o.forces()) < 1e-12, strat
        if strat == feahip.ASM_AUTO:
            assert s.assembly_in_use() == feahip.ASM_GATHER      # state kernel + gather kernel, as for 10-node tetrahedra
        s.create_residual_forces()                               # the residual alone
        assert rel(s.forces(), o.forces()) < 1e-12, strat
    assert s.assembly_in_use() == feahip.ASM_ATOMIC
    assert rel(s.graddefs(), o.graddefs()) < 1e-13 and rel(s.stresses(), o.stresses()) < 1e-12
    g, d = s.shape_gradients()
    assert rel(g, o.grads()) < 1e-13 and rel(d, o.detj()) < 1e-13
    s.close(); o.close()


@pytest.mark.gpu
def test_hex8_newton_patch_test_reaches_the_closed_form():
    """Uniaxial recipe of SURVEY.md 8(d) on a block of bricks: three load increments of full Newton on the HIP
    path give the closed-form sigma_yy at every Gauss point, the same iteration sequence and displacements as the
    oracle."""
    kw = dict(dims=(3, 12, 3), hexa=True, recipe="uniaxial", dy=0.05, load_increments_count=3, max_newton_count=12,
              desired_tolerance=1e-22, modified_newton=False)
    deck = mesh.bar_deck(**kw)
    s, o = feahip.FeaSolver(deck), OracleSolver(deck)
    sd, sits, stol = s.solve(solver_type=feahip.CHOLESKY)
    od, oits, otol = o.solve(3, 12, False, 1e-22, feahip.CHOLESKY)
    assert sd == od == 3 and list(sits[:3]) == list(oits[:3])
    k1 = 1 + 3 * 0.05 / 6
    k2, syy = nh_closed_form(k1)
    S = s.stresses()
    assert np.abs(S[:, :, 1, 1] - syy).max() < 1e-10 * syy
    assert rel(s.nodes() - deck.nodes, o.nodes() - deck.nodes) < 1e-10
    s.close(); o.close()
