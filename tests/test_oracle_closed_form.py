"""Pins the oracle's element / Newton loops with the reference's known answers.

fea_solver.c cannot be compiled in this image (its libspmatrix / libsexp /
liblogger headers are not in the tree), so the restated loops are pinned by
what the reference itself holds for this path:
  * its own input decks  solver-large/data/*_analytical.sexp (fixtures in
    tests/golden/decks), which put the bar in homogeneous uniaxial tension,
  * its own closed-form answers for exactly that state:
      exact-solutions/uniaxial/uniaxial_neohookean_bonet.m:20-45
      exact-solutions/uniaxial/uniaxial.m:1-44 (model A5, n = 5)
    with stretch step 0.05/6 per increment (the constant 0.008333 of
    uniaxial_neohookean_bonet.m:40 and fea_solver.c:1447).
Driven by full Newton to machine precision the FE answer equals the closed
form at every Gauss point (quadratic tets pass the finite-strain patch test).
"""
import math
import os

import numpy as np
import pytest

import feahip
from oracle_binding import OracleSolver

LAM = MU = 100.0


def nh_closed_form(k1):
    k2 = 1.0
    for _ in range(80):                       # fsolve of k2_from_sigma22 (:20-26)
        f = MU * (k2 * k2 - 1) + LAM * math.log(k1 * k2 * k2)
        k2 -= f / (2 * MU * k2 + 2 * LAM / k2)
    J = k1 * k2 * k2
    return k2, (MU * (k1 ** 2 - 1) + LAM * math.log(J)) / J       # Txx (:28-37)


def a5_closed_form(k1, n=5):
    kk1 = k1 ** (n - 3)
    kk2 = (3 * LAM + 2 * MU - LAM * kk1) / (2 * LAM + 2 * MU)    # k2_An (uniaxial.m:30-34)
    k2 = kk2 ** (1.0 / (n - 3))
    s = (k1 ** (n - 3 - 1) / k2 ** 2) * ((LAM + 2 * MU) * kk1 + 2 * LAM * kk2 - (3 * LAM + 2 * MU)) / (n - 3)
    return k2, s


def test_closed_form_table_of_baseline_md():
    # the values BASELINE.md section 2 quotes from these formulas
    assert nh_closed_form(1 + 1 * 0.05 / 6)[1] == pytest.approx(2.079482974179, abs=5e-12)
    assert nh_closed_form(1 + 2 * 0.05 / 6)[1] == pytest.approx(4.151485075847, abs=5e-12)
    assert nh_closed_form(2.0)[1] == pytest.approx(241.938011182922, abs=5e-10)
    assert a5_closed_form(1 + 1 * 0.05 / 6)[1] == pytest.approx(2.118310407550, abs=5e-12)
    assert a5_closed_form(1.5)[1] == pytest.approx(340.909090909091, abs=5e-10)
    assert a5_closed_form(2.0)[1] == pytest.approx(3000.0, abs=1e-9)


@pytest.mark.parametrize("name,closed,steps,newton", [
    ("neohook_brick_analytical", nh_closed_form, 2, 7),
    ("a5_brick_analytical", a5_closed_form, 1, 13),
])
def test_analytical_deck_reaches_closed_form(decks_dir, name, closed, steps, newton):
    deck = feahip.Deck.load(os.path.join(decks_dir, name + ".sexp"))
    o = OracleSolver(deck)
    # full Newton, converged far below the deck's 1e-6; the decks leave one rigid
    # rotation free (singular tangent), which a Krylov solve tolerates
    done, its, tol = o.solve(steps, newton, False, 1e-24, feahip.PCG_ILU, 1e-13, 20000)
    k1 = 1 + steps * 0.05 / 6
    k2, syy = closed(k1)
    S, F = o.stresses(), o.graddefs()
    assert np.abs(S[:, :, 1, 1] - syy).max() < 2e-10 * syy           # every Gauss point
    assert np.abs(S[:, :, 0, 0]).max() < 1e-9 and np.abs(S[:, :, 2, 2]).max() < 1e-9
    # F = R diag(k2,k1,k2): compare rotation-free quantities
    assert np.abs(F[:, :, 1, 1] - k1).max() < 1e-11
    assert np.abs(np.linalg.det(F) - k1 * k2 * k2).max() < 1e-11
    C = np.einsum("egki,egkj->egij", F, F)
    assert np.abs(np.trace(C, axis1=2, axis2=3) - (k1 ** 2 + 2 * k2 ** 2)).max() < 1e-10
    # quadratic convergence of <u,f> in the first step (Neo-Hookean tangent is consistent)
    if "neohook" in name:
        first = np.abs(tol[:its[0]])
        assert first[0] == pytest.approx(5.39291681, rel=1e-8)
        assert first[3] < 1e-7 and first[4] < 1e-16


def test_modified_newton_iteration_counts_on_clamped_deck(decks_dir):
    """The deck's own settings (modified Newton, tolerance 1e-6 on <u,f>):
    13 and 12 iterations for the first two increments, and the direct and the
    Krylov solver agree on every <u,f> (the solution is pinned by K u = f)."""
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    o = OracleSolver(deck)
    done, its, tol = o.solve(2, deck.max_newton_count, True, deck.desired_tolerance, feahip.CHOLESKY)
    assert done == 2 and list(its) == [13, 12]
    o2 = OracleSolver(deck)
    done2, its2, tol2 = o2.solve(2, deck.max_newton_count, True, deck.desired_tolerance, feahip.PCG_ILU, 1e-15)
    assert list(its2) == [13, 12]
    assert np.abs(tol - tol2).max() < 1e-11
    assert np.abs(o.nodes() - o2.nodes()).max() < 1e-12


def test_lame_cylinder_small_strain_matches_lame_solution():
    """BASELINE configs[3]: the thick-walled cylinder of exact-solutions/lame
    (r in [1,2], plane strain: end faces keep their axial coordinate), inner
    surface pushed out by a small d, outer surface free, model A5.  In the
    small-strain limit u(r) = A r + B / r with sigma_rr(r_o) = 0 and
    u(r_i) = d: for lambda = mu, B = 2 A r_o^2, A = d / 9, so u(r_o) = 2 d / 3
    and u(1.5) = (1.5 + 16/3) d / 9.  Quadratic tets on a 2 x 16 x 1 mapped
    Kuhn block reach that within 1 %."""
    import mesh
    d = 1e-4
    deck = mesh.cylinder_deck(2, 16, 1, quadratic=True, zlo=0.0, zhi=0.5, du=d, load_increments_count=1,
                              max_newton_count=6, desired_tolerance=1e-24, modified_newton=False)
    o = OracleSolver(deck)
    done, its, tol = o.solve(1, 6, False, 1e-24, feahip.CHOLESKY)
    assert done == 1
    u = o.nodes() - deck.nodes
    r = np.hypot(deck.nodes[:, 0], deck.nodes[:, 1])
    ur = (u[:, 0] * deck.nodes[:, 0] + u[:, 1] * deck.nodes[:, 1]) / r
    assert np.abs(u[:, 2]).max() < 2e-3 * d        # plane strain (mid-plane nodes drift by the mesh's asymmetry only)
    for radius, expect in ((1.0, d), (1.5, (1.5 + 8 / 1.5) * d / 9), (2.0, 2 * d / 3)):
        sel = np.abs(r - radius) < 1e-9
        assert sel.sum() >= 32
        assert np.abs(ur[sel] - expect).max() < 0.01 * d, (radius, ur[sel].min(), ur[sel].max(), expect)
    o.close()


@pytest.mark.parametrize("name,closed", [("neohook_brick_analytical", nh_closed_form), ("a5_brick_analytical", a5_closed_form)])
@pytest.mark.parametrize("n", [3, 60, 120])
def test_oracle_at_finite_strain_homogeneous_state(decks_dir, name, closed, n):
    """Pins the restated stress / residual / tangent loops at FINITE strain (k1 up to 2, the end of the decks' 120
    increments) without a 120-step Newton run: put the reference's analytical deck into the exact homogeneous
    uniaxial state x = A + diag(k2, k1, k2)(X - A) of the closed form and check
      * sigma_yy at every Gauss point = the closed form, lateral stresses zero, F = diag(k2, k1, k2);
      * the assembled residual vanishes at every node off the loaded faces (the state is an equilibrium);
      * Neo-Hookean: the assembled tangent is the derivative of the assembled residual (central differences) --
        ln J, the spatial tangent and both stiffness parts are exercised far from the reference configuration."""
    deck = feahip.Deck.load(os.path.join(decks_dir, name + ".sexp"))
    k1 = 1 + n * 0.05 / 6
    k2, syy = closed(k1)
    A = deck.nodes.min(axis=0)
    x = A + (deck.nodes - A) * np.array([k2, k1, k2])
    o = OracleSolver(deck)
    o.set_nodes(x)
    o.update_state()
    S, F = o.stresses(), o.graddefs()
    assert np.abs(S[:, :, 1, 1] - syy).max() < 1e-11 * abs(syy)
    lateral = max(np.abs(S[:, :, 0, 0]).max(), np.abs(S[:, :, 2, 2]).max(), np.abs(S[:, :, 0, 1]).max())
    assert lateral < 1e-10 * abs(syy)
    assert np.abs(F - np.diag([k2, k1, k2])).max() < 1e-12
    o.create_residual_forces()
    f = o.forces().reshape(-1, 3)
    y = deck.nodes[:, 1]
    inner = (y > y.min() + 1e-9) & (y < y.max() - 1e-9)
    area = k2 * k2                                   # deformed cross-section of the unit bar
    assert np.abs(f[inner]).max() < 1e-10 * abs(syy) * area
    assert np.abs(f[~inner][:, 1]).sum() == pytest.approx(2 * abs(syy) * area, rel=2e-6)     # the two faces carry sigma_yy * area (deck coordinates have 7 digits)
    if "neohook" in name:
        o.create_stiffness()
        rng = np.random.default_rng(7)
        d = rng.standard_normal(x.shape)
        Kd = o.spmv(d.ravel())
        eps = 1e-6
        fs = []
        for sgn in (+1, -1):
            o.set_nodes(x + sgn * eps * d)
            o.update_state()
            o.create_residual_forces()
            fs.append(o.forces().copy())
        fd = (fs[1] - fs[0]) / (2 * eps)             # the residual vector holds MINUS the internal forces (fea_solver.c:1109)
        assert np.abs(Kd - fd).max() < 2e-7 * np.abs(Kd).max()
    o.close()
