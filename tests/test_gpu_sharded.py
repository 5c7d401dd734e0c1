"""The row-sharded solve on one MI355X: n contexts of the same mesh, each
owning a slab of rows, driven by the in-process group transport (same kernels,
same halo plan as the RCCL path; only the byte mover differs)."""
import os

import numpy as np
import pytest

import feahip
import mesh
from oracle_binding import OracleSolver

pytestmark = pytest.mark.gpu


def rel(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


@pytest.mark.parametrize("strategy,quadratic", [(feahip.ASM_AUTO, False), (feahip.ASM_STAGED, False), (feahip.ASM_AUTO, True)])
@pytest.mark.parametrize("n", [2, 3])
def test_sharded_assembly_is_the_unsharded_one(n, strategy, quadratic):
    """Every rank assembles the rows it owns from maps built for those rows alone.  Staged visits: the same bits as
    the unsharded run.  Gather (what AUTO picks here, for linear and for 10-node tetrahedra): the chunks of a shard
    start at its first row, so a block the unsharded run writes as the transpose of its mirror may be summed directly
    (or the other way round) -- equal to rounding; the residual of the 4-node kernel (no mirrors) to the bit, that of
    the 10-node kernel (a row's visits are summed in slices whose length depends on the chunk) to rounding."""
    deck = mesh.bar_deck(dims=(3, 20, 3), quadratic=True, brick=(3, 4, 4)) if quadratic else mesh.bar_deck(dims=(3, 40, 3))
    x = mesh.deformed_state(deck.nodes)
    one = feahip.FeaSolver(deck)
    one.set_nodes(x)
    one.set_assembly(strategy)
    one.create_stiffness_and_residual()
    off, idx, val = one.matrix_yale()
    f = one.forces()
    g = feahip.FeaGroup(deck, n)
    g.each("set_nodes", x)
    g.each("set_assembly", strategy)
    g.each("create_stiffness_and_residual")
    seen = np.zeros(len(deck.nodes), dtype=int)
    kscale, fscale = np.abs(val).max(), np.abs(f).max()
    row_of_value = np.repeat(np.arange(one.ndof), np.diff(off))   # Yale values -> the caller's dof row
    for nd, r in zip(g.nodes, g.ranks):                           # nd: the caller's ids of the rank's nodes (a slab of the
        seen[nd] += 1                                             # library's numbering, not a range of the caller's)
        own_node = np.zeros(len(deck.nodes), dtype=bool); own_node[nd] = True
        mine = own_node[row_of_value // 3]
        d = r.owned_dofs()
        _, _, v = r.matrix_yale()
        if strategy == feahip.ASM_STAGED:
            assert np.array_equal(v[mine], val[mine])             # owned rows: same bits
        else:
            assert r.assembly_in_use() == feahip.ASM_GATHER
            assert np.abs(v[mine] - val[mine]).max() < 4e-16 * kscale
        assert np.all(v[~mine] == 0)                              # nothing else written
        if quadratic:
            assert np.abs(r.forces()[d] - f[d]).max() < 4e-16 * fscale
            r.create_residual_forces()                            # the residual alone: same kernel without the blocks
            assert np.abs(r.forces()[d] - f[d]).max() < 4e-16 * fscale
        else:
            assert np.array_equal(r.forces()[d], f[d])
    assert np.all(seen == 1)
    g.close(); one.close()


@pytest.mark.parametrize("rank_contexts", [False, True])
@pytest.mark.parametrize("precond", [0, 1])
def test_interior_product_never_reads_a_halo_row(rank_contexts, precond, monkeypatch):
    """The single-reduction PCG multiplies the rows that touch no halo column WHILE the halo rows of z travel on the
    communication stream (exchange_begin ... exchange_end; the default loop of every sharded solve, RCCL included).
    FEAHIP_TEST_POISON_HALO makes the in-process transport overwrite the halo rows with NaN at exchange_begin and hold
    the copies back until everything enqueued up to exchange_end has run: if the interior range (install_plan) held a
    chunk with a halo column, or an event dependency were missing, NaN would reach w and the solve.  It must be the
    solve of the unpoisoned run, bit for bit."""
    deck = mesh.bar_deck(dims=(6, 96, 6) if precond else (3, 48, 3))

    def run(n):
        g = feahip.FeaGroup(deck, n, rank_contexts=rank_contexts)
        g.each("set_pcg_variant", 1); g.each("set_preconditioner", precond)
        g.each("update_nodes_with_bc", 1.0); g.each("create_stiffness_and_residual"); g.each("apply_prescribed_bc", 0.0)
        it, res = g.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
        u, e = g.gather("solution"), g.energy()
        g.close()
        return it, res, u, e

    for n in (2, 3):
        monkeypatch.delenv("FEAHIP_TEST_POISON_HALO", raising=False)
        it0, res0, u0, e0 = run(n)
        monkeypatch.setenv("FEAHIP_TEST_POISON_HALO", "1")
        it1, res1, u1, e1 = run(n)
        assert np.isfinite(u1).all() and np.isfinite(res1) and res1 < 1e-14
        assert it1 == it0 and np.array_equal(u1, u0) and e1 == e0


@pytest.mark.parametrize("n,solver", [(2, feahip.PCG_ILU), (3, feahip.CG)])
def test_sharded_linear_solve(n, solver):
    deck = mesh.bar_deck(dims=(3, 48, 3))
    one = feahip.FeaSolver(deck)
    one.update_nodes_with_bc(1.0); one.create_stiffness_and_residual(); one.apply_prescribed_bc(0.0)
    it1, res1 = one.solve_slae(solver, 1e-15, 20000)
    g = feahip.FeaGroup(deck, n)
    g.each("update_nodes_with_bc", 1.0); g.each("create_stiffness_and_residual"); g.each("apply_prescribed_bc", 0.0)
    itn, resn = g.solve_slae(solver, 1e-15, 20000)
    assert resn < 1e-14 and abs(itn - it1) <= 2
    assert rel(g.gather("solution"), one.solution()) < 1e-11
    assert g.energy() == pytest.approx(one.energy(), rel=1e-11)
    g.close(); one.close()


@pytest.mark.parametrize("solver,precond", [(feahip.PCG_ILU, 0), (feahip.CG, 0), (feahip.PCG_ILU, 1)])
def test_single_reduction_pcg_is_the_same_solve(solver, precond):
    """feahip_set_pcg_variant: the single-reduction loop (one all-reduce of {r.z, w.z, r.r} per iteration, halo rows of
    z in flight under the rows that touch no halo column) against the reference-shaped two-reduction loop, on one
    context and on 2 and 3 in-process ranks, block-Jacobi, plain CG and the multigrid: the same solution to 1e-11,
    the same iteration count within 2, the same <u,f>.  A sharded context runs the single-reduction loop by default."""
    deck = mesh.bar_deck(dims=(6, 96, 6) if precond else (3, 48, 3))
    ref = feahip.FeaSolver(deck)
    ref.set_pcg_variant(0); ref.set_preconditioner(precond)
    ref.update_nodes_with_bc(1.0); ref.create_stiffness_and_residual(); ref.apply_prescribed_bc(0.0)
    it0, res0 = ref.solve_slae(solver, 1e-15, 20000)
    u0, e0 = ref.solution(), ref.energy()
    one = feahip.FeaSolver(deck)
    one.set_pcg_variant(1); one.set_preconditioner(precond)
    one.update_nodes_with_bc(1.0); one.create_stiffness_and_residual(); one.apply_prescribed_bc(0.0)
    it1, res1 = one.solve_slae(solver, 1e-15, 20000)
    assert res1 < 1e-14 and abs(it1 - it0) <= 2
    assert rel(one.solution(), u0) < 1e-11 and one.energy() == pytest.approx(e0, rel=1e-11)
    for n in (2, 3):
        for variant in (-1, 0):                                    # -1: the default of a sharded context = single reduction
            g = feahip.FeaGroup(deck, n)
            g.each("set_pcg_variant", variant); g.each("set_preconditioner", precond)
            g.each("update_nodes_with_bc", 1.0); g.each("create_stiffness_and_residual"); g.each("apply_prescribed_bc", 0.0)
            itn, resn = g.solve_slae(solver, 1e-15, 20000)
            assert resn < 1e-14
            if not precond:                                       # (the multigrid is block-Jacobi over the ranks: another preconditioner)
                assert abs(itn - it0) <= 2
            assert rel(g.gather("solution"), u0) < 1e-11
            assert g.energy() == pytest.approx(e0, rel=1e-11)
            g.close()
    one.close(); ref.close()


def test_sharded_newton_matches_oracle(decks_dir):
    """The shipped clamped deck on 2 ranks: same iteration sequence as the
    oracle, displacements within 1e-10 (BASELINE.json)."""
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    o = OracleSolver(deck)
    od, oits, otol = o.solve(1, deck.max_newton_count, True, deck.desired_tolerance, feahip.CHOLESKY)
    g = feahip.FeaGroup(deck, 2)
    gd, gits, gtol = g.solve(1, deck.max_newton_count, True, deck.desired_tolerance, feahip.CHOLESKY)
    assert gd == od == 1 and list(gits) == list(oits) == [13]
    assert np.abs(gtol - otol).max() < 1e-10 * np.abs(otol).max()
    assert rel(g.gather("nodes") - deck.nodes, o.nodes() - deck.nodes) < 1e-10
    g.close()


def test_sharded_full_newton_patch_test():
    from test_oracle_closed_form import nh_closed_form
    deck = mesh.bar_deck(dims=(3, 30, 3), recipe="uniaxial", dy=0.05)
    g = feahip.FeaGroup(deck, 3)
    done, its, tol = g.solve(1, 8, False, 1e-22, feahip.PCG_ILU, 1e-15)
    k1 = 1 + 0.05 / 6
    k2, syy = nh_closed_form(k1)
    A = deck.nodes.min(axis=0)
    expect = A + (deck.nodes - A) * np.array([k2, k1, k2])
    assert np.abs(g.gather("nodes") - expect).max() < 1e-11
    g.close()


@pytest.mark.parametrize("n", [2, 3])
def test_sharded_multigrid_preconditioner(n):
    """Sharded solve with feahip_set_preconditioner(1): every rank runs the
    W-cycle on its own diagonal block (block-Jacobi over the ranks, no
    communication inside the preconditioner).  Same solution as the unsharded
    block-Jacobi solve, far fewer iterations than sharded block-Jacobi, and a
    sharded Newton step lands where the unsharded one does."""
    deck = mesh.bar_deck(dims=(6, 126, 6))
    one = feahip.FeaSolver(deck)
    one.update_nodes_with_bc(1.0); one.create_stiffness_and_residual(); one.apply_prescribed_bc(0.0)
    it_bj, _ = one.solve_slae(feahip.PCG_ILU, 1e-15, 40000)
    g = feahip.FeaGroup(deck, n)
    g.each("set_preconditioner", 1)
    g.each("update_nodes_with_bc", 1.0); g.each("create_stiffness_and_residual"); g.each("apply_prescribed_bc", 0.0)
    it_mg, res = g.solve_slae(feahip.PCG_ILU, 1e-15, 40000)
    assert res < 1e-14 and 0 < it_mg < 0.5 * it_bj
    assert rel(g.gather("solution"), one.solution()) < 1e-10
    assert g.energy() == pytest.approx(one.energy(), rel=1e-10)
    g.close(); one.close()
    deck = mesh.bar_deck(dims=(6, 126, 6), dy=0.01, load_increments_count=1, max_newton_count=12, modified_newton=False,
                         desired_tolerance=1e-16)
    a = feahip.FeaSolver(deck)
    ra = a.solve(solver_type=feahip.PCG_ILU, solver_tolerance=1e-15)
    g = feahip.FeaGroup(deck, n)
    g.each("set_preconditioner", 1)
    gd, gits, gtol = g.solve(1, 12, deck.modified_newton, deck.desired_tolerance, feahip.PCG_ILU, 1e-15)
    assert gd == ra[0] == 1 and list(gits) == list(ra[1])
    assert rel(g.gather("nodes") - deck.nodes, a.nodes() - deck.nodes) < 1e-10
    g.close(); a.close()


def test_sharded_line_search_equals_single_rank():
    """The golden-section line search inside the Newton driver on 2 ranks (halo
    copies of u exchanged once per step, trial configurations on every rank):
    same iteration count, same <u,f> sequence, same nodes as one rank."""
    deck = mesh.bar_deck(dims=(3, 24, 3), dy=0.3, load_increments_count=1, max_newton_count=25,
                         desired_tolerance=1e-16, modified_newton=False)
    one = feahip.FeaSolver(deck)
    d1, its1, tol1 = one.solve(solver_type=feahip.CHOLESKY, line_search=4)
    g = feahip.FeaGroup(deck, 2)
    g.each("set_line_search", 4)
    gd, gits, gtol = g.solve(1, 25, False, 1e-16, feahip.CHOLESKY)
    assert gd == d1 == 1 and list(gits) == list(its1)
    assert np.abs(gtol - tol1).max() < 1e-9 * np.abs(tol1).max()
    assert rel(g.gather("nodes") - deck.nodes, one.nodes() - deck.nodes) < 1e-10
    g.close(); one.close()


def test_rccl_single_rank_comm():
    """RCCL transport with one rank (all this box can host): unique id,
    communicator, all-reduce of the CG scalars, empty halo exchange."""
    deck = mesh.bar_deck(dims=(2, 12, 2))
    s = feahip.FeaSolver(deck)
    s.comm_init(0, 1, feahip.comm_unique_id())
    assert s.owned_rows() == (0, len(deck.nodes))
    s.update_nodes_with_bc(1.0); s.create_stiffness_and_residual(); s.apply_prescribed_bc(0.0)
    it, res = s.solve_slae(feahip.PCG_ILU, 1e-14, 5000)
    ref = feahip.FeaSolver(deck)
    ref.update_nodes_with_bc(1.0); ref.create_stiffness_and_residual(); ref.apply_prescribed_bc(0.0)
    it2, _ = ref.solve_slae(feahip.PCG_ILU, 1e-14, 5000)
    # (the communicator selects the single-reduction recurrence: same iterates in exact arithmetic, counts within 2)
    assert abs(it - it2) <= 2 and rel(s.solution(), ref.solution()) < 1e-13
    assert s.energy() == pytest.approx(ref.energy(), rel=1e-13)
    s.close(); ref.close()
