"""A rank that holds only its slab (fea-large_amd/csrc/rankmesh.cpp, feahip_create_rank).

The reference keeps one row-wise store of the whole matrix in one process (fea_solver.c:444-448).  A rank context holds
the sub-mesh of one rank -- owned nodes, the elements around them, their halo nodes -- locally indexed, and builds its
block rows from its own elements alone.  CPU: the rows a rank builds are the rows of the whole pattern, every node is
owned once, the halo plans of the ranks fit each other, and a rank of 8 holds an eighth of the memory (plus its
surface).  GPU: assembly, linear solve and the Newton loop over rank contexts equal the unsharded run.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import feahip
import mesh
from oracle_binding import OracleSolver


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def adjacency(deck):
    el = deck.elements
    n = el.shape[1]
    rows = np.repeat(el, n, axis=1).ravel(); cols = np.tile(el, (1, n)).ravel()
    A = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(len(deck.nodes),) * 2)
    A.sum_duplicates(); A.sort_indices()
    return A


@pytest.mark.parametrize("quadratic,nranks", [(False, 2), (False, 3), (False, 8), (True, 2), (True, 4)])
def test_rank_rows_are_the_rows_of_the_whole_pattern(quadratic, nranks):
    deck = mesh.bar_deck(dims=(3, 30, 3), quadratic=True) if quadratic else mesh.bar_deck(dims=(5, 64, 5))
    A = adjacency(deck)
    seen = np.zeros(len(deck.nodes), dtype=int)
    blocks = 0
    for r in range(nranks):
        m = feahip.host_rank_mesh(deck, r, nranks, pattern=True)
        own = m["node_global"][:m["owned_nodes"]]
        seen[own] += 1
        for i, a in enumerate(own):                           # the block row of the caller's node a, as this rank built it
            cols = np.sort(m["colidx"][m["rowptr"][i]:m["rowptr"][i + 1]])
            assert np.array_equal(cols, A.indices[A.indptr[a]:A.indptr[a + 1]])
        blocks += m["blocks_owned_rows"]
        # local elements = every element with an owned node; local nodes = their nodes, owned first
        touches = np.isin(deck.elements, own).any(axis=1)
        assert np.array_equal(m["elem_global"], np.nonzero(touches)[0])
        assert set(m["node_global"]) == set(np.unique(deck.elements[touches])) | set(own)
    assert np.all(seen == 1) and blocks == A.nnz


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_rank_halo_plans_fit_each_other(nranks):
    deck = mesh.bar_deck(dims=(4, 60, 4))
    plans = [feahip.host_rank_plan(deck, r, nranks) for r in range(nranks)]
    meshes = [feahip.host_rank_mesh(deck, r, nranks, pattern=True) for r in range(nranks)]
    for a in range(nranks):
        own_a = set(meshes[a]["node_global"][:meshes[a]["owned_nodes"]])
        halo_a = set(meshes[a]["node_global"][meshes[a]["owned_nodes"]:])
        got = set()
        for k, b in enumerate(plans[a]["peers"]):
            kb = plans[b]["peers"].index(a)
            assert np.array_equal(plans[a]["send"][k], plans[b]["recv"][kb])      # same rows, same order, both ways
            assert np.array_equal(plans[a]["recv"][k], plans[b]["send"][kb])
            assert set(plans[a]["send"][k]) <= own_a
            got |= set(plans[a]["recv"][k])
        assert got == halo_a                                  # every halo node arrives from exactly its owner


def test_a_rank_of_eight_holds_an_eighth():
    """BASELINE.json configs[4] (n = 112 TET10 block, 8 ranks) cannot be built on this host, but what a rank holds
    scales with its slab: at n = 14 (98 784 elements) a rank of 8 holds at most 1/8 of the nodes, elements and blocks
    plus its two cut surfaces, and from those counts a rank of the n = 112 block (512 x the volume, 64 x the surface)
    stays under 100 GB of device memory (K and its modified-Newton copy 35 GB, the 27-point state records 30 GB) -- a third of
    one MI355X's 288 GB."""
    n, nranks = 14, 8
    deck = mesh.bar_deck(n=n, quadratic=True)
    A = adjacency(deck)
    whole = {"nodes": len(deck.nodes), "elements": len(deck.elements), "blocks": A.nnz}
    worst = 0.0
    for r in (0, 3, 7):
        m = feahip.host_rank_mesh(deck, r, nranks)
        assert m["owned_nodes"] <= 1.02 * whole["nodes"] / nranks + 48
        # two cut surfaces of (2n+1)^2 nodes, two layers of elements each
        assert m["local_nodes"] <= m["owned_nodes"] + 2 * 2.2 * (2 * n + 1) ** 2
        assert m["blocks_owned_rows"] <= 1.05 * whole["blocks"] / nranks + 60 * 48
        vol = 8 ** 3                                          # n = 112 against n = 14
        surf = 8 ** 2
        halo_nodes = m["local_nodes"] - m["owned_nodes"]
        nodes112 = m["owned_nodes"] * vol + halo_nodes * surf
        blocks112_owned = m["blocks_owned_rows"] * vol
        blocks112_local = blocks112_owned + (m["blocks_local_rows"] - m["blocks_owned_rows"]) * surf
        elems112 = m["local_elements"] * vol
        # K window + modified-Newton copy, block pattern, gather maps + state records (27 points), vectors, mesh arrays
        dev = 2 * 72 * blocks112_owned + 4 * blocks112_local + elems112 * (400 + 27 * 144) + 12 * 24 * nodes112 + elems112 * 40 + nodes112 * 64
        worst = max(worst, dev)
    assert worst < 100e9


# ---------------------------------------------------------------------------------------------- on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,quadratic", [(2, False), (3, False), (2, True)])
def test_rank_contexts_assemble_and_solve_like_the_whole_mesh(n, quadratic):
    """n rank contexts in one process (in-process group transport): every rank's owned rows of K and f equal the
    unsharded context's (its local Yale matrix against the whole one, row by row through node_global), the linear solve
    gives the same displacement increment and <u,f>, and one Newton iteration moves the nodes to the same place."""
    deck = mesh.bar_deck(dims=(3, 20, 3), quadratic=True) if quadratic else mesh.bar_deck(dims=(3, 48, 3))
    x = mesh.deformed_state(deck.nodes, k1=1.03)
    one = feahip.FeaSolver(deck)
    one.set_nodes(x); one.create_stiffness_and_residual()
    off, idx, val = one.matrix_yale(); f = one.forces()
    K = sp.csr_matrix((val, idx, off), shape=(one.ndof, one.ndof))
    g = feahip.FeaGroup(deck, n, rank_contexts=True)
    seen = np.zeros(len(deck.nodes), dtype=int)
    for r in g.ranks:
        assert r.N < len(deck.nodes) and r.E < len(deck.elements)          # it holds its slab, not the mesh
        r.set_nodes(x[r.node_global]); r.create_stiffness_and_residual()
        lo, li, lv = r.matrix_yale()
        Kl = sp.csr_matrix((lv, li, lo), shape=(r.ndof, r.ndof))
        ng = r.node_global.astype(np.int64)
        gd = (3 * ng[:, None] + np.arange(3)[None, :]).ravel()               # local dof -> the deck's dof
        own = np.arange(3 * r.n_own)
        ref = K[gd[own]][:, gd]                                              # owned rows of the whole K, local column order
        assert abs(Kl[own] - ref).max() < 1e-13 * np.abs(val).max()
        assert np.abs(r.forces()[own] - f[gd[own]]).max() < 1e-13 * np.abs(f).max()
        seen[ng[:r.n_own]] += 1
    assert np.all(seen == 1)
    # the linear solve of the first Newton step
    one.set_nodes(deck.nodes); one.update_nodes_with_bc(1.0); one.create_stiffness_and_residual(); one.apply_prescribed_bc(0.0)
    it1, _ = one.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
    for r in g.ranks:
        r.set_nodes(deck.nodes[r.node_global])
    g.each("update_nodes_with_bc", 1.0); g.each("create_stiffness_and_residual"); g.each("apply_prescribed_bc", 0.0)
    itn, resn = g.solve_slae(feahip.PCG_ILU, 1e-15, 20000)
    assert resn < 1e-14 and abs(itn - it1) <= 2
    assert rel(g.gather("solution"), one.solution()) < 1e-11
    assert g.energy() == pytest.approx(one.energy(), rel=1e-11)
    one.update_nodes_with_solution(); g.update_nodes_with_solution()
    assert rel(g.gather("nodes") - deck.nodes, one.nodes() - deck.nodes) < 1e-11
    g.close(); one.close()


@pytest.mark.gpu
def test_rank_contexts_newton_matches_oracle(decks_dir):
    """The reference's clamped deck over two rank contexts: the oracle's iteration sequence (13 modified-Newton
    iterations), <u,f> of every iteration and the displacements within 1e-10 (BASELINE.json)."""
    deck = feahip.Deck.load(os.path.join(decks_dir, "neohook_brick.sexp"))
    o = OracleSolver(deck)
    od, oits, otol = o.solve(1, deck.max_newton_count, True, deck.desired_tolerance, feahip.CHOLESKY)
    g = feahip.FeaGroup(deck, 2, rank_contexts=True)
    gd, gits, gtol = g.solve(1, deck.max_newton_count, True, deck.desired_tolerance, feahip.CHOLESKY)
    assert gd == od == 1 and list(gits) == list(oits) == [13]
    assert np.abs(gtol - otol).max() < 1e-10 * np.abs(otol).max()
    assert rel(g.gather("nodes") - deck.nodes, o.nodes() - deck.nodes) < 1e-10
    g.close()


@pytest.mark.gpu
def test_rank_contexts_multigrid():
    """The sharded multigrid on rank contexts (every rank its own hierarchy): same solution as the unsharded solve."""
    deck = mesh.bar_deck(dims=(6, 126, 6))
    one = feahip.FeaSolver(deck)
    one.update_nodes_with_bc(1.0); one.create_stiffness_and_residual(); one.apply_prescribed_bc(0.0)
    one.solve_slae(feahip.PCG_ILU, 1e-15, 40000)
    g = feahip.FeaGroup(deck, 2, rank_contexts=True)
    g.each("set_preconditioner", 1)
    g.each("update_nodes_with_bc", 1.0); g.each("create_stiffness_and_residual"); g.each("apply_prescribed_bc", 0.0)
    it, res = g.solve_slae(feahip.PCG_ILU, 1e-15, 40000)
    assert res < 1e-14
    assert rel(g.gather("solution"), one.solution()) < 1e-10
    g.close(); one.close()
