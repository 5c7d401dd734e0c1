"""BASELINE.json sizes (configs[1] = 1M linear tets, configs[2] = 10M) checked
through size-independent properties -- the oracle would need minutes there:
K symmetric, K.(rigid translation) = 0 before BCs, stress-free reference
state, residual = -K.u consistency, the uniaxial closed form as the exact FE
answer (patch test), sharded = unsharded."""
import os

import numpy as np
import pytest

import feahip
import mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def block_1m():
    deck = mesh.bar_deck(n=31, recipe="clamped")          # 1 072 476 TET4, 191 488 nodes
    s = feahip.FeaSolver(deck)
    yield deck, s
    s.close()


def test_1m_counts_and_reference_state(block_1m):
    deck, s = block_1m
    assert len(deck.elements) == 1072476 and len(deck.nodes) == 191488
    s.set_nodes(deck.nodes)
    s.create_residual_forces()
    assert np.abs(s.forces()).max() < 1e-11
    assert s.update_state() == 0


def test_1m_stiffness_properties(block_1m):
    deck, s = block_1m
    s.set_nodes(mesh.deformed_state(deck.nodes))
    s.create_stiffness_and_residual()
    rng = np.random.default_rng(5)
    scale = None
    for ax in range(3):
        t = np.zeros(s.ndof); t[ax::3] = 1.0
        y = s.spmv(t)
        if scale is None:
            scale = np.abs(s.spmv(rng.normal(size=s.ndof))).max()
        assert np.abs(y).max() < 1e-11 * scale                     # translation-free
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    assert abs(a @ s.spmv(b) - b @ s.spmv(a)) < 1e-11 * abs(a @ s.spmv(b))   # symmetric
    # the three strategies agree at this size too
    f0 = s.forces()
    ya = s.spmv(a)
    for strat in (feahip.ASM_ROWOWNER, feahip.ASM_STAGED):
        s.set_assembly(strat)
        s.create_stiffness_and_residual()
        assert np.abs(s.spmv(a) - ya).max() < 1e-12 * np.abs(ya).max()
        assert np.abs(s.forces() - f0).max() < 1e-12 * np.abs(f0).max()
    s.set_assembly(feahip.ASM_AUTO)


def test_1m_uniaxial_patch_test_closed_form():
    """config 2 recipe: prescribed-displacement uniaxial tension; the closed form of
    exact-solutions/uniaxial/uniaxial_neohookean_bonet.m is the exact FE answer."""
    from test_oracle_closed_form import nh_closed_form
    deck = mesh.bar_deck(n=31, recipe="uniaxial")
    dy = mesh.increment_for(31)
    s = feahip.FeaSolver(deck)
    done, its, tol = s.solve(load_increments=1, max_newton=8, modified_newton=False, desired_tolerance=1e-22,
                             solver_type=feahip.PCG_ILU, solver_tolerance=1e-15, solver_max_iter=40000)
    assert done == 1 or its[0] == 8
    k1 = 1 + dy / 6
    k2, syy = nh_closed_form(k1)
    A = deck.nodes.min(axis=0)
    expect = A + (deck.nodes - A) * np.array([k2, k1, k2])
    assert np.abs(s.nodes() - expect).max() < 2e-11
    S = s.stresses()
    assert np.abs(S[:, 0, 1, 1] - syy).max() < 1e-8
    s.close()


def test_1m_multigrid_newton_step_equals_block_jacobi(block_1m):
    """One Newton step of the clamped 1M-tet block solved with both
    preconditioners: same <u,f>, same displacements to 1e-10, and the
    multigrid PCG needs less than a sixth of the iterations."""
    deck, s = block_1m
    out = []
    for kind in (0, 1):
        s.set_preconditioner(kind)
        s.set_nodes(deck.nodes)
        s.update_nodes_with_bc(1.0)
        s.create_stiffness_and_residual()
        s.apply_prescribed_bc(0.0)
        it, res = s.solve_slae(feahip.PCG_ILU, 1e-14, 40000)
        assert res < 1e-13
        out.append((it, s.energy(), s.solution()))
    s.set_preconditioner(0)
    (it0, e0, u0), (it1, e1, u1) = out
    assert it1 * 6 < it0
    assert abs(e1 - e0) < 1e-10 * abs(e0)
    assert np.abs(u1 - u0).max() < 1e-10 * np.abs(u0).max()


@pytest.mark.parametrize("brick", [None, (3, 4, 4)])
def test_500k_tet10_assembly_properties(brick):
    """configs[4] element (10-node, here 5 points) on 497 664 elements, lexicographic and brick numbering: what AUTO
    runs (the state + gather kernels) agrees with the shared-state kernel and with the generic row-owner kernel, K is
    symmetric and translation-free, and the residual alone equals the residual of the fused assembly."""
    deck = mesh.bar_deck(n=24, quadratic=True, brick=brick)
    s = feahip.FeaSolver(deck)
    s.set_nodes(mesh.deformed_state(deck.nodes))
    s.create_stiffness_and_residual()
    assert s.assembly_in_use() == feahip.ASM_GATHER       # AUTO, either numbering (lexicographic ids: an element in ~7 chunks)
    assert s.update_state() == 0
    rng = np.random.default_rng(11)
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    ya, f0 = s.spmv(a), s.forces()
    t = np.zeros(s.ndof); t[2::3] = 1.0
    assert np.abs(s.spmv(t)).max() < 1e-11 * np.abs(ya).max()
    assert abs(b @ ya - a @ s.spmv(b)) < 1e-11 * abs(b @ ya)
    s.create_residual_forces()
    # f is a sum of element contributions ~100x its own size here (near equilibrium): 1e-11 of max|f|
    assert np.abs(s.forces() - f0).max() < 1e-11 * np.abs(f0).max()
    for strat in (feahip.ASM_SHARED, feahip.ASM_ROWOWNER):
        s.set_assembly(strat)
        s.create_stiffness_and_residual()
        assert np.abs(s.spmv(a) - ya).max() < 1e-12 * np.abs(ya).max()
        assert np.abs(s.forces() - f0).max() < 1e-11 * np.abs(f0).max()
    s.close()


def test_1m_lame_cylinder_a5_sharded_equals_unsharded():
    """configs[3] geometry and model at 1.2M linear tets (40 x 160 x 32 cells of
    the r-theta-z block): K symmetric and translation-free along the axis
    before BCs, reference state stress-free, and the rows one rank of 8 owns
    come out as in the unsharded assembly (to rounding: the gather chunks of a
    shard start at its first row, so which blocks are written as transposes of
    their mirror block differs; bit-identical with the staged visits)."""
    deck = mesh.cylinder_deck(40, 160, 32)
    assert len(deck.elements) == 40 * 160 * 32 * 6
    s = feahip.FeaSolver(deck)
    s.create_residual_forces()
    assert np.abs(s.forces()).max() < 1e-10
    s.update_nodes_with_bc(1.0)
    s.create_stiffness_and_residual()
    assert s.update_state() == 0
    rng = np.random.default_rng(13)
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    ya = s.spmv(a)
    t = np.zeros(s.ndof); t[2::3] = 1.0
    assert np.abs(s.spmv(t)).max() < 1e-11 * np.abs(ya).max()
    assert abs(b @ ya - a @ s.spmv(b)) < 1e-11 * abs(b @ ya)
    s.set_row_shard(5, 8)
    d = s.owned_dofs()                                     # the caller's dofs of the rank's slab
    s.create_stiffness_and_residual()
    assert np.abs(s.spmv(a)[d] - ya[d]).max() < 1e-14 * np.abs(ya).max()
    s.set_row_shard(0, 1)
    s.set_assembly(feahip.ASM_STAGED)
    s.create_stiffness_and_residual()
    yb = s.spmv(a)
    s.set_row_shard(5, 8)
    s.create_stiffness_and_residual()
    assert np.array_equal(s.spmv(a)[d], yb[d])
    s.close()


def test_tet10_27_point_rule_at_scale():
    """configs[4] element and rule (10-node, 27 Gauss points) on 165 888
    elements: what AUTO runs (state + gather kernels) = generic kernel, K symmetric, and the
    27-point K equals the 5-point K to the accuracy of the quadrature on this
    smooth state (both rules integrate the same polynomial tangent exactly
    to O(h^2) here)."""
    deck = mesh.bar_deck(dims=(12, 96, 24), quadratic=True, gauss=27)
    s = feahip.FeaSolver(deck)
    x = mesh.deformed_state(deck.nodes)
    s.set_nodes(x)
    s.create_stiffness_and_residual()
    assert s.update_state() == 0
    rng = np.random.default_rng(17)
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    ya = s.spmv(a)
    assert abs(b @ ya - a @ s.spmv(b)) < 1e-11 * abs(b @ ya)
    s.set_assembly(feahip.ASM_ROWOWNER)
    s.create_stiffness_and_residual()
    assert np.abs(s.spmv(a) - ya).max() < 1e-12 * np.abs(ya).max()
    s.close()
    d5 = mesh.bar_deck(dims=(12, 96, 24), quadratic=True, gauss=5)
    s5 = feahip.FeaSolver(d5)
    s5.set_nodes(x)
    s5.create_stiffness_and_residual()
    assert np.abs(s5.spmv(a) - ya).max() < 1e-3 * np.abs(ya).max()
    s5.close()


def test_config4_one_ranks_worth_on_one_gpu(tmp_path):
    """BASELINE.json configs[4] is 50 577 408 ten-node tetrahedra with the 27-point rule on 8 GPUs: 6 322 176 elements
    per rank.  The same element count, rule and kernels in ONE context on the one GPU the tests have (the n = 56
    block: 56 x 336 x 56 cubes): one stiffness + residual assembly, K . translation = 0, symmetry, no inverted Gauss
    point, f = 0 in the reference state; what the context holds and how long the assembly takes go to a report the
    round's profiles keep.  The matrix has 2.2e9 scalar non-zeros: the 32-bit Yale getter must refuse, not wrap
    (fea_solver.c:1491-1506 plugs exactly this element and rule in)."""
    import json
    import time
    import torch
    deck = mesh.bar_deck(n=56, quadratic=True, gauss=27)
    assert len(deck.elements) == 6322176
    free0 = torch.cuda.mem_get_info(0)[0]
    t0 = time.perf_counter()
    s = feahip.FeaSolver(deck)
    s.create_residual_forces()
    assert np.abs(s.forces()).max() < 1e-9
    s.set_nodes(mesh.deformed_state(deck.nodes))
    s.create_stiffness_and_residual(); s.sync()
    setup_s = time.perf_counter() - t0
    assert s.update_state() == 0
    assert s.assembly_in_use() == feahip.ASM_GATHER
    held = free0 - torch.cuda.mem_get_info(0)[0]
    z = s.sizes()
    assert z["nnzb"] * 9 >= 2 ** 31
    ms = s.time_kernel(0, warmup=1, iters=3)
    rng = np.random.default_rng(27)
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    ya = s.spmv(a)
    t = np.zeros(s.ndof); t[1::3] = 1.0
    assert np.abs(s.spmv(t)).max() < 1e-10 * np.abs(ya).max()
    assert abs(b @ ya - a @ s.spmv(b)) < 1e-10 * abs(b @ ya)
    with pytest.raises(feahip.FeaHipError):
        s._chk(s._lib.feahip_get_matrix_yale(s._ctx, feahip._i(np.zeros(4, dtype=np.int32)), feahip._i(np.zeros(4, dtype=np.int32)),
                                             feahip._d(np.zeros(4))))
    report = {"workload": "6 322 176 TET10 / 27 GP (one rank of eight of BASELINE configs[4]) in one context", "nodes": z["N"],
              "scalar_nnz": z["nnzb"] * 9, "device_bytes": int(held), "assembly_ms": ms, "elements_per_s": z["E"] / (ms * 1e-3),
              "setup_s": setup_s, "gather_maps": s.assembly_stats()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "config4_one_rank.json"), "w") as fh:
        json.dump(report, fh)
    assert held < 120e9                                        # well inside one MI355X (288 GB)
    s.close()


def test_10m_assembly_properties():
    """configs[2]: the headline mesh.  One assembly, checked by K.t = 0,
    symmetry and f = 0 at the reference state."""
    deck = mesh.bar_deck(n=66, recipe="clamped")
    assert len(deck.elements) == 10349856 and len(deck.nodes) == 1782133
    s = feahip.FeaSolver(deck)
    s.create_residual_forces()
    assert np.abs(s.forces()).max() < 1e-10
    s.set_nodes(mesh.deformed_state(deck.nodes))
    s.create_stiffness_and_residual()
    assert s.update_state() == 0
    assert s.assembly_in_use() == feahip.ASM_GATHER              # the lexicographic deck, numbered by the library: the gather kernel
    rng = np.random.default_rng(9)
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    yb = s.spmv(b)
    t = np.zeros(s.ndof); t[1::3] = 1.0
    assert np.abs(s.spmv(t)).max() < 1e-11 * np.abs(yb).max()
    assert abs(a @ yb - b @ s.spmv(a)) < 1e-11 * abs(a @ yb)
    # sharded = unsharded on the owned rows (spot check of one rank of 8)
    ya = s.spmv(a)
    s.set_row_shard(3, 8)
    r0, r1 = s.owned_rows()
    d = s.owned_dofs()
    s.create_stiffness_and_residual()
    assert np.abs(s.spmv(a)[d] - ya[d]).max() < 1e-14 * np.abs(ya).max()
    s.set_row_shard(0, 1)
    s.set_assembly(feahip.ASM_STAGED)
    s.create_stiffness_and_residual()
    yb = s.spmv(a)
    s.set_row_shard(3, 8)
    assert s.owned_rows() == (r0, r1)
    s.create_stiffness_and_residual()
    assert np.array_equal(s.spmv(a)[d], yb[d])
    s.close()


def test_10m_a5_cylinder_properties():
    """configs[3] at its full size on one device: the Lame cylinder (r in [1,2], z in [-8,8], exact-solutions/lame)
    as 64 x 104 x 256 cells of the (r, axial, theta) block = 10 223 616 linear tets, model A5.  One assembly in the
    bumped state, checked by size-independent properties: no inverted element, K symmetric, K . (axial translation)
    = 0, f = 0 at the reference state; and one rank of 8 assembles its rows like the unsharded run."""
    deck = mesh.cylinder_deck(64, 256, 104)
    assert len(deck.elements) == 64 * 256 * 104 * 6 == 10223616
    s = feahip.FeaSolver(deck)
    s.create_residual_forces()
    assert np.abs(s.forces()).max() < 1e-10
    s.update_nodes_with_bc(1.0)
    s.create_stiffness_and_residual()
    assert s.update_state() == 0
    rng = np.random.default_rng(23)
    a, b = rng.normal(size=s.ndof), rng.normal(size=s.ndof)
    ya = s.spmv(a)
    t = np.zeros(s.ndof); t[2::3] = 1.0
    assert np.abs(s.spmv(t)).max() < 1e-11 * np.abs(ya).max()
    assert abs(b @ ya - a @ s.spmv(b)) < 1e-11 * abs(b @ ya)
    s.set_row_shard(6, 8)
    d = s.owned_dofs()
    s.create_stiffness_and_residual()
    assert np.abs(s.spmv(a)[d] - ya[d]).max() < 1e-13 * np.abs(ya).max()
    s.close()
