"""Synthetic decks: Kuhn-tetrahedralised blocks on the reference's bar.

The reference ships five TET10 decks of a 1 x 6 x 1 bar [0,1]x[1,7]x[0,1]
(solver-large/data/*.sexp) generated from TetGen output by
utilities/tetgenProcessor/FEATask.hs:177-207.  BASELINE.json asks for 1M-50M
element blocks, which do not exist in the reference, so they are generated
here on the same bar with the same boundary-condition recipe (end faces
y = 1 and y = 7 prescribed, 0.05-style increments along y).

Block of nx x ny x nz cubes, 6 tetrahedra per cube sharing the (0,0,0)-(1,1,1)
diagonal (one per axis permutation); odd permutations get vertices 1 and 2
swapped so every signed volume is positive.  Node numbering: x fastest, then
z, y slowest -- so a contiguous range of node ids is a slab across the long
axis (what the multi-GPU row partition cuts).  TET10 mid-side nodes sit on the
exact edge mid-points, which are precisely the remaining points of the
half-spacing grid; local order 4:(0,1) 5:(1,2) 6:(0,2) 7:(0,3) 8:(1,3) 9:(2,3)
as fea_solver.c:1287-1300.
"""
import itertools
import copy
import math

import numpy as np

from feahip import (CG, HEXAHEDRA8, MODEL_COMPRESSIBLE_NEOHOOKEAN, TETRAHEDRA4, TETRAHEDRA10, Deck)

_EDGES = [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]


def _kuhn_corner_offsets():
    tets = []
    for perm in itertools.permutations(range(3)):
        v = [np.zeros(3, dtype=np.int64)]
        for ax in perm:
            nxt = v[-1].copy()
            nxt[ax] += 1
            v.append(nxt)
        inv = sum(1 for i in range(3) for j in range(i + 1, 3) if perm[i] > perm[j])
        if inv % 2 == 1:
            v[1], v[2] = v[2], v[1]
        tets.append(np.stack(v))
    return np.stack(tets)            # [6][4][3]


def block_dims(n):
    """The BASELINE block of isotropic cubes n x 6n x n."""
    return n, 6 * n, n


def brick_numbering(gx, gy, gz, brick):
    """new id of every lexicographic node id (x fastest, then z, y slowest) when the grid is numbered brick by
    brick: bricks of bx x by x bz nodes in the same lexicographic order, nodes inside a brick likewise.  A
    contiguous id range is then a spatially compact cluster (what the assembly kernel's chunks want: fewer
    elements touch a chunk's rows) and still a slab across the long axis (what the row shard cuts).
    SURVEY.md 8(d) allows a locality numbering of the synthetic blocks; which one is used is reported."""
    bx, by, bz = brick
    ids = np.arange(gx * gy * gz, dtype=np.int64)
    i, k, j = ids % gx, (ids // gx) % gz, ids // (gx * gz)
    bi, bk, bj = i // bx, k // bz, j // by
    li, lk, lj = i % bx, k % bz, j % by
    key = np.lexsort((li, lk, lj, bi, bk, bj))           # last key is the slowest
    new_id = np.empty_like(ids)
    new_id[key] = ids
    return new_id


def kuhn_block(nx, ny, nz, quadratic=False, origin=(0.0, 1.0, 0.0), size=(1.0, 6.0, 1.0), brick=None):
    """(nodes[N][3], elements[E][4 or 10]) of the block; brick = (bx, by, bz) numbers the nodes brick by brick
    (brick_numbering) instead of lexicographically."""
    off = _kuhn_corner_offsets()
    m = 2 if quadratic else 1          # grid refinement: TET10 nodes live on the half-spacing grid
    gx, gy, gz = m * nx + 1, m * ny + 1, m * nz + 1

    def nid(i, j, k):                  # x fastest, then z, y slowest
        return (j * gz + k) * gx + i

    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    # cube order follows the node order (y slowest) so elements are sorted like their nodes
    order = np.lexsort((ci.ravel(), ck.ravel(), cj.ravel()))
    cubes = np.stack([ci.ravel()[order], cj.ravel()[order], ck.ravel()[order]], axis=1)   # [C][3]
    corner = cubes[:, None, None, :] + off[None, :, :, :]            # [C][6][4][3] in cube units
    corner = corner.reshape(-1, 4, 3) * m                             # grid units
    cols = [nid(corner[:, a, 0], corner[:, a, 1], corner[:, a, 2]) for a in range(4)]
    if quadratic:
        for (a, b) in _EDGES:
            mid = (corner[:, a, :] + corner[:, b, :]) // 2
            cols.append(nid(mid[:, 0], mid[:, 1], mid[:, 2]))
    elements = np.stack(cols, axis=1).astype(np.int32)

    j, k, i = np.meshgrid(np.arange(gy), np.arange(gz), np.arange(gx), indexing="ij")
    nodes = np.empty((gx * gy * gz, 3))
    nodes[:, 0] = origin[0] + size[0] * i.ravel() / (gx - 1)
    nodes[:, 1] = origin[1] + size[1] * j.ravel() / (gy - 1)
    nodes[:, 2] = origin[2] + size[2] * k.ravel() / (gz - 1)
    if brick is not None:
        new_id = brick_numbering(gx, gy, gz, brick)
        elements = new_id[elements].astype(np.int32)
        out = np.empty_like(nodes)
        out[new_id] = nodes
        nodes = out
    return nodes, elements


def hex_block(nx, ny, nz, origin=(0.0, 1.0, 0.0), size=(1.0, 6.0, 1.0), brick=None):
    """(nodes[N][3], elements[E][8]) of the block as one trilinear brick per cube (BASELINE.json's "synthetic hex
    meshes"; the reference has tetrahedra only).  Corner order of fea_elements.c: bottom face counter-clockwise seen
    from +t, then the top face; local axes (r, s, t) = (x, y, z), so every Jacobian is positive."""
    gx, gy, gz = nx + 1, ny + 1, nz + 1

    def nid(i, j, k):                  # x fastest, then z, y slowest (as kuhn_block)
        return (j * gz + k) * gx + i

    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    order = np.lexsort((ci.ravel(), ck.ravel(), cj.ravel()))
    i, j, k = ci.ravel()[order], cj.ravel()[order], ck.ravel()[order]
    corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    elements = np.stack([nid(i + a, j + b, k + c) for (a, b, c) in corners], axis=1).astype(np.int32)
    jj, kk, ii = np.meshgrid(np.arange(gy), np.arange(gz), np.arange(gx), indexing="ij")
    nodes = np.empty((gx * gy * gz, 3))
    nodes[:, 0] = origin[0] + size[0] * ii.ravel() / (gx - 1)
    nodes[:, 1] = origin[1] + size[1] * jj.ravel() / (gy - 1)
    nodes[:, 2] = origin[2] + size[2] * kk.ravel() / (gz - 1)
    if brick is not None:
        new_id = brick_numbering(gx, gy, gz, brick)
        elements = new_id[elements].astype(np.int32)
        out = np.empty_like(nodes)
        out[new_id] = nodes
        nodes = out
    return nodes, elements


def bar_boundary(nodes, recipe, dy):
    """Prescribed displacements on the end faces y=min and y=max.

    "clamped":  every end-face node type 7 (as data/neohook_brick.sexp).
    "uniaxial": end faces type 2 (y only, as data/*_analytical.sexp), corner
                A=(xmin,ymin,zmin) type 7 and corner B=(xmax,ymin,zmin) type 6;
                the extra z pin at B removes the rigid rotation about y that
                the reference's analytical decks leave free."""
    y = nodes[:, 1]
    lo, hi = y.min(), y.max()
    eps = 1e-9 * (hi - lo)
    bot = np.nonzero(np.abs(y - lo) < eps)[0]
    top = np.nonzero(np.abs(y - hi) < eps)[0]
    ids = np.concatenate([bot, top]).astype(np.int32)
    vals = np.zeros((len(ids), 3))
    vals[len(bot):, 1] = dy
    if recipe == "clamped":
        types = np.full(len(ids), 7, dtype=np.int32)
    elif recipe == "uniaxial":
        types = np.full(len(ids), 2, dtype=np.int32)
        xb, zb = nodes[bot, 0], nodes[bot, 2]
        a = np.argmin(xb + zb)                              # (xmin, zmin)
        b = np.argmin(-xb + zb + 2 * (xb.max() - xb.min()))  # (xmax, zmin)
        types[a] = 7
        types[b] = 6
    else:
        raise ValueError(recipe)
    return ids, types, vals


def increment_for(n):
    """Per-increment face displacement: the decks' 0.05, kept below the
    element size so the bumped face layer is not inverted."""
    return 0.05 * min(1.0, 4.0 / n)


def bar_deck(n=None, dims=None, quadratic=False, recipe="clamped", model=MODEL_COMPRESSIBLE_NEOHOOKEAN,
             gauss=None, dy=None, brick=None, hexa=False, **kw):
    nx, ny, nz = dims if dims is not None else block_dims(n)
    if hexa:
        nodes, elements = hex_block(nx, ny, nz, brick=brick)
        if dy is None:
            dy = increment_for(max(nx, nz))
        ids, types, vals = bar_boundary(nodes, recipe, dy)
        kw.setdefault("solver_type", CG)
        return Deck(model=model, parameters=[100.0, 100.0], ele_type=HEXAHEDRA8, gauss_nodes_count=8 if gauss is None else gauss,
                    nodes=nodes, elements=elements, presc_node=ids, presc_type=types, presc_values=vals, **kw)
    nodes, elements = kuhn_block(nx, ny, nz, quadratic, brick=brick)
    if dy is None:
        dy = increment_for(max(nx, nz))
    ids, types, vals = bar_boundary(nodes, recipe, dy)
    if gauss is None:
        gauss = 5 if quadratic else 1
    kw.setdefault("solver_type", CG)
    return Deck(model=model, parameters=[100.0, 100.0], ele_type=TETRAHEDRA10 if quadratic else TETRAHEDRA4,
                gauss_nodes_count=gauss, nodes=nodes, elements=elements, presc_node=ids, presc_type=types,
                presc_values=vals, **kw)


def cylinder_deck(nr, nt, nz, quadratic=False, ri=1.0, ro=2.0, zlo=-8.0, zhi=8.0, du=None,
                  model=None, gauss=None, **kw):
    """BASELINE.json configs[3]: the Lame problem's hollow cylinder (dimensions
    of exact-solutions/lame/lame_small.m:8-14: r in [1,2], z in [-8,8]) as a
    Kuhn block in (r, axial, theta) mapped to (r cos t, -r sin t, axial) --
    the sign keeps every signed volume positive -- and closed in theta.
    Node numbering: r fastest, then theta, axial slowest, so a contiguous id
    range is a slab along the axis (the row shard).  TET10 mid-side nodes are
    mapped like the vertices (curved edges).  Boundary conditions: the inner
    surface moves radially by `du` per increment (types 1|2), both end faces
    keep their axial coordinate (type 4), the outer surface is free."""
    from feahip import MODEL_A5
    if model is None:
        model = MODEL_A5
    if nt < 3:
        raise ValueError("need at least 3 cells around the cylinder")
    pn, pe = kuhn_block(nr, nz, nt, quadratic, origin=(ri, zlo, 0.0), size=(ro - ri, zhi - zlo, 2.0 * math.pi))
    m = 2 if quadratic else 1
    gx, gy, gz = m * nr + 1, m * nz + 1, m * nt + 1
    ids = np.arange(gx * gy * gz)
    i, k, j = ids % gx, (ids // gx) % gz, ids // (gx * gz)
    keep = k < gz - 1
    new_id = np.full(len(ids), -1, dtype=np.int64)
    new_id[keep] = np.arange(int(keep.sum()))
    seam = ~keep                                        # theta = 2 pi is theta = 0
    new_id[seam] = new_id[(j[seam] * gz + 0) * gx + i[seam]]
    elements = new_id[pe].astype(np.int32)
    r, ax, th = pn[keep, 0], pn[keep, 1], pn[keep, 2]
    nodes = np.stack([r * np.cos(th), -r * np.sin(th), ax], axis=1)
    if du is None:
        du = 0.05 * min(1.0, 4.0 * (ro - ri) / nr)
    inner = i[keep] == 0
    ends = (j[keep] == 0) | (j[keep] == gy - 1)
    sel = np.nonzero(inner | ends)[0].astype(np.int32)
    types = (np.where(inner[sel], 3, 0) | np.where(ends[sel], 4, 0)).astype(np.int32)
    vals = np.zeros((len(sel), 3))
    rin = np.hypot(nodes[sel, 0], nodes[sel, 1])
    vals[:, 0] = np.where(inner[sel], du * nodes[sel, 0] / rin, 0.0)
    vals[:, 1] = np.where(inner[sel], du * nodes[sel, 1] / rin, 0.0)
    if gauss is None:
        gauss = 5 if quadratic else 1
    kw.setdefault("solver_type", CG)
    return Deck(model=model, parameters=[100.0, 100.0], ele_type=TETRAHEDRA10 if quadratic else TETRAHEDRA4,
                gauss_nodes_count=gauss, nodes=nodes, elements=elements, presc_node=sel, presc_type=types,
                presc_values=vals, **kw)


def neohookean_lateral_stretch(k1, lam=100.0, mu=100.0):
    """k2 of the uniaxial state: root of mu(k2^2-1)+lam ln(k1 k2^2) = 0
    (exact-solutions/uniaxial/uniaxial_neohookean_bonet.m:20-26)."""
    k2 = 1.0
    for _ in range(60):
        f = mu * (k2 * k2 - 1) + lam * math.log(k1 * k2 * k2)
        df = 2 * mu * k2 + 2 * lam / k2
        k2 -= f / df
    return k2


def deformed_state(nodes, k1=1.1, wiggle=1e-3):
    """Current coordinates for assembly-only runs: the homogeneous uniaxial
    state at stretch k1 plus a smooth deterministic perturbation so F differs
    per element."""
    k2 = neohookean_lateral_stretch(k1)
    A = nodes.min(axis=0)
    x = A + (nodes - A) * np.array([k2, k1, k2])
    X = nodes
    u = wiggle * np.sin(2 * np.pi * X[:, 0]) * np.sin(np.pi * (X[:, 1] - 1.0) / 3.0) * np.sin(2 * np.pi * X[:, 2])
    return x + u[:, None]


def jitter_permute(deck, amp=0.2, seed=4, spacing=None):
    """The same mesh off the lattice: every node displaced by a deterministic pseudo-random vector of at most `amp`
    spacings per axis (amp <= 0.2 keeps every Kuhn tetrahedron's volume positive) and the node ids randomly
    permuted -- what a mesh generator's output looks like to the library: no structure in the ids, no exact lattice
    in the coordinates.  Prescribed displacements follow their nodes."""
    rng = np.random.default_rng(seed)
    N = len(deck.nodes)
    if spacing is None:
        e = deck.elements[: min(len(deck.elements), 4096)]
        d = np.abs(deck.nodes[e[:, 0]] - deck.nodes[e[:, 1]])
        spacing = np.array([np.median(d[:, k][d[:, k] > 0]) if (d[:, k] > 0).any() else 0.0 for k in range(3)])
        spacing[spacing == 0] = spacing[spacing > 0].min()
    nodes = deck.nodes + amp * spacing * (2.0 * rng.random((N, 3)) - 1.0)
    new_id = rng.permutation(N)                          # new id of old node
    out = copy.copy(deck)
    out.nodes = np.empty_like(nodes)
    out.nodes[new_id] = nodes
    out.elements = np.ascontiguousarray(new_id[deck.elements].astype(np.int32))
    out.presc_node = np.ascontiguousarray(new_id[deck.presc_node].astype(np.int32)) if len(deck.presc_node) else deck.presc_node
    return out


def tiled(deck, copies):
    """cx x cy x cz translated copies of a deck's mesh side by side (separate bodies: no shared nodes), the copies'
    nodes and elements appended in order -- a way to time a mesh generator's mesh at a size the generator's own
    output does not have.  Prescribed nodes are replicated with their values."""
    cx, cy, cz = copies
    ext = deck.nodes.max(axis=0) - deck.nodes.min(axis=0)
    N = len(deck.nodes)
    nodes, elements, pn, pt, pv = [], [], [], [], []
    k = 0
    for iz in range(cz):
        for iy in range(cy):
            for ix in range(cx):
                nodes.append(deck.nodes + 1.05 * ext * np.array([ix, iy, iz]))
                elements.append(deck.elements + k * N)
                if len(deck.presc_node):
                    pn.append(deck.presc_node + k * N); pt.append(deck.presc_type); pv.append(deck.presc_values)
                k += 1
    out = copy.copy(deck)
    out.nodes = np.ascontiguousarray(np.concatenate(nodes))
    out.elements = np.ascontiguousarray(np.concatenate(elements).astype(np.int32))
    if pn:
        out.presc_node = np.ascontiguousarray(np.concatenate(pn).astype(np.int32))
        out.presc_type = np.ascontiguousarray(np.concatenate(pt).astype(np.int32))
        out.presc_values = np.ascontiguousarray(np.concatenate(pv))
    return out


def corner_tets(deck):
    """The linear tetrahedra on the corner nodes of a deck of 10-node tetrahedra (local nodes 0-3), the mid-edge nodes
    dropped and the rest renumbered in order: an unstructured LINEAR-tet mesh out of the reference's TetGen decks, which
    only come with 10-node elements.  Prescribed nodes that are mid-edge nodes go with them."""
    from feahip import TETRAHEDRA4
    corners = np.unique(deck.elements[:, :4])
    new_id = np.full(len(deck.nodes), -1, dtype=np.int64)
    new_id[corners] = np.arange(len(corners))
    out = copy.copy(deck)
    out.nodes = np.ascontiguousarray(deck.nodes[corners])
    out.elements = np.ascontiguousarray(new_id[deck.elements[:, :4]].astype(np.int32))
    out.nodes_per_element = 4
    out.ele_type = TETRAHEDRA4
    out.gauss_nodes_count = 1
    if len(deck.presc_node):
        keep = new_id[deck.presc_node] >= 0
        out.presc_node = np.ascontiguousarray(new_id[deck.presc_node[keep]].astype(np.int32))
        out.presc_type = np.ascontiguousarray(deck.presc_type[keep])
        out.presc_values = np.ascontiguousarray(deck.presc_values[keep])
    return out
