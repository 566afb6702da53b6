"""Host-side logic of the product (no GPU): boundary dataclasses, array meshes, indexers, stamp
listing and the KKT -> SPD reduction algebra, checked with scipy as the calculator."""
import pickle

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import helpers as H
from padne_amd import mesh, problem, reduction, solver, structured, synthetic


# ---- problem.py ------------------------------------------------------------------------------

def test_resistor_validation_and_terminals():
    a, b = problem.NodeID(), problem.NodeID()
    with pytest.raises(ValueError):
        problem.Resistor(a=a, b=b, resistance=0.0)
    r = problem.Resistor(a=a, b=b, resistance=1.0)
    assert r.terminals == [a, b] and not r.is_source and r.extra_variable_count == 0
    v = problem.VoltageSource(p=a, n=b, voltage=1.0)
    assert v.is_source and v.extra_variable_count == 1
    c = problem.CurrentSource(f=a, t=b, current=1.0)
    assert c.is_source and c.extra_variable_count == 0
    g = problem.VoltageRegulator(v_p=a, v_n=b, s_f=a, s_t=b, voltage=1.0, gain=0.5)
    assert g.terminals == [a, b, a, b] and g.extra_variable_count == 1


def test_network_nodes_and_type_check():
    a, b, c = problem.NodeID(), problem.NodeID(), problem.NodeID()
    net = problem.Network(connections=[], elements=[problem.Resistor(a, b, 1.0), problem.CurrentSource(b, c, 1.0)])
    assert list(net.nodes) == [a, b, c] and net.has_source
    assert a != problem.NodeID()                       # identity semantics
    with pytest.raises(TypeError):
        problem.Network(connections=[], elements=[problem.Resistor("x", b, 1.0)])


def test_layer_caches_geoms():
    layer = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 1, 1)), name="F.Cu", conductance=2082.5)
    assert len(layer.geoms) == 1


# ---- mesh.py -----------------------------------------------------------------------------------

def test_from_triangle_soup_and_views():
    pts = [mesh.Point(0, 0), mesh.Point(1, 0), mesh.Point(0, 1)]
    m = mesh.Mesh.from_triangle_soup(pts, [(0, 1, 2)])
    assert len(m.vertices) == 3 and len(m.faces) == 1
    assert [int(v.i) for v in m.faces[0].vertices] == [2, 0, 1]      # (v3, v1, v2), mesh.py:320-325
    assert m.vertices[1].p == mesh.Point(1.0, 0.0)
    assert m.euler_characteristic() == 1


def test_non_manifold_rejected():
    pts = [mesh.Point(0, 0), mesh.Point(1, 0), mesh.Point(0, 1), mesh.Point(0, -1)]
    with pytest.raises(ValueError):
        mesh.Mesh.from_triangle_soup(pts, [(0, 1, 2), (0, 1, 3)])    # tests/test_mesh.py:712-733


def test_topology_fixtures():
    g = H.load_golden("star")
    m = mesh.Mesh(g["xy0"], g["tri0"])
    assert m.edge_count() == 13 and m.euler_characteristic() == 1     # tests/test_mesh.py:762-767
    g = H.load_golden("square_with_hole")
    m = mesh.Mesh(g["xy0"], g["tri0"])
    assert m.edge_count() == 16 and m.euler_characteristic() == 0     # tests/test_mesh.py:805-810


def test_forms_index_by_vertex_and_face():
    m = mesh.Mesh.from_triangle_soup([mesh.Point(0, 0), mesh.Point(1, 0), mesh.Point(0, 1)], [(0, 1, 2)])
    z = mesh.ZeroForm(m)
    z[m.vertices[2]] = 4.0
    assert z[m.vertices[2]] == 4.0 and z.values.dtype == np.float64
    other = mesh.Mesh.from_triangle_soup([mesh.Point(0, 0), mesh.Point(1, 0), mesh.Point(0, 1)], [(0, 1, 2)])
    with pytest.raises(KeyError):
        z[other.vertices[0]]
    t = mesh.TwoForm(m)
    t[m.faces[0]] = 2.0
    assert t[m.faces[0]] == 2.0
    with pytest.raises(KeyError):
        t[other.faces[0]] = 1.0


def test_mesh_and_forms_pickle():
    xy, tri = synthetic.jittered_grid(5, 4)
    m = mesh.Mesh(xy, tri)
    z = mesh.ZeroForm(m)
    z.values[:] = np.arange(len(xy))
    z2 = pickle.loads(pickle.dumps(z))
    assert np.array_equal(z2.values, z.values) and np.array_equal(z2.mesh.triangles, tri)


def test_mesher_config_validation():
    with pytest.raises(ValueError):
        mesh.Mesher.Config(minimum_angle=61)
    with pytest.raises(ValueError):
        mesh.Mesher.Config(variable_density_max_distance=0.1)
    assert mesh.Mesher.Config().is_variable_density
    with pytest.raises(mesh.MeshingException):
        mesh.Mesher().poly_to_mesh(object())


# ---- indexers ---------------------------------------------------------------------------------

def test_vertex_indexer_contiguous_blocks():
    m1 = mesh.Mesh.from_triangle_soup([mesh.Point(0, 0), mesh.Point(1, 0), mesh.Point(0, 1)], [(0, 1, 2)])
    m2 = mesh.Mesh.from_triangle_soup([mesh.Point(2, 0), mesh.Point(3, 0), mesh.Point(3, 1), mesh.Point(2, 1)],
                                      [(0, 1, 2), (0, 2, 3)])
    vi = solver.VertexIndexer.create([m1, m2])                        # tests/test_solver.py:855-918
    assert len(vi.global_index_to_vertex_index) == 7
    for k in range(3):
        assert vi.mesh_vertex_index_to_global_index[(0, k)] == k
    for k in range(4):
        assert vi.mesh_vertex_index_to_global_index[(1, k)] == 3 + k
    for g, (mi, vk) in enumerate(vi.global_index_to_vertex_index):
        assert vi.mesh_vertex_index_to_global_index[(mi, vk)] == g


def _strip_problem():
    layer = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 10, 1)), name="F.Cu", conductance=1.0)
    cl = problem.Connection(layer=layer, point=mesh.Point(0.0, 0.5))
    cr = problem.Connection(layer=layer, point=mesh.Point(10.0, 0.5))
    inner = problem.NodeID()
    net = problem.Network(connections=[cl, cr],
                          elements=[problem.VoltageSource(p=cr.node_id, n=cl.node_id, voltage=1.0),
                                    problem.Resistor(a=cr.node_id, b=inner, resistance=2.0),
                                    problem.Resistor(a=inner, b=cl.node_id, resistance=3.0)])
    prob = problem.Problem(layers=[layer], networks=[net])
    msh = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.5)).poly_to_mesh(layer.geoms[0])
    return prob, [msh], [0], cl, cr, inner, net


def test_node_indexer_snapping_and_numbering():
    prob, meshes, m2l, cl, cr, inner, net = _strip_problem()
    vi = solver.VertexIndexer.create(meshes)
    ni = solver.NodeIndexer.create(prob, meshes, m2l, vi, prob.networks)
    xy = meshes[0].points
    assert np.allclose(xy[ni.node_to_global_index[cl.node_id]], [0.0, 0.5])
    assert np.allclose(xy[ni.node_to_global_index[cr.node_id]], [10.0, 0.5])
    assert ni.node_to_global_index[inner] == len(vi) and ni.internal_node_count == 1
    assert ni.extra_source_to_global_index[net.elements[0]] == len(vi) + 1
    stamps, r = solver.allocate_system(vi, ni)
    assert stamps.shape == (len(vi) + 3, len(vi) + 3)
    assert solver.find_best_ground_node_index(prob, ni) == ni.node_to_global_index[cl.node_id]


def test_stamps_reproduce_the_reference_matrix_on_a_lil():
    """Same += sequence as solver.py:469-560 when pointed at a scipy lil_matrix."""
    g = H.load_golden("lumped_only")
    a, b = problem.NodeID(), problem.NodeID()
    net = problem.Network(connections=[], elements=[problem.CurrentSource(f=a, t=b, current=1.1),
                                                    problem.Resistor(a=a, b=b, resistance=2.2)])
    ni = solver.NodeIndexer(node_to_global_index={a: 0, b: 1}, internal_node_count=2)
    L = sp.lil_matrix((3, 3))
    r = np.zeros(3)
    solver.stamp_network_into_system(net, ni, L, r)
    solver.setup_ground_node(0, L, r)
    assert np.array_equal(L.toarray(), H.golden_L(g).toarray()) and np.array_equal(r, g["r"])
    # and the listing form emits the same triples
    S = solver.StampList(3)
    r2 = np.zeros(3)
    solver.stamp_network_into_system(net, ni, S, r2)
    solver.setup_ground_node(0, S, r2)
    rows, cols, vals = S.arrays()
    assert np.array_equal(sp.coo_matrix((vals, (rows, cols)), shape=(3, 3)).toarray(), L.toarray())
    assert [c.index for c in S.constraints] == [2]


def test_unknown_element_raises():
    class Odd(problem.BaseLumped):
        @property
        def terminals(self):
            return [problem.NodeID()]

    class Net:
        elements = [Odd()]
    with pytest.raises(NotImplementedError):
        solver.stamp_network_into_system(Net(), solver.NodeIndexer(), solver.StampList(1), np.zeros(1))


# ---- reduction algebra -------------------------------------------------------------------------

def _solve_through_the_reduction(L, r, layout, pins=None):
    """solve_system's algebra with scipy standing in for the device (matrix products and the SPD solve)."""
    red = reduction.build_reduction(layout, pins)
    N = layout.size
    free = red.index_map >= 0
    P = sp.coo_matrix((np.ones(free.sum()), (np.flatnonzero(free), red.index_map[free])), shape=(N, red.n_free)).tocsr()
    A = (-(P.T @ L @ P)).tocsc()
    assert red.n_free == 0 or abs(A - A.T).max() < 1e-9 * abs(A).max()
    b0 = red.rhs(r, L @ red.c)

    def solve(b):
        return spla.spsolve(A, b) if red.n_free else np.zeros(0)

    v = red.expand(solve(b0))
    known = {}
    if red.regulators:
        Z = [P @ solve(red.project(c.gamma)) for c in red.regulators]
        keys = [c.index for c in red.regulators]
        K = len(keys)

        def cur(i):
            vv = v + sum(i[k] * Z[k] for k in range(K))
            m = red.multipliers(r - L @ vv, dict(zip(keys, i)))
            return np.array([m[q] for q in keys])
        F0 = cur(np.zeros(K))
        J = np.stack([cur(np.eye(K)[k]) - F0 for k in range(K)], axis=1)
        i_reg = np.linalg.solve(np.eye(K) - J, F0)
        v = v + sum(i_reg[k] * Z[k] for k in range(K))
        known = dict(zip(keys, i_reg))
    for idx, val in red.multipliers(r - L @ v, known).items():
        if idx >= 0:
            v[idx] = val
    return v


@pytest.mark.parametrize("name", H.golden_names())
def test_sparse_description_of_the_reduction_is_the_dense_index_map(name):
    """padne_kkt_create builds its index map on the device from the O(#constraints) lists of the reduction:
    imap[i] = i - #{e in elim : e < i} for the potentials that are not eliminated, tied members through their
    representative.  The same formula in numpy must give the dense map of the host restatement, and ``index_of`` (single
    lookups without the dense map) must agree with it; the known part c lives on the listed unknowns only."""
    g = H.load_golden(name)
    L = H.golden_L(g)
    layout = reduction.infer_layout(L, g["r"])
    red = reduction.build_reduction(layout)
    N, n_pot = layout.size, layout.n_potential
    assert red._index_map is None                                       # nothing dense was made to build it
    assert np.all(np.diff(red.elim) > 0) and (len(red.elim) == 0 or red.elim[-1] < n_pot)
    assert red.n_free == n_pot - len(red.elim)
    i = np.arange(N)
    pos = np.searchsorted(red.elim, i)
    hit = (pos < len(red.elim)) & (red.elim[np.minimum(pos, max(len(red.elim) - 1, 0))] == i) if len(red.elim) else np.zeros(N, bool)
    dev = np.where((i < n_pot) & ~hit, i - pos, -1).astype(np.int32)
    for member, rep in red.tied:
        assert member > rep and dev[rep] >= 0
        dev[member] = dev[rep]
    lookups = [red.index_of(x) for x in range(N)]
    assert np.array_equal(dev, red.index_map) and lookups == list(red.index_map)
    c = np.zeros(N)
    for x, val in red.known.items():
        c[x] = val
    assert np.array_equal(c, red.c) and all(val != 0.0 for val in red.known.values())


@pytest.mark.parametrize("name", ["unit_square", "two_layer_via", "voltage_source", "glue_sources", "regulator",
                                  "lumped_only", "strip20"])
def test_reduction_reproduces_the_direct_solve(name):
    """The KKT -> SPD rewriting is exact algebra: with scipy doing the matrix work, the reduced
    solve + multiplier recovery must reproduce the reference's v (incl. currents)."""
    g = H.load_golden(name)
    L = H.golden_L(g)
    r = g["r"]
    layout = reduction.infer_layout(L, r)
    meshes, sig, stamps, r2, n_pot = H.product_system(g)
    assert layout.n_potential == n_pot
    assert sorted(c.index for c in layout.constraints) == sorted(c.index for c in stamps.constraints)
    by_idx = {c.index: c for c in stamps.constraints}
    for c in layout.constraints:
        e = by_idx[c.index]
        assert (c.p, c.n) == (e.p, e.n) and c.gamma == pytest.approx(e.gamma)
    v = _solve_through_the_reduction(L, r, layout)
    scale = np.abs(g["v"]).max()
    tol = 1e-7 if name == "regulator" else 1e-9
    np.testing.assert_allclose(v, g["v"], rtol=0, atol=tol * scale)
    assert np.linalg.norm(L @ v - r) < 1e-8 * max(scale, 1)


@pytest.mark.parametrize("elements", [
    # an internal node (1) between two voltage sources in series: its row is two +-1 entries on a zero diagonal
    [("V", 1, 0, 1.0, 4), ("V", 2, 1, 2.0, 5), ("R", 2, 3, 4.0), ("R", 3, 0, 2.0)],
    # a current source feeding the terminal (1) of a voltage source: that row is a single +1 on a zero diagonal
    [("I", 0, 1, 0.5), ("V", 1, 0, 1.0, 4), ("R", 2, 0, 4.0), ("R", 2, 3, 1.0), ("R", 3, 0, 1.0)],
])
def test_layout_inference_does_not_mistake_source_only_nodes_for_multipliers(elements):
    """Bare-matrix entry of solve_system: potential rows that touch only source terminals look like constraint rows
    (zero diagonal, +-1 entries) but are not part of the trailing multiplier block."""
    from oracle import padne_oracle as O
    L, r = O.assemble_system([], 4, elements, 0)
    layout = reduction.infer_layout(L, r)
    n_extra = sum(1 for e in elements if e[0] == "V")
    assert layout.n_potential == 4 and len(layout.constraints) == n_extra + 1
    assert reduction.infer_layout(L, r, n_potential=4).n_potential == 4
    with pytest.raises(reduction.SingularSystemError):
        reduction.infer_layout(L, r, n_potential=3)
    v = _solve_through_the_reduction(L, r, layout)
    v_ref, _, _ = O.solve_system(L, r)
    np.testing.assert_allclose(v, v_ref, rtol=0, atol=1e-12 * np.abs(v_ref).max())


def test_floating_copper_is_found_and_pinned():
    """An unterminated current loop (tests/test_solver.py:1829-1833): a source pushes current from the grounded island
    into an island nothing else ties to it.  The reference's matrix is singular; here the floating island is found from
    the meshes and the lumped links alone, held at 0 V at its first vertex, and the ground current is the net current
    injected into the grounded island."""
    from oracle import padne_oracle as O
    xy0, tri0 = synthetic.jittered_grid(7, 6, seed=1)
    xy1, tri1 = synthetic.jittered_grid(5, 5, seed=2)
    n0, n1 = len(xy0), len(xy1)
    els = [("I", 10, n0 + 7, 0.75), ("R", 3, n0 + n1, 2.0)]            # + a dangling resistor to an internal node
    L, r = O.assemble_system([(xy0, tri0, 2082.5), (xy1, tri1, 2082.5)], 1, els, 0)
    layout = reduction.infer_layout(L, r)
    offs = np.array([0, n0, n0 + n1])
    pins = reduction.floating_component_pins(layout.n_potential, 0, layout.constraints, mesh_offsets=offs,
                                             links=np.array([[3, n0 + n1]]))
    assert pins == [n0]
    assert reduction.floating_component_pins(layout.n_potential, 0, layout.constraints, matrix=L) == [n0]
    # tie the islands with a resistor, or with a voltage source: nothing floats any more
    assert reduction.floating_component_pins(layout.n_potential, 0, layout.constraints, mesh_offsets=offs,
                                             links=np.array([[3, n0 + n1], [5, n0 + 2]])) == []
    tie = [reduction.Constraint(index=layout.size, p=n0 + 4, n=2, value=1.0)]
    assert reduction.floating_component_pins(layout.n_potential, 0, layout.constraints + tie, mesh_offsets=offs,
                                             links=np.zeros((0, 2), int)) == [n0 + n1]
    v = _solve_through_the_reduction(L, r, layout, pins)
    assert abs(abs(v[-1]) - 0.75) < 1e-12 and v[n0] == 0.0 and v[0] == 0.0
    # each island on its own is a regular problem: same potentials
    La, ra = O.assemble_system([(xy0, tri0, 2082.5)], 1, [("R", 3, n0, 2.0)], 0)
    ra[10] += 0.75
    va = O.solve_system(La, ra)[0]
    np.testing.assert_allclose(v[:n0], va[:n0], rtol=0, atol=1e-12 * np.abs(va[:n0]).max())
    Lb, rb = O.assemble_system([(xy1, tri1, 2082.5)], 0, [], 0)
    rb[7] -= 0.75
    vb = O.solve_system(Lb, rb)[0]
    np.testing.assert_allclose(v[n0:n0 + n1], vb[:n1], rtol=0, atol=1e-12 * np.abs(vb[:n1]).max())


def test_voltage_source_loop_is_rejected():
    cons = [reduction.Constraint(3, 0, 1, 1.0), reduction.Constraint(4, 1, 2, 1.0), reduction.Constraint(5, 0, 2, 2.0),
            reduction.Constraint(6, 0, -1, 0.0)]
    with pytest.raises(reduction.SingularSystemError):
        reduction.build_reduction(reduction.KKTLayout(size=7, n_potential=3, constraints=cons))


def test_chain_of_sources_offsets():
    cons = [reduction.Constraint(4, 1, 0, 1.5), reduction.Constraint(5, 2, 1, 0.5), reduction.Constraint(6, 0, -1, 0.0)]
    red = reduction.build_reduction(reduction.KKTLayout(size=7, n_potential=4, constraints=cons))
    assert red.n_free == 1 and list(red.index_map[:4]) == [-1, -1, -1, 0]
    assert np.allclose(red.c[:3], [0.0, 1.5, 2.0])


def test_synthetic_generators_are_deterministic():
    a = synthetic.layered_system(2, 12, 10, via_lattice=2)
    b = synthetic.layered_system(2, 12, 10, via_lattice=2)
    assert np.array_equal(a.meshes[1][0], b.meshes[1][0]) and np.array_equal(a.resistors[0], b.resistors[0])
    assert a.n_vertices == 240 and a.meshes[0][1].shape == (2 * 11 * 9, 3)
    assert synthetic.via_ring_resistance(0.5) == pytest.approx(16 * 0.5 / (5.95e4 * np.pi * (0.185**2 - 0.15**2)))


def test_locality_ordering_is_a_permutation_that_keeps_mesh_blocks():
    g = H.load_golden("two_layer_via")
    layout = reduction.infer_layout(H.golden_L(g), g["r"])
    red = reduction.build_reduction(layout)
    before = red.index_map.copy()
    xy = np.concatenate([g["xy0"], g["xy1"]])
    offs = np.array([0, len(g["xy0"]), len(xy)])
    reduction.apply_locality_ordering(red, xy, offs)
    free = before >= 0
    assert np.array_equal(red.index_map >= 0, free)
    assert sorted(red.index_map[free]) == sorted(before[free])               # a permutation of the same labels
    n0 = len(g["xy0"])
    first_block = red.index_map[:n0][red.index_map[:n0] >= 0]
    second_block = red.index_map[n0:len(xy)][red.index_map[n0:len(xy)] >= 0]
    assert first_block.max() < second_block.min()                              # mesh blocks stay in order
    # neighbours in space get closer in index on a shuffled mesh
    rng = np.random.default_rng(0)
    pxy, ptri = synthetic.jittered_grid(80, 80)
    perm = rng.permutation(len(pxy))
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    sxy, stri = pxy[perm], inv[ptri]
    assert reduction.ordering_is_scattered(stri, len(sxy)) and not reduction.ordering_is_scattered(ptri, len(pxy))
    keys = reduction.morton_keys(sxy)
    rank = np.empty(len(sxy), dtype=np.int64); rank[np.argsort(keys, kind="stable")] = np.arange(len(sxy))
    d_before = np.abs(stri[:, 0].astype(np.int64) - stri[:, 1]).mean()
    d_after = np.abs(rank[stri[:, 0]] - rank[stri[:, 1]]).mean()
    assert d_after * 10 < d_before
    # the strip numbering is a band numbering: almost every 64-row tile touches at most 3 runs of 128 indices
    strip = reduction.strip_index(sxy, np.zeros(len(sxy), dtype=np.int64))
    order = np.lexsort((sxy[:, 0], strip))
    srank = np.empty(len(sxy), dtype=np.int64); srank[order] = np.arange(len(sxy))
    e = np.concatenate([stri[:, [0, 1]], stri[:, [1, 2]], stri[:, [2, 0]]])
    rows = np.concatenate([srank[e[:, 0]], srank[e[:, 1]], np.arange(len(sxy))])
    cols = np.concatenate([srank[e[:, 1]], srank[e[:, 0]], np.arange(len(sxy))])
    ok = 0
    n_tiles = (len(sxy) + 63) // 64
    for t in range(n_tiles):
        c = np.unique(cols[(rows >= 64 * t) & (rows < 64 * t + 64)])
        runs, i = 0, 0
        while i < len(c):
            runs += 1
            i = np.searchsorted(c, c[i] + 127, side="right")
        ok += runs <= 3
    assert ok >= 0.9 * n_tiles


def test_reduction_run_plan_matches_the_plain_formulas():
    """expand / rhs use slice copies over the contiguous runs of the index map; same result as the masks."""
    from padne_amd.reduction import Reduction
    rng = np.random.default_rng(4)
    n = 20000
    imap = np.full(n + 3, -1, dtype=np.int32)
    elim = np.zeros(n, dtype=bool)
    elim[rng.choice(n, 40, replace=False)] = True          # ground-like holes
    keep = ~elim
    imap[:n][keep] = np.arange(keep.sum(), dtype=np.int32)
    merged = np.flatnonzero(elim)[:15]                      # tied nodes numbered through another node
    imap[merged] = imap[:n][keep][rng.integers(0, keep.sum(), 15)]
    n_free = int(keep.sum())
    c = np.zeros(n + 3)
    c[merged] = rng.uniform(-1, 1, 15)
    red = Reduction.from_dense(imap, n_free, c)
    y = rng.uniform(-1, 1, n_free)
    r = rng.uniform(-1, 1, n + 3)
    free = imap >= 0
    v_ref = c.copy()
    v_ref[free] += y[imap[free]]
    b_ref = -np.bincount(imap[free], weights=r[free], minlength=n_free)
    assert len(red._plan()[1]) > 5                          # the long runs are really used
    assert np.array_equal(red.expand(y), v_ref)
    assert np.allclose(red.rhs(r, None), b_ref, rtol=0, atol=1e-15)
    # a scattered map (after the locality reordering) takes the generic path
    perm = rng.permutation(n_free).astype(np.int32)
    red2 = Reduction.from_dense(np.where(imap >= 0, perm[np.maximum(imap, 0)], -1).astype(np.int32), n_free, c)
    v2 = c.copy()
    v2[free] += y[red2.index_map[free]]
    assert np.array_equal(red2.expand(y), v2)


def test_strip_order_equals_the_three_key_sort_it_replaces():
    """_strip_order sorts one 64-bit key [mesh | strip | x] and repairs ties by index: the result is the permutation
    of lexsort((x, strip, mesh)) with non-mesh unknowns behind the meshes in their old order -- also with equal x
    coordinates, interleaved mesh ids and the fall-back branch for key fields that do not fit."""
    rng = np.random.default_rng(4)
    n = 20000
    xy = rng.uniform(0, 50, (n, 2))
    xy[500:1000, 0] = xy[1000:1500, 0]                       # equal x inside strips
    xy[3000:3400] = xy[3400:3800]                            # coincident points
    owner = np.concatenate([rng.permutation(n), -np.ones(7, dtype=np.int64)]).astype(np.int64)
    has = owner >= 0
    mesh_id = np.where(has, (owner >= n // 3).astype(np.int64) + (owner >= 2 * n // 3), 0)     # three meshes, interleaved
    order = reduction._strip_order(n + 7, has, mesh_id, xy, owner)
    strip = np.zeros(n + 7, dtype=np.int64)
    strip[has] = reduction.strip_index(xy[owner[has]], mesh_id[has])
    k_x = np.arange(n + 7, dtype=np.float64)
    k_x[has] = xy[owner[has], 0]
    want = np.lexsort((k_x, strip, np.where(has, mesh_id, 2 ** 40)))
    same_key = (np.where(has, mesh_id, 2 ** 40)[order] == np.where(has, mesh_id, 2 ** 40)[want]).all() and \
        (strip[order] == strip[want]).all()
    assert same_key and np.array_equal(np.sort(order), np.arange(n + 7))
    assert np.array_equal(order[-7:], np.arange(n, n + 7))                       # non-mesh unknowns: last, old order
    # x is quantised to 32 bits inside each mesh: the order may differ from the exact sort only between points
    # closer than that resolution, and never between equal keys (those are in index order)
    assert np.abs(k_x[order][:-7] - k_x[want][:-7]).max() <= 50 / 2.0 ** 31
    eq = (k_x[order][1:] == k_x[order][:-1]) & (strip[order][1:] == strip[order][:-1]) & (mesh_id[order][1:] == mesh_id[order][:-1])
    assert (order[1:][eq] > order[:-1][eq]).all()


# ---- the Problem-level seam with OTHER people's classes (padne hands over its own padne.problem objects) ---

def _lookalike_problem_module():
    """Classes with the reference's names and fields that do NOT derive from padne_amd.problem (what padne's
    own ``padne.problem`` objects are to this package)."""
    import types
    from dataclasses import dataclass, field

    @dataclass(frozen=True)
    class Layer:
        shape: object
        name: str
        conductance: float

    @dataclass(frozen=True, eq=False)
    class NodeID:
        pass

    @dataclass(frozen=True)
    class Connection:
        layer: Layer
        point: object
        node_id: NodeID = field(default_factory=NodeID)

    @dataclass(frozen=True)
    class Resistor:
        a: NodeID
        b: NodeID
        resistance: float
        extra_variable_count = 0
        terminals = property(lambda self: [self.a, self.b])

    @dataclass(frozen=True)
    class CurrentSource:
        f: NodeID
        t: NodeID
        current: float
        extra_variable_count = 0
        terminals = property(lambda self: [self.f, self.t])

    @dataclass(frozen=True)
    class VoltageSource:
        p: NodeID
        n: NodeID
        voltage: float
        extra_variable_count = 1
        terminals = property(lambda self: [self.p, self.n])

    @dataclass(frozen=True)
    class VoltageRegulator:
        v_p: NodeID
        v_n: NodeID
        s_f: NodeID
        s_t: NodeID
        voltage: float
        gain: float
        extra_variable_count = 1
        terminals = property(lambda self: [self.v_p, self.v_n, self.s_f, self.s_t])

    @dataclass(frozen=True)
    class Network:
        connections: list
        elements: list

        @property
        def nodes(self):
            out = {}
            for e in self.elements:
                for t in e.terminals:
                    out.setdefault(t, len(out))
            return out

    @dataclass(frozen=True)
    class Problem:
        layers: list
        networks: list

    return types.SimpleNamespace(Layer=Layer, NodeID=NodeID, Connection=Connection, Resistor=Resistor,
                                 CurrentSource=CurrentSource, VoltageSource=VoltageSource,
                                 VoltageRegulator=VoltageRegulator, Network=Network, Problem=Problem)


def _index_and_stamp(g, P):
    """The product's host half of solve_meshed (numbering + stamp listing) for a problem-level fixture."""
    prob, nodes, flat = H.build_problem(g, P)
    ms = H.problem_meshes(g)
    meshes = [mesh.Mesh(xy, tri) for xy, tri, _ in ms]
    m2l = [m[2] for m in ms]
    vindex = solver.VertexIndexer.create(meshes)
    nix = solver.NodeIndexer.create(prob, meshes, m2l, vindex, prob.networks)
    stamps, r = solver.allocate_system(vindex, nix)
    for net in prob.networks:
        solver.stamp_network_into_system(net, nix, stamps, r)
    ground = solver.find_best_ground_node_index(prob, nix)
    solver.setup_ground_node(ground, stamps, r)
    return prob, nodes, flat, nix, stamps, r, ground


def _check_against_problem_golden(g, nodes, flat, nix, stamps, r, ground):
    from oracle import padne_oracle as O
    assert [nix.node_to_global_index[nodes[int(k)]] for k in g["node_ids"]] == [int(x) for x in g["node_global"]]
    assert [nix.extra_source_to_global_index.get(el, -1) for el in flat] == [int(x) for x in g["extra_index"]]
    assert nix.internal_node_count == int(g["internal_node_count"]) and ground == int(g["ground"])
    assert np.array_equal(r, g["r"]) and stamps.shape[0] == int(g["N"])
    # listed stamps + the oracle's mesh Laplacians = the matrix the reference assembled
    N = int(g["N"])
    rows, cols, vals = stamps.arrays()
    S = sp.coo_matrix((vals, (rows, cols)), shape=(N, N)).tocsr()
    off = 0
    blocks = []
    for xy, tri, layer in H.problem_meshes(g):
        Lm = O.laplace_operator(xy, tri).tocoo()
        blocks.append(sp.coo_matrix((float(g["layer_sigma"][layer]) * Lm.data, (Lm.row + off, Lm.col + off)), shape=(N, N)))
        off += len(xy)
    total = (S + sum(blocks)).tocsr()
    total.eliminate_zeros()
    ref = H.golden_L(g)
    assert H.same_structure(total, ref)
    np.testing.assert_allclose(total.data, ref.data, rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", H.problem_golden_names())
@pytest.mark.parametrize("family", ["padne_amd", "lookalike"])
def test_problem_seam_numbering_and_stamps_vs_reference_golden(name, family):
    """Numbering (KD-tree snapping, internal nodes, extra unknowns), ground choice and the listed stamps equal what
    the reference's NodeIndexer.create / assemble_system produced for the same Problem -- also when the Problem is
    made of classes this package has never seen (duck typing by class name, not isinstance)."""
    g = H.load_golden(name)
    P = problem if family == "padne_amd" else _lookalike_problem_module()
    prob, nodes, flat, nix, stamps, r, ground = _index_and_stamp(g, P)
    _check_against_problem_golden(g, nodes, flat, nix, stamps, r, ground)
    kinds = {solver.element_kind(e) for e in flat}
    assert kinds <= {"Resistor", "CurrentSource", "VoltageSource", "VoltageRegulator"} and None not in kinds


@pytest.mark.reference
@pytest.mark.parametrize("name", H.problem_golden_names())
def test_problem_seam_with_the_reference_own_objects(name):
    """padne's solve() passes ITS OWN padne.problem / padne.mesh objects to solve_meshed.
    Build the fixture's Problem and meshes from the reference's classes (build container only) and run the product's
    numbering, stamping and Mesh.from_reference on them."""
    from oracle import ref_loader
    if not ref_loader.reference_available():
        pytest.skip("reference not mounted (GPU box)")
    ref = ref_loader.load_reference()
    g = H.load_golden(name)
    prob, nodes, flat, nix, stamps, r, ground = _index_and_stamp(g, ref.problem)
    _check_against_problem_golden(g, nodes, flat, nix, stamps, r, ground)
    assert all(type(e).__module__ == "padne.problem" for e in flat)
    # half-edge meshes of the reference -> arrays
    for xy, tri, _ in H.problem_meshes(g)[:1] if name == "problem_c1" else H.problem_meshes(g):
        rm = ref.mesh.Mesh.from_triangle_soup([ref.mesh.Point(float(x), float(y)) for x, y in xy],
                                              [tuple(int(i) for i in t) for t in tri])
        m = mesh.Mesh.from_reference(rm)
        assert np.array_equal(m.points, xy) and np.array_equal(m.triangles, tri)


def test_mesh_from_cgal_output_is_the_array_hand_off():
    """mesh.py:782-785 feeds cgal_output['vertices'] / ['triangles'] (lists of tuples, _cgal.cpp:479-488) to
    from_triangle_soup; from_cgal_output takes the same dict without building per-vertex objects."""
    xy, tri = synthetic.jittered_grid(6, 5, 0.6, seed=2)
    out = {"vertices": [(float(x), float(y)) for x, y in xy], "triangles": [tuple(int(i) for i in t) for t in tri]}
    m = mesh.Mesh.from_cgal_output(out)
    assert m.points.dtype == np.float64 and m.triangles.dtype == np.int32
    assert np.array_equal(m.points, xy) and np.array_equal(m.triangles, tri)
    same = mesh.Mesh.from_triangle_soup([mesh.Point(*p) for p in out["vertices"]], out["triangles"])
    assert np.array_equal(same.points, m.points) and np.array_equal(same.triangles, m.triangles)
    assert len(mesh.Mesh.from_cgal_output({"vertices": [], "triangles": []}).points) == 0
    with pytest.raises(IndexError):
        mesh.Mesh.from_cgal_output({"vertices": [(0, 0), (1, 0), (0, 1)], "triangles": [(0, 1, 3)]})
    with pytest.raises(ValueError, match="Non-manifold"):
        mesh.Mesh.from_cgal_output({"vertices": [(0, 0), (1, 0), (0, 1), (0, -1)], "triangles": [(0, 1, 2), (0, 1, 3)]},
                                   validate=True)

    class HalfEdgeMeshWithSoup:              # what the mesher stub of INTEGRATION.md leaves behind
        _padne_hip_soup = out
        vertices = faces = ()
    assert np.array_equal(mesh.Mesh.from_reference(HalfEdgeMeshWithSoup()).triangles, tri)


def test_rank_launcher_starts_ranks_and_reports_exit_codes_and_timeouts(tmp_path):
    """tests/rank_launcher.py starts the rank processes of the two-process GPU tests from a process that never touches the
    GPU: every rank gets RANK / WORLD_SIZE / MASTER_PORT, exit codes and output come back per rank, a rank that outlives
    the time limit is killed (and only the processes started here)."""
    import importlib.util
    import os
    import sys
    spec = importlib.util.spec_from_file_location("rank_launcher", os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_launcher.py"))
    rl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rl)
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "print('rank', r, 'of', os.environ['WORLD_SIZE'], 'port', os.environ['MASTER_PORT'], sys.argv[1], flush=True)\n"
                      "if sys.argv[1] == 'hang' and r == 1: time.sleep(60)\n"
                      "sys.exit(3 * r)\n")
    ans = rl.run({"script": str(script), "args": ["go"], "n": 3, "env": {"EXTRA": "1"}, "timeout": 60})
    assert ans["rc"] == [0, 3, 6] and not ans["timed_out"]
    assert all(f"rank {r} of 3" in ans["out"][r] for r in range(3))
    ans = rl.run({"script": str(script), "args": ["hang"], "n": 2, "timeout": 2})
    assert ans["timed_out"] and ans["rc"][0] == 0 and ans["rc"][1] != 0

