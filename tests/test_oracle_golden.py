"""Pin the CPU oracle against golden vectors produced by the real reference (tests/golden/*.npz).

Runs without a GPU.  Bar: bitwise for everything except (a) the Laplacian diagonal, which the
reference sums in half-edge orbit order and the oracle in column order (rtol 1e-14), and (b) the
direct solve, compared at 1e-10 of the largest potential."""
import numpy as np
import pytest
import scipy.sparse as sp

import helpers as H
from oracle import padne_oracle as O

NAMES = H.golden_names()
# the regulator case couples two islands through 100 kOhm next to a 2 kS sheet (as the reference's own
# test does, tests/test_solver.py:122-125): condition ~1e9, so two LU runs that differ by one ulp in the
# diagonal agree only to ~1e-9 of the largest unknown.
SOLVE_TOL = {"regulator": 1e-8}


def test_fixtures_present():
    assert len(NAMES) >= 10


@pytest.mark.parametrize("name", NAMES)
def test_laplace_operator_matches_reference(name):
    g = H.load_golden(name)
    for i, (xy, tri, _, _) in enumerate(H.meshes_of(g)):
        ref = H.golden_lap(g, i)
        got = O.laplace_operator(xy, tri).tocsr()
        assert H.same_structure(ref, got)
        ro, rd = H.offdiag_and_diag(ref)
        go, gd = H.offdiag_and_diag(got)
        assert np.array_equal(ro.data, go.data), "off-diagonal cotangent weights must be bit-identical"
        np.testing.assert_allclose(gd, rd, rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", NAMES)
def test_assembled_system_matches_reference(name):
    g = H.load_golden(name)
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in H.meshes_of(g)], int(g["n_internal"]), H.elements_of(g),
                             int(g["ground"]))
    ref = H.golden_L(g)
    assert H.same_structure(ref, L)
    ro, rd = H.offdiag_and_diag(ref)
    go, gd = H.offdiag_and_diag(L)
    assert np.array_equal(ro.data, go.data)
    np.testing.assert_allclose(gd, rd, rtol=1e-14, atol=0)
    assert np.array_equal(r, g["r"])


@pytest.mark.parametrize("name", NAMES)
def test_solve_and_diagnostics_match_reference(name):
    g = H.load_golden(name)
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in H.meshes_of(g)], int(g["n_internal"]), H.elements_of(g),
                             int(g["ground"]))
    v, gc, res = O.solve_system(L, r)
    scale = np.abs(g["v"]).max()
    np.testing.assert_allclose(v, g["v"], rtol=0, atol=SOLVE_TOL.get(name, 1e-10) * scale)
    assert abs(gc - float(g["ground_node_current"])) <= SOLVE_TOL.get(name, 1e-10) * max(scale, 1.0)
    assert res < 1e-9                                    # tests/test_solver.py:2083-2089


@pytest.mark.parametrize("name", NAMES)
def test_power_density_matches_reference(name):
    g = H.load_golden(name)
    for i, (xy, tri, sigma, _) in enumerate(H.meshes_of(g)):
        got = O.power_density(xy, tri, g[f"pot{i}"], sigma)
        assert np.array_equal(got, g[f"pow{i}"]), "same operations in the same order: bit-identical"


@pytest.mark.parametrize("name", NAMES)
def test_potentials_are_slices_of_v(name):
    g = H.load_golden(name)
    off = 0
    for i, (xy, _, _, _) in enumerate(H.meshes_of(g)):
        assert np.array_equal(g[f"pot{i}"], g["v"][off:off + len(xy)])
        off += len(xy)


# ---- Problem-level fixtures: the reference's whole post-meshing path (VertexIndexer, NodeIndexer.create with its
# KD-tree snapping, assemble_system with find_best_ground_node_index, solve_system, produce_layer_solutions) ----------

PNAMES = H.problem_golden_names()


def _oracle_problem(g):
    meshes = H.problem_meshes(g)
    nets = H.problem_networks(g)
    off = np.concatenate([[0], np.cumsum([len(m[0]) for m in meshes])])
    layer_points, layer_gidx = {}, {}
    for li in range(len(g["layer_sigma"])):
        blocks = [(m[0], np.arange(len(m[0])) + off[i]) for i, m in enumerate(meshes) if m[2] == li]
        if blocks:
            layer_points[li] = np.concatenate([b[0] for b in blocks])
            layer_gidx[li] = np.concatenate([b[1] for b in blocks])
    n2g, extra, internal = O.node_indexer_create(layer_points, layer_gidx, int(off[-1]), nets)
    ground = O.find_best_ground_node_index(nets, n2g)
    els = O.globalise_elements(nets, n2g, extra)
    L, r = O.assemble_system([(m[0], m[1], float(g["layer_sigma"][m[2]])) for m in meshes], internal, els, ground)
    return n2g, extra, internal, ground, L, r


def test_problem_fixtures_present():
    assert "problem_c1" in PNAMES            # config C1 of BASELINE.json


@pytest.mark.parametrize("name", PNAMES)
def test_problem_level_numbering_and_system_match_reference(name):
    g = H.load_golden(name)
    n2g, extra, internal, ground, L, r = _oracle_problem(g)
    assert [n2g[int(k)] for k in g["node_ids"]] == [int(x) for x in g["node_global"]]
    assert extra == [int(x) for x in g["extra_index"]]
    assert internal == int(g["internal_node_count"]) and ground == int(g["ground"])
    ref = H.golden_L(g)
    assert H.same_structure(ref, L)
    ro, rd = H.offdiag_and_diag(ref)
    go, gd = H.offdiag_and_diag(L)
    assert np.array_equal(ro.data, go.data)
    np.testing.assert_allclose(gd, rd, rtol=1e-14, atol=0)
    assert np.array_equal(r, g["r"])


@pytest.mark.parametrize("name", PNAMES)
def test_problem_level_solution_matches_reference(name):
    g = H.load_golden(name)
    *_, L, r = _oracle_problem(g)
    v, gc, res = O.solve_system(L, r)
    n_vert = sum(len(m[0]) for m in H.problem_meshes(g))
    scale = np.abs(g["v"][:n_vert]).max()
    assert np.abs(v[:n_vert] - g["v"][:n_vert]).max() <= 1e-8 * scale          # the north-star bar, potentials
    cur = np.abs(g["v"][n_vert:]).max()
    assert np.abs(v[n_vert:] - g["v"][n_vert:]).max() <= 1e-8 * max(cur, 1.0)    # internal nodes, source currents
    assert res < 1e-9 and float(g["residual_norm"]) < 1e-9                       # tests/test_solver.py:2083-2089
    off = 0
    for i, (xy, tri, layer) in enumerate(H.problem_meshes(g)):
        assert np.array_equal(g[f"pot{i}"], g["v"][off:off + len(xy)])
        assert np.array_equal(O.power_density(xy, tri, g[f"pot{i}"], float(g["layer_sigma"][layer])), g[f"pow{i}"])
        off += len(xy)


def test_c1_is_the_via_tht_4layer_plumbing_case():
    """Shape of config C1: 4 layers of 2082.5 S, 16-resistor rings per adjacent layer pair with
    R = 16 R_via, three 0.1 Ohm resistors, one 1 V source whose negative terminal is the ground."""
    g = H.load_golden("problem_c1")
    nets = H.problem_networks(g)
    assert list(g["layer_sigma"]) == [2082.5] * 4
    rings = [n for n in nets if len(n["elements"]) == 16]
    assert len(rings) == 6 * 3 and all(e[0] == "R" for n in rings for e in n["elements"])
    for n in rings:
        la, lb = n["connections"][0][0], n["connections"][1][0]
        assert lb == la + 1 and len({e[3] for e in n["elements"]}) == 1
    lumped = [e for n in nets if len(n["elements"]) == 1 for e in n["elements"]]
    assert sorted(e[0] for e in lumped) == ["R", "R", "R", "V"]
    assert [e[3] for e in lumped if e[0] == "R"] == [0.1] * 3 and [e[3] for e in lumped if e[0] == "V"] == [1.0]
    n_vert = sum(len(m[0]) for m in H.problem_meshes(g))
    vsrc = next(n for n in nets if n["elements"][0][0] == "V")
    ids = {int(k): int(v) for k, v in zip(g["node_ids"], g["node_global"])}
    p, n = ids[vsrc["elements"][0][1]], ids[vsrc["elements"][0][2]]
    assert int(g["ground"]) == n
    assert abs((g["v"][p] - g["v"][n]) - 1.0) < 1e-3                             # tests/test_solver.py:1205
    assert g["v"][n_vert:-1].shape == (1,)                                       # one source-current unknown
