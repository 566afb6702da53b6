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
