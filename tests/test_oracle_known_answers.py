"""Known-answer tests of the reference's own test-suite, restated for the CPU oracle (no GPU).

Sources: tests/test_solver.py:65-147 (network stamps), :776-852 (unit-square Laplacian),
:757-773 (Laplacian invariants), :923-971 (power density), :1042-1112 (triangle gradient);
tests/test_mesh.py:712-733 (non-manifold rejection)."""
import numpy as np
import pytest

from oracle import padne_oracle as O
from padne_amd import synthetic


def solve_network(n_nodes, elements):
    """test_solver.py:46-63: stamp, drop node 0, dense solve."""
    n_extra = sum(1 for e in elements if e[0] in ("V", "REG"))
    L, r = O.assemble_system([], n_nodes, elements, 0)
    N = n_nodes + n_extra
    Ld = L.toarray()[:N, :N]
    v = np.linalg.solve(Ld[1:, 1:], r[1:N])
    return np.concatenate([[0.0], v])


def test_current_into_resistor():
    v = solve_network(2, [("I", 0, 1, 1.1), ("R", 0, 1, 2.2)])
    assert v[1] - v[0] == pytest.approx(1.1 * 2.2, abs=1e-6)


def test_voltage_into_resistor():
    v = solve_network(2, [("V", 0, 1, 3.3, 2), ("R", 0, 1, 2.2)])
    assert v[0] - v[1] == pytest.approx(3.3, abs=1e-6)
    assert v[2] == pytest.approx(3.3 / 2.2, abs=1e-6)


def test_voltage_regulator():
    # nodes: 0=p 1=n 2=f 3=t ; extra unknown 4
    els = [("R", 2, 3, 1.4), ("R", 0, 1, 2.2), ("R", 3, 1, 100000), ("REG", 0, 1, 2, 3, 3.3, 0.3, 4)]
    v = solve_network(4, els)
    assert v[0] - v[1] == pytest.approx(3.3, abs=1e-6)
    assert v[4] == pytest.approx(3.3 / 2.2, abs=1e-6)
    assert v[2] - v[3] == pytest.approx(3.3 / 2.2 * 0.3 * 1.4, abs=1e-6)


def test_unit_square_laplacian():
    pts = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]], float)
    tri = np.array([[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]])
    L = O.laplace_operator(pts, tri).toarray()
    expected = np.zeros((5, 5))
    expected[4, :4] = 1
    expected[4, 4] = -4
    for i in range(4):
        expected[i, 4] = 1
        expected[i, i] = -1
    np.testing.assert_allclose(L, expected, rtol=1e-5, atol=1e-5)


def test_laplacian_invariants():
    xy, tri = synthetic.jittered_grid(9, 7, 0.6, seed=11)
    L = O.laplace_operator(xy, tri).toarray()
    assert np.all(np.diag(L) < 0)
    off = L - np.diag(np.diag(L))
    assert np.all(off >= 0)
    assert np.array_equal(L, L.T)
    assert np.all(np.abs(L.sum(axis=1)) < 1e-5)


def test_zero_weight_edges_are_dropped():
    # right-angle corners: the hypotenuse-opposite... the cell diagonal gets cot(90 deg) = 0 from both sides
    xy, tri = synthetic.jittered_grid(4, 4, 1.0, jitter=0.0)
    L = O.laplace_operator(xy, tri).tocsr()
    assert np.all(L.data != 0)
    assert L[5, 10] == 0 and L[5, 6] > 0


def test_non_manifold_rejected():
    tri = np.array([[0, 1, 2], [0, 1, 3]])
    with pytest.raises(ValueError):
        O.check_manifold(4, tri)
    with pytest.raises(ValueError):
        O.laplace_operator(np.array([[0, 0], [1, 0], [0, 1], [0, -1]], float), tri)


def test_empty_mesh():
    L = O.laplace_operator(np.zeros((0, 2)), np.zeros((0, 3), int))
    assert L.shape == (0, 0) and L.nnz == 0


@pytest.mark.parametrize("values,expected", [((5.0, 5.0, 5.0), (0.0, 0.0)), ((0.0, 1.0, 0.0), (1.0, 0.0)),
                                             ((0.0, 0.0, 1.0), (0.0, 1.0)), ((0.0, 1.0, 1.0), (1.0, 1.0))])
def test_triangle_gradient(values, expected):
    p = [np.array([0.0, 0.0]), np.array([1.0, 0.0]), np.array([0.0, 1.0])]
    gx, gy = O.triangle_gradient(p[0], p[1], p[2], *values)
    assert gx == pytest.approx(expected[0], abs=1e-10)
    assert gy == pytest.approx(expected[1], abs=1e-10)


def test_power_density_constant_and_linear():
    xy = np.array([[0, 0], [1, 0], [0, 1]], float)
    tri = np.array([[0, 1, 2]])
    assert O.power_density(xy, tri, np.full(3, 5.0), 1.0)[0] == pytest.approx(0.0, abs=1e-10)
    assert O.power_density(xy, tri, xy[:, 0], 2.0)[0] == pytest.approx(2.0, abs=1e-6)


def test_linear_strip_potential():
    """test_solver.py:461-595: ideal source across a 10x1 strip -> V linear in x."""
    nx, ny = 41, 5
    xy, tri = synthetic.jittered_grid(nx, ny, 0.25, seed=2, jitter=0.1)
    n = nx * ny
    left = [j * nx for j in range(ny)]
    right = [j * nx + nx - 1 for j in range(ny)]
    els = [("V", right[j], left[j], 1.0, n + j) for j in range(ny)]
    for j in range(1, ny):                                  # tie the left pads together (0 V glue)
        els.append(("V", left[j], left[0], 0.0, n + ny + j - 1))
    L, r = O.assemble_system([(xy, tri, 1.0)], 0, els, left[0])
    v, gc, res = O.solve_system(L, r)
    x = xy[:, 0]
    assert np.all(np.abs(v[:n] - x / x.max()) < 0.05)
    assert abs(gc) < 1e-9 and res < 1e-9


def test_coaxial_potential():
    """test_solver.py:597-751: annulus r in [1, 9], V = ln(9/r)/ln 9 within 0.03."""
    xy, tri = synthetic.annulus_mesh(1.0, 9.0, 33, 96)
    n = len(xy)
    inner = list(range(96))
    outer = list(range(n - 96, n))
    els = [("V", inner[0], outer[0], 1.0, n)]
    k = n + 1
    for a in inner[1:]:
        els.append(("V", a, inner[0], 0.0, k)); k += 1
    for a in outer[1:]:
        els.append(("V", a, outer[0], 0.0, k)); k += 1
    L, r = O.assemble_system([(xy, tri, 1.0)], 0, els, outer[0])
    v, gc, res = O.solve_system(L, r)
    rad = np.hypot(xy[:, 0], xy[:, 1])
    assert np.all(np.abs(v[:n] - np.log(9 / rad) / np.log(9)) < 0.03)
    ring = v[96 * 10:96 * 11]
    assert ring.max() - ring.min() < 1e-3


def test_pcg_restatement_agrees_with_direct_solve():
    sysm = synthetic.layered_system(2, 24, 24, via_lattice=3)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    v, _, _ = O.solve_system(L, r)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr()
    x, it, rel = O.pcg_jacobi(A, -r[1:n], rtol=1e-13)
    assert rel <= 1e-13
    assert np.abs(x - v[1:n]).max() <= 1e-9 * np.abs(v[:n]).max()
