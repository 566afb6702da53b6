"""Parity of the HIP path against the oracle and the reference's golden vectors (needs an MI355X).

Everything here goes through the C ABI (padne_amd._hip -> libpadne_hip.so).  Bars:
  * assembled matrices: identical structure, bit-identical values vs the oracle (same operations in
    the same order), and vs the reference goldens bit-identical except the Laplacian diagonal
    (orbit-order vs column-order sum, rtol 1e-14);
  * SpMV: bit-identical to a sequential CSR product (scipy);
  * power density / gradients: bit-identical;
  * potentials: <= 1e-8 relative to the reference's direct solve (north_star), in practice ~1e-12.
"""
import pickle
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import helpers as H
from oracle import padne_oracle as O
from padne_amd import _hip, mesh, problem, solver, structured, synthetic

pytestmark = pytest.mark.gpu
NAMES = H.golden_names()
REL_TOL = 1e-8          # BASELINE.json north_star: potentials within 1e-8 relative


def flat(meshes_spec):
    xy = np.concatenate([m[0] for m in meshes_spec]) if meshes_spec else np.zeros((0, 2))
    tri = np.concatenate([m[1] for m in meshes_spec]) if meshes_spec else np.zeros((0, 3), np.int32)
    mvo = np.concatenate([[0], np.cumsum([len(m[0]) for m in meshes_spec])]).astype(np.int64)
    mto = np.concatenate([[0], np.cumsum([len(m[1]) for m in meshes_spec])]).astype(np.int64)
    sig = np.array([m[2] for m in meshes_spec], dtype=float)
    return xy, tri, mvo, mto, sig


# ---- assembly ---------------------------------------------------------------------------------

@pytest.mark.parametrize("name", NAMES)
def test_laplace_operator_vs_golden_and_oracle(ctx, name):
    g = H.load_golden(name)
    for i, (xy, tri, _, _) in enumerate(H.meshes_of(g)):
        got = solver.laplace_operator(mesh.Mesh(xy, tri)).tocsr()
        orc = O.laplace_operator(xy, tri).tocsr()
        assert H.same_structure(got, orc) and np.array_equal(got.data, orc.data), "bitwise vs oracle"
        ref = H.golden_lap(g, i)
        assert H.same_structure(got, ref)
        ro, rd = H.offdiag_and_diag(ref)
        go, gd = H.offdiag_and_diag(got)
        assert np.array_equal(ro.data, go.data)
        np.testing.assert_allclose(gd, rd, rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", NAMES)
def test_assemble_system_vs_golden_and_oracle(ctx, name):
    g = H.load_golden(name)
    meshes, sig, stamps, r, n_pot = H.product_system(g)
    L = solver.assemble_from_arrays(meshes, sig, stamps, n_pot)
    got = L.tocsr()
    orc, r_o = O.assemble_system([(m[0], m[1], m[2]) for m in H.meshes_of(g)], int(g["n_internal"]),
                                 H.elements_of(g), int(g["ground"]))
    orc.sort_indices()
    assert H.same_structure(got, orc) and np.array_equal(got.data, orc.data), "bitwise vs oracle"
    assert np.array_equal(r, r_o) and np.array_equal(r, g["r"])
    ref = H.golden_L(g)
    assert H.same_structure(got, ref)
    ro, rd = H.offdiag_and_diag(ref)
    go, gd = H.offdiag_and_diag(got)
    assert np.array_equal(ro.data, go.data)
    np.testing.assert_allclose(gd, rd, rtol=1e-14, atol=0)
    L.dev.close()


@pytest.mark.parametrize("nx,ny,seed,jitter,origin", [(2, 2, 0, 0.2, (0.0, 0.0)), (7, 5, 3, 0.2, (0.0, 0.0)),
                                                   (57, 49, 2, 0.2, (110.6, 99.1)), (301, 177, 11, 0.35, (-3.5, 2.25)),
                                                   (40, 30, 5, 0.0, (0.0, 0.0))])
def test_device_mesh_generator_is_the_numpy_generator_bit_for_bit(ctx, nx, ny, seed, jitter, origin):
    """padne_generate_grid_mesh (the structured stand-in for the CGAL mesher, off the host): same vertices,
    same jitter (numpy's PCG64 stream evaluated per vertex by jump-ahead), same triangles as synthetic.jittered_grid."""
    xy_d, tri_d = ctx.generate_grid_mesh(nx, ny, 0.6, seed=seed, jitter=jitter, origin=origin)
    xy, tri = synthetic.jittered_grid(nx, ny, 0.6, seed=seed, jitter=jitter, origin=origin)
    assert np.array_equal(xy_d.numpy(), xy) and np.array_equal(tri_d.numpy(), tri)


def test_assembly_from_device_resident_meshes(ctx):
    """The synthetic configs end to end without the mesh ever being on the host: generated on the device, assembled from
    device pointers -- the matrix is the one the host arrays give, bit for bit (and therefore the oracle's)."""
    host = synthetic.layered_system(3, 61, 47, via_lattice=4)
    dev, xy_d, tri_d = synthetic.layered_system_on_device(ctx, 3, 61, 47, via_lattice=4)
    assert dev.n_vertices == host.n_vertices and np.array_equal(dev.mesh_offsets, host.mesh_offsets)
    for got, want in zip(dev.resistors + dev.current_sources, host.resistors + host.current_sources):
        assert np.array_equal(got, want)
    assert np.array_equal(xy_d.numpy(), np.concatenate([m[0] for m in host.meshes]))
    assert np.array_equal(tri_d.numpy(), np.concatenate([m[1] for m in host.meshes]))
    xy, tri, mvo, mto, sig = flat(host.meshes)
    N = host.n_vertices + 1
    a, b, rr = host.resistors
    gg = 1 / rr
    rows = np.concatenate([np.stack([a, a, b, b], 1).reshape(-1), [N - 1, 0]])
    cols = np.concatenate([np.stack([a, b, b, a], 1).reshape(-1), [0, N - 1]])
    vals = np.concatenate([np.stack([-gg, gg, -gg, gg], 1).reshape(-1), [1.0, 1.0]])
    L_host = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals).to_scipy()
    L_dev = ctx.assemble_system(N, xy_d, tri_d, dev.mesh_offsets, dev._tri_offsets, sig, rows, cols, vals).to_scipy()
    assert H.same_structure(L_host, L_dev) and np.array_equal(L_host.data, L_dev.data)
    with pytest.raises(ValueError):
        ctx.assemble_system(N, xy_d, tri, mvo, mto, sig, rows, cols, vals)          # one on the device, one on the host


def test_assembly_is_run_to_run_bitwise_reproducible(ctx):
    sysm = synthetic.layered_system(3, 70, 50, via_lattice=5)
    xy, tri, mvo, mto, sig = flat(sysm.meshes)
    N = sysm.n_vertices + 1
    a, b, rr = sysm.resistors
    gg = 1 / rr
    rows = np.stack([a, a, b, b], 1).reshape(-1)
    cols = np.stack([a, b, b, a], 1).reshape(-1)
    vals = np.stack([-gg, gg, -gg, gg], 1).reshape(-1)
    first = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals).to_scipy()
    for _ in range(3):
        again = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals).to_scipy()
        assert np.array_equal(first.indptr, again.indptr) and np.array_equal(first.indices, again.indices)
        assert np.array_equal(first.data, again.data)


def test_assembly_edge_cases(ctx):
    e64 = np.zeros(0, np.int64)
    # empty system
    m = ctx.assemble_system(0, np.zeros((0, 2)), np.zeros((0, 3), np.int32), [0], [0], np.zeros(0), e64, e64, np.zeros(0))
    assert m.shape == (0, 0) and m.nnz == 0
    # stamps only, duplicates that cancel exactly are not stored
    m = ctx.assemble_system(3, np.zeros((0, 2)), np.zeros((0, 3), np.int32), [0], [0], np.zeros(0),
                            [0, 0, 1, 2], [1, 1, 1, 0], [2.0, -2.0, 3.0, 4.0]).to_scipy()
    assert m.nnz == 2 and m[1, 1] == 3.0 and m[2, 0] == 4.0
    # isolated vertex (no triangle): empty row, like lil += 0.0
    xy = np.array([[0, 0], [1, 0], [0, 1], [5, 5]], float)
    m = ctx.assemble_system(4, xy, np.array([[0, 1, 2]], np.int32), [0, 4], [0, 1], [1.0], e64, e64, np.zeros(0)).to_scipy()
    assert m[3].nnz == 0
    # non-manifold soup -> ValueError like Mesh.from_triangle_soup (tests/test_mesh.py:712-733)
    xy = np.array([[0, 0], [1, 0], [0, 1], [0, -1]], float)
    with pytest.raises(ValueError):
        ctx.assemble_system(4, xy, np.array([[0, 1, 2], [0, 1, 3]], np.int32), [0, 4], [0, 2], [1.0], e64, e64, np.zeros(0))
    # bow-tie vertex (two fans meeting in one vertex) is non-manifold too
    xy = np.array([[0, 0], [1, 0], [1, 1], [-1, 0], [-1, -1]], float)
    with pytest.raises(ValueError):
        ctx.assemble_system(5, xy, np.array([[0, 1, 2], [0, 3, 4]], np.int32), [0, 5], [0, 2], [1.0], e64, e64, np.zeros(0))
    # index out of range / repeated vertex
    with pytest.raises(ValueError):
        ctx.assemble_system(3, xy[:3], np.array([[0, 1, 3]], np.int32), [0, 3], [0, 1], [1.0], e64, e64, np.zeros(0))
    with pytest.raises(ValueError):
        ctx.assemble_system(3, xy[:3], np.array([[0, 1, 1]], np.int32), [0, 3], [0, 1], [1.0], e64, e64, np.zeros(0))
    with pytest.raises(ValueError):
        ctx.assemble_system(3, xy[:3], np.array([[0, 1, 2]], np.int32), [0, 3], [0, 1], [1.0], [7], [0], [1.0])


def test_long_rows_and_ragged_meshes(ctx):
    """A star node with hundreds of resistors (multi-pad terminal, kicad.py:535-556) and meshes of
    very different size in one system."""
    rng = np.random.default_rng(3)
    parts = [synthetic.jittered_grid(40, 30, seed=1), synthetic.jittered_grid(3, 2, seed=2), synthetic.jittered_grid(17, 23, seed=3)]
    ms = [(p[0], p[1], s) for p, s in zip(parts, (2082.5, 10.0, 500.0))]
    nv = sum(len(m[0]) for m in ms)
    hub = nv
    pads = rng.choice(nv, size=700, replace=False)
    els = [("R", int(p), hub, 1e-3) for p in pads] + [("I", 5, int(nv - 3), 1.0)]
    Lo, ro = O.assemble_system(ms, 1, els, 0)
    Lo.sort_indices()
    xy, tri, mvo, mto, sig = flat(ms)
    rows, cols, vals = [], [], []
    for _, a, b, rr in els[:-1]:
        g = 1 / rr
        rows += [a, a, b, b]; cols += [a, b, b, a]; vals += [-g, g, -g, g]
    N = nv + 2
    rows += [N - 1, 0]; cols += [0, N - 1]; vals += [1.0, 1.0]
    Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    got = Ld.to_scipy()
    assert H.same_structure(got, Lo) and np.array_equal(got.data, Lo.data)
    x = rng.uniform(-1, 1, N)
    assert np.array_equal(Ld.matvec(x), Lo @ x)          # the 701-entry hub row spans several LDS passes? no: one
    # a row longer than one LDS pass (2048 non-zeros)
    big = sp.csr_matrix(rng.uniform(-1, 1, (50, 6000)) * (rng.uniform(0, 1, (50, 6000)) < 0.9))
    B = ctx.csr_from_scipy(big)
    xb = rng.uniform(-1, 1, 6000)
    assert np.array_equal(B.matvec(xb), big @ xb)


# ---- SpMV -------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(1, 1), (255, 255), (256, 300), (257, 100), (5000, 5000), (100000, 100000)])
def test_spmv_bitwise_vs_sequential_csr(ctx, shape):
    rng = np.random.default_rng(shape[0])
    A = H.random_csr(shape[0], shape[1], 7, seed=shape[0] + 1)
    d = ctx.csr_from_scipy(A)
    x = rng.uniform(-1, 1, shape[1])
    assert np.array_equal(d.matvec(x), A @ x)
    assert d.spmv_bytes == 12 * A.nnz + 20 * shape[0] + 4


def test_spmv_empty_rows_and_empty_matrix(ctx):
    A = sp.csr_matrix((np.array([1.0, 2.0]), np.array([0, 3]), np.array([0, 0, 1, 1, 2, 2])), shape=(5, 4))
    d = ctx.csr_from_scipy(A)
    assert np.array_equal(d.matvec(np.arange(4.0)), A @ np.arange(4.0))
    E = ctx.csr_from_scipy(sp.csr_matrix((0, 0)))
    assert E.matvec(np.zeros(0)).shape == (0,)
    with pytest.raises(ValueError):
        d.matvec(np.zeros(3))


def test_spmv_linearity_at_full_size(ctx):
    """Size-independent property at BASELINE scale (N = 1M): A(ax + by) == a Ax + b Ay to rounding,
    and rows of the reduced Laplacian sum to the Dirichlet coupling only."""
    sysm = synthetic.config("C2")
    xy, tri, mvo, mto, sig = flat(sysm.meshes)
    e64 = np.zeros(0, np.int64)
    L = ctx.assemble_system(sysm.n_vertices, xy, tri, mvo, mto, sig, e64, e64, np.zeros(0))
    rng = np.random.default_rng(0)
    x, y = rng.uniform(-1, 1, (2, sysm.n_vertices))
    lhs = L.matvec(2.5 * x - 0.5 * y)
    rhs = 2.5 * L.matvec(x) - 0.5 * L.matvec(y)
    assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(rhs).max()
    ones = L.matvec(np.ones(sysm.n_vertices))
    assert np.abs(ones).max() <= 1e-8 * 2082.5          # constants are in the null space of the Laplacian


# ---- solve ------------------------------------------------------------------------------------

def fixture_conditioning(L, r, n_pot):
    """How far one ulp in the diagonal moves the potentials of a system (relative to the largest): the accuracy to
    which its solution is defined at all, measured with the checker's own direct solve."""
    d = L.diagonal()
    worst = 0.0
    v0 = O.solve_system(L, r)[0]
    for seed in (0, 1):
        k = np.random.default_rng(seed).integers(-1, 2, len(d))
        v1 = O.solve_system((L + sp.diags(d * (k * 2.220446049250313e-16))).tocsr(), r)[0]
        worst = max(worst, np.abs(v1[:n_pot] - v0[:n_pot]).max() / max(np.abs(v0[:n_pot]).max(), 1e-300))
    return worst, v0


@pytest.mark.parametrize("name", NAMES)
def test_solve_system_vs_reference_golden(ctx, name):
    """Potentials within 1e-8 relative of the reference's direct solve (north star); residual of the ORIGINAL system
    below the reference's absolute bar of 1e-9 (tests/test_solver.py:2083-2089).  One fixture, `regulator`, is defined
    less sharply than 1e-8: it couples two islands through 100 kOhm next to a 2 kS sheet (as the reference's own
    regulator test does, tests/test_solver.py:122-125), so a single ulp in the diagonal of L moves its potentials by
    3e-7 -- the reference's LU and the checker's LU of the same system differ by 1.9e-7 there.  That fixture is held to
    twice its own conditioning, and additionally to the checker's solve of the bit-identical device matrix."""
    g = H.load_golden(name)
    meshes, sig, stamps, r, n_pot = H.product_system(g)
    L = solver.assemble_from_arrays(meshes, sig, stamps, n_pot)
    v, info = solver.solve_system(L, r)
    scale_pot = np.abs(g["v"][:n_pot]).max()
    tol = REL_TOL
    if name == "regulator":
        sens, v_chk = fixture_conditioning(L.tocsr(), r, n_pot)
        assert 1e-8 < sens < 1e-6                               # the fixture really is that ill-conditioned
        tol = 2 * sens
        assert np.abs(v[:n_pot] - v_chk[:n_pot]).max() <= tol * scale_pot
    assert np.abs(v[:n_pot] - g["v"][:n_pot]).max() <= tol * scale_pot
    scale_cur = max(np.abs(g["v"][n_pot:]).max(), 1e-30)
    assert np.abs(v[n_pot:] - g["v"][n_pot:]).max() <= max(REL_TOL * scale_cur, 1e-9)
    assert abs(info.ground_node_current - float(g["ground_node_current"])) <= max(REL_TOL * scale_cur, 1e-9)
    assert info.residual_norm < 1e-9                             # tests/test_solver.py:2083-2089, absolute
    L.dev.close()


@pytest.mark.parametrize("name", ["voltage_source", "regulator", "two_layer_via"])
def test_solve_system_accepts_a_bare_scipy_matrix(ctx, name):
    """Drop-in seam: solve_system(L, r) with the lil_matrix the reference's own assemble_system builds."""
    g = H.load_golden(name)
    v, info = solver.solve_system(H.golden_L(g).tolil(), g["r"])
    n_pot = int(g["N"]) - 1 - sum(1 for e in H.elements_of(g) if e[0] in ("V", "REG"))
    tol = 7e-7 if name == "regulator" else REL_TOL        # twice the fixture's conditioning, see fixture_conditioning
    assert np.abs(v[:n_pot] - g["v"][:n_pot]).max() <= tol * np.abs(g["v"][:n_pot]).max()
    assert info.residual_norm < 1e-9


def test_solve_system_keeps_its_device_plan_with_the_assembled_system(ctx):
    """``solve_system`` (solver.py:767-780) on the device: the reduction plan (index map built from the O(#constraints)
    lists, A = -P^T L P, its hierarchy, the N-vectors) is kept with the assembled system by the STRUCTURE of the
    reduction.  A second right-hand side on the same system -- other source currents, another source voltage -- reuses
    it and still equals the reference's direct solve; a system with another structure (a further voltage source) gets a new
    plan; ``close`` releases it."""
    g = H.load_golden("voltage_source")
    meshes, sig, stamps, r, n_pot = H.product_system(g)
    L = solver.assemble_from_arrays(meshes, sig, stamps, n_pot)
    Lh = H.golden_L(g)
    v, info = solver.solve_system(L, r)
    assert len(L._plans) == 1
    plan = next(iter(L._plans.values()))
    scale = np.abs(g["v"]).max()
    assert np.abs(v - g["v"]).max() <= REL_TOL * scale and info.residual_norm < 1e-9
    A = plan.reduced_matrix().to_scipy()
    assert A.shape[0] == plan.n_free and abs(A - A.T).max() <= 1e-12 * abs(A).max()
    r2 = r.copy()
    iv = [c.index for c in L.layout.constraints if c.n >= 0][0]
    r2[iv] = 2.5 * r[iv] + 0.125                                     # another source voltage: c changes, the structure not
    r2[3] += 0.75
    r2[n_pot - 2] -= 0.75
    v2, info2 = solver.solve_system(L, r2)
    assert next(iter(L._plans.values())) is plan and len(L._plans) == 1
    v2_ref = O.solve_system(Lh, r2)[0]
    assert np.abs(v2 - v2_ref).max() <= REL_TOL * np.abs(v2_ref).max() and info2.residual_norm < 1e-9
    again, _ = solver.solve_system(L, r)                             # and back: nothing of the second solve sticks
    assert np.array_equal(again, v)
    L.close()
    assert not L._plans


def test_pcg_vs_direct_solve_on_layered_system(ctx):
    sysm = synthetic.layered_system(4, 150, 150, via_lattice=8)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    v_ref, gc, _ = O.solve_system(Lo, ro)
    v, info = solver.solve_system(Lo, ro)
    n = sysm.n_vertices
    assert np.abs(v[:n] - v_ref[:n]).max() <= REL_TOL * np.abs(v_ref[:n]).max()
    assert info.residual_norm < 1e-9 and abs(info.ground_node_current) < 1e-9
    assert 5 < info.iterations < 100 and info.rel_residual <= 1.1e-12        # multigrid-preconditioned


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_pcg_multiple_rhs_and_initial_guess(ctx, precond):
    xy, tri = synthetic.jittered_grid(60, 60, seed=9)
    Lo = O.laplace_operator(xy, tri).tocsr()
    A = (-2082.5 * Lo[1:, 1:]).tocsr()
    d = ctx.csr_from_scipy(A)
    rng = np.random.default_rng(2)
    B = rng.uniform(-1, 1, (3, A.shape[0]))
    res = d.solve_spd(B, rtol=1e-12, precond=precond)
    for k in range(3):
        assert np.linalg.norm(B[k] - A @ res.x[k]) <= 2e-12 * np.linalg.norm(B[k])
    warm = d.solve_spd(B[0], rtol=1e-12, x0=res.x[0], precond=precond)
    assert warm.iterations <= 2
    zero = d.solve_spd(np.zeros(A.shape[0]), precond=precond)
    assert zero.iterations == 0 and not zero.x.any()


def test_pcg_reports_breakdown_and_non_convergence(ctx):
    A = sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 1.0]]))           # indefinite
    with pytest.raises(_hip.HipError):
        ctx.csr_from_scipy(A).solve_spd(np.array([1.0, -1.0]))
    xy, tri = synthetic.jittered_grid(40, 40, seed=1)
    A = (-O.laplace_operator(xy, tri).tocsr()[1:, 1:]).tocsr()
    for precond, cap in (("jacobi", 5), ("amg", 2)):
        with pytest.raises(_hip.NotConvergedError):
            ctx.csr_from_scipy(A).solve_spd(np.ones(A.shape[0]), max_iter=cap, precond=precond)


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_solves_are_bitwise_reproducible(ctx, precond):
    xy, tri = synthetic.jittered_grid(120, 90, seed=4)
    A = (-O.laplace_operator(xy, tri).tocsr()[1:, 1:]).tocsr()
    b = np.random.default_rng(0).uniform(-1, 1, A.shape[0])
    x1 = ctx.csr_from_scipy(A).solve_spd(b, precond=precond).x       # two independently built hierarchies
    x2 = ctx.csr_from_scipy(A).solve_spd(b, precond=precond).x
    assert np.array_equal(x1, x2)


def test_converging_iteration_is_complete_when_the_gpu_is_shared(ctx):
    """the deferred ``x += alpha p`` rides on the p update, whose workgroup 0 also sets ``done``.  A
    workgroup of that launch that is dispatched late -- the card is shared with another stream or process -- must not
    take the early exit on the flag its own launch just set, or its slice of x misses the last update (the true-residual
    check would then restart: same tolerance, but timing-dependent iteration counts and bits).  The p update now reads the
    flag as the x/r update of the same iteration saw it.  Here: the same solve on a quiet card and while a second stream
    keeps every CU busy with streaming kernels; no restart, same iteration count, same bits."""
    import threading
    import torch
    xy, tri = synthetic.jittered_grid(700, 700, seed=6)               # 490 k unknowns: the vector kernels fill their grid
    A = (-O.laplace_operator(xy, tri).tocsr()[1:, 1:]).tocsr()
    b = np.zeros(A.shape[0])                                         # 1 A in, 1 A out: the right-hand side of the configs
    b[200 * 700 + 150], b[500 * 700 + 560] = 1.0, -1.0
    d = ctx.csr_from_scipy(A)
    quiet = d.solve_spd(b, precond="amg")
    assert quiet.status == _hip.OK and quiet.restarts == 0
    stop = threading.Event()
    launched = []

    def hog():
        t = torch.ones(64 * 1024 * 1024, dtype=torch.float32, device="cuda:0")     # 256 MB, streamed over and over
        n = 0
        while not stop.is_set():
            for _ in range(20):
                t.mul_(1.0000001)
            torch.cuda.synchronize()
            n += 20
        launched.append(n)

    th = threading.Thread(target=hog, daemon=True)
    th.start()
    try:
        busy = [d.solve_spd(b, precond="amg") for _ in range(6)]
    finally:
        stop.set()
        th.join(60)
    assert launched and launched[0] > 0
    for res in busy:
        assert res.status == _hip.OK and res.restarts == 0
        assert res.iterations == quiet.iterations
        assert np.array_equal(res.x, quiet.x)


# ---- multigrid preconditioner ----------------------------------------------------------------------

def layered_spd(nl=3, nx=90, ny=70, lattice=5):
    sysm = synthetic.layered_system(nl, nx, ny, via_lattice=lattice)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    return A, -ro[1:n], Lo, ro, n


def test_small_systems_are_preconditioned_by_their_dense_inverse(ctx):
    """A system the coarsest-level inverse takes as it stands (up to 2048 unknowns: the boards of a few hundred to two
    thousand vertices the reference is mostly run on) is a hierarchy of ONE level: the inverse itself is the preconditioner
    (`dense_gemv_dot`), the loop ends after a step or two where the diagonal preconditioner took hundreds -- and the
    potentials are those of scipy's direct solve.  Beyond 2048 the cycle takes over (two levels), below 33 Jacobi stays."""
    import scipy.sparse.linalg as spla
    for nl, nx, ny in ((1, 6, 5), (1, 12, 9), (2, 20, 14), (2, 36, 28), (4, 23, 22), (4, 24, 22)):
        A, b, _, _, n = layered_spd(nl, nx, ny, 4)
        d = ctx.csr_from_scipy(A)
        res = d.solve_spd(b, precond="amg", rtol=1e-12)
        jac = d.solve_spd(b, precond="jacobi", rtol=1e-12)
        ref = spla.spsolve(A.tocsc(), b)
        assert res.status == _hip.OK and res.precond_fallbacks == 0
        assert np.abs(res.x - ref).max() <= 1e-9 * np.abs(ref).max(), (A.shape, res.iterations)
        assert np.linalg.norm(A @ res.x - b) <= 2e-12 * np.linalg.norm(b)
        if A.shape[0] <= 32:
            assert res.levels == 0 and res.iterations == jac.iterations
        elif A.shape[0] <= 2048:
            assert res.levels == 1 and res.iterations <= 3 < jac.iterations, (A.shape, res.iterations, jac.iterations)
        else:
            assert res.levels >= 2 and res.iterations < 40
        # a second right-hand side on the cached inverse, and a warm start
        res2 = d.solve_spd(2.0 * b, precond="amg", rtol=1e-12, x0=res.x)
        assert res2.status == _hip.OK and np.abs(res2.x - 2.0 * ref).max() <= 1e-9 * np.abs(ref).max()
        d.close()


def test_multigrid_hierarchy_is_galerkin_and_partition_of_unity(ctx):
    A, b, _, _, _ = layered_spd()
    d = ctx.csr_from_scipy(A)
    res = d.solve_spd(b, precond="amg")
    assert res.levels >= 3 and 1.0 < res.operator_complexity < 2.0
    for lvl in range(res.levels - 1):
        Al, P, R = d.amg_level(lvl, "A"), d.amg_level(lvl, "P"), d.amg_level(lvl, "R")
        Ac = d.amg_level(lvl + 1, "A")
        assert abs(R - P.T).max() == 0.0                               # R is exactly P^T
        assert R.has_sorted_indices                                   # ... with its rows in column order (placed in order, not sorted)
        ref = (P.T @ Al @ P).tocsr()
        assert abs(Ac - ref).max() <= 1e-12 * abs(ref).max()           # Galerkin product
        assert abs(Ac - Ac.T).max() <= 1e-12 * abs(Ac).max()
        # rows whose equation sums to zero are interpolated with unit row sum (constants are reproduced)
        interior = np.abs(np.asarray(Al.sum(axis=1)).ravel()) <= 1e-9 * Al.diagonal()
        assert np.abs(np.asarray(P.sum(axis=1)).ravel()[interior] - 1.0).max() < 1e-12
        assert P.shape[1] < 0.5 * P.shape[0]


def test_mailbox_round_trips_change_nothing_but_the_waiting(ctx, switches):
    """Counts and flags of the setup reach the host through a host-coherent page the device posts into (read_back,
    mail_ticket in capi.hip) instead of memcpy + synchronise.  A context created with PADNE_NO_MAILBOX=1 takes the old
    route: same hierarchy decisions, so the same iterations and bit-identical potentials -- also for the assembly and the
    x-window plan, which use the same helper."""
    sysm = synthetic.layered_system(2, 180, 150, via_lattice=5)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    b = -ro[1:n]

    def run(c):
        d = c.csr_from_scipy(A)
        res = d.solve_spd(b, precond="amg", rtol=1e-12)
        d.close()
        return res
    with_mail = run(ctx)
    switches.set("PADNE_NO_MAILBOX", "1")
    other = _hip.Context(0)
    try:
        without = run(other)
    finally:
        other.close()
    assert with_mail.levels == without.levels >= 3 and with_mail.iterations == without.iterations
    assert np.array_equal(with_mail.x, without.x)
    assert np.linalg.norm(A @ with_mail.x - b) <= 1e-11 * np.linalg.norm(b)


@pytest.mark.parametrize("layers,nx,ny,coarse_n", [(2, 180, 150, None), (2, 180, 150, 200), (8, 240, 200, None), (3, 90, 70, 400),
                                                  (8, 400, 330, None)])
def test_dense_inverse_on_the_matrix_cores_against_the_vector_kernels(switches, layers, nx, ny, coarse_n):
    """Default form of the coarsest-level inverse: 64 pivots per launch, the rank-64 updates as v_mfma_f64_16x16x4_f64 products
    (`gj64_step`), the next pivot block inverted by workgroup 0 of the launch before.  Against the vector kernels
    (PADNE_GJ_VECTOR=1): the sums are formed in another order, so the double-precision cycle built on either inverse agrees to
    rounding times the conditioning of the coarsest operator -- not bit for bit --, the solves take the same iterations and give
    the same potentials; coarsest levels from 65 unknowns (below that the vector kernels run anyway) to ~1900, sizes that are
    and are not multiples of 64 and of the 96 x 32 tile."""
    switches.set("PADNE_AMG_F64", "1")
    if coarse_n is not None:
        switches.set("PADNE_AMG_COARSE_N", str(coarse_n))
    sysm = synthetic.layered_system(layers, nx, ny, via_lattice=5)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    b = -ro[1:n]
    probe = np.random.default_rng(7).uniform(-1, 1, n - 1)

    def run():
        c = _hip.Context(0)
        try:
            d = c.csr_from_scipy(A)
            res = d.solve_spd(b, precond="amg", rtol=1e-12)
            n_coarse = d.amg_shapes()[-1]["A"][0]
            z = d.amg_apply(probe)
            d.close()
        finally:
            c.close()
        return res, n_coarse, z
    cores, n_cores, z_cores = run()
    switches.set("PADNE_GJ_VECTOR", "1")
    vec, n_vec, z_vec = run()
    assert n_cores == n_vec and 64 < n_vec <= 2048, n_vec
    assert cores.levels == vec.levels and abs(cores.iterations - vec.iterations) <= 1
    assert np.abs(z_cores - z_vec).max() <= 1e-9 * np.abs(z_vec).max(), f"coarsest level of {n_vec} unknowns"
    assert not np.array_equal(z_cores, z_vec), "the switch changed nothing: which kernels ran?"
    assert np.abs(cores.x - vec.x).max() <= 1e-9 * np.abs(vec.x).max()
    assert np.linalg.norm(A @ cores.x - b) <= 1e-11 * np.linalg.norm(b)


def test_fused_up_leg_of_the_fine_level_is_the_same_cycle(ctx, switches):
    """Level 0 of the float cycle applies coarse correction, post-smoothing sweep and the exit product of r.z in ONE
    sparse product with W = P - c D^-1 A P (built from the merge slots of A P on the second stream).  Algebraically the
    same V(1,1) cycle as prolongation + damped Jacobi: same iteration count (one either way for float rounding), same
    potentials to the solve tolerance, against PADNE_AMG_W=none on the same matrix."""
    sysm = synthetic.layered_system(3, 260, 200, via_lattice=5)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    b = -ro[1:n]
    assert A.shape[0] >= 65536

    def run():
        d = ctx.csr_from_scipy(A)
        res = d.solve_spd(b, precond="amg", rtol=1e-12)
        d.close()
        return res
    with_w = run()
    switches.set("PADNE_AMG_W", "none")
    without = run()
    assert with_w.precond_fallbacks == 0 and without.precond_fallbacks == 0
    assert with_w.levels == without.levels >= 3
    assert abs(with_w.iterations - without.iterations) <= 1
    scale = np.abs(without.x).max()
    assert np.abs(with_w.x - without.x).max() <= 1e-9 * scale
    for res in (with_w, without):
        assert np.linalg.norm(A @ res.x - b) <= 1e-11 * np.linalg.norm(b)


def test_fused_up_leg_of_the_inner_levels_is_the_same_cycle(ctx, switches):
    """The inner levels take the same W form of the up-leg (built at the end of the setup, when the Lanczos estimate has
    settled the level's damping): against PADNE_AMG_W=fine (W on the fine level only) one cycle applied to a probe
    agrees to float rounding, the single solve and the lockstep solve of four and eight right-hand sides take the same
    iterations (one either way) and give the same potentials to the solve tolerance."""
    sysm = synthetic.layered_system(3, 260, 200, via_lattice=5)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    b = -ro[1:n]
    rng = np.random.default_rng(11)
    B = np.stack([b] + [A @ rng.uniform(-1, 1, n - 1) for _ in range(7)])
    probe = rng.uniform(-1, 1, n - 1)

    def run():
        d = ctx.csr_from_scipy(A)
        one = d.solve_spd(b, precond="amg", rtol=1e-12)
        z = d.amg_apply(probe)
        four = d.solve_spd(B[:4], precond="amg", rtol=1e-12)
        eight = d.solve_spd(B, precond="amg", rtol=1e-12)
        d.close()
        return one, z, four, eight
    inner = run()
    switches.set("PADNE_AMG_W", "fine")
    fine_only = run()
    assert inner[0].levels == fine_only[0].levels >= 4      # at least two inner levels carry a W
    assert np.abs(inner[1] - fine_only[1]).max() <= 2e-5 * np.abs(fine_only[1]).max()
    assert not np.array_equal(inner[1], fine_only[1]), "the switch did not change the cycle: is the inner W built at all?"
    for a, c, rhs in ((inner[0], fine_only[0], b), (inner[2], fine_only[2], B[:4]), (inner[3], fine_only[3], B)):
        assert a.precond_fallbacks == 0 and c.precond_fallbacks == 0
        k = 1 if rhs.ndim == 1 else rhs.shape[0]
        assert abs(a.iterations - c.iterations) <= k
        assert np.abs(a.x - c.x).max() <= 1e-9 * np.abs(c.x).max()
        R = (A @ a.x.T).T - rhs
        assert np.linalg.norm(R) <= 1e-11 * np.linalg.norm(rhs)


@pytest.mark.parametrize("hub", [False, True])
def test_windowed_setup_kernels_build_the_same_hierarchy(ctx, switches, hub):
    """A fine level with an x-window plan (>= 65536 rows, band matrix) takes the windowed setup kernels: one-byte
    neighbour positions in the independent-set rounds, the decision fused into the second pass, prolongator rows merged
    in registers and written straight into CSR, A P consumed from its merge slots.  Same operations in the same order as
    the general kernels: every operator of the hierarchy is BIT-IDENTICAL to the one PADNE_NO_XWINDOW=1 builds -- with
    tiles that have no plan (via rows) inside the same launches, and (hub) with a row of more than 13 entries, which
    sends the prolongator of the whole level back to the general kernel."""
    sysm = synthetic.layered_system(2, 300, 240, via_lattice=6)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    if hub:
        els += [("R", 300 * 120 + 150, 300 * 7 * k + 31 * k + 5, 0.05) for k in range(1, 17)]      # one vertex, 16 far resistors
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    b = -ro[1:n]
    assert A.shape[0] >= 65536 and (np.diff(A.indptr).max() > 13) == hub

    def hierarchy():
        d = ctx.csr_from_scipy(A)
        res = d.solve_spd(b, precond="amg")
        ops = [(d.amg_level(l, "A"), d.amg_level(l, "P"), d.amg_level(l, "R")) for l in range(res.levels - 1)]
        d.close()
        return res, ops
    switches.set("PADNE_VERBOSE", "xw")
    res_w, ops_w = hierarchy()
    switches.set("PADNE_NO_XWINDOW", "1")
    res_g, ops_g = hierarchy()
    assert res_w.levels == res_g.levels >= 3 and res_w.iterations == res_g.iterations
    assert np.array_equal(res_w.x, res_g.x)
    for (Aw, Pw, Rw), (Ag, Pg, Rg) in zip(ops_w, ops_g):
        for W, G in ((Aw, Ag), (Pw, Pg), (Rw, Rg)):
            assert W.shape == G.shape and np.array_equal(W.indptr, G.indptr) and np.array_equal(W.indices, G.indices)
            assert np.array_equal(W.data, G.data)
    # and the hierarchy is the one the properties ask for
    A0, P0, R0 = ops_w[0]
    assert abs(R0 - P0.T).max() == 0.0
    ref = (P0.T @ A0 @ P0).tocsr()
    assert abs(ops_w[1][0] - ref).max() <= 1e-12 * abs(ref).max()


def test_sparse_products_with_rows_beyond_every_limit_of_the_lane_group_kernels(ctx):
    """Vertices with 13 ... 600 far neighbours make rows that leave the lane-group product kernels at every one of their limits --
    more entries of X than lanes (16, 32, 64), more products than the bit string of their first products holds, more distinct
    columns than half the hash table -- and go down the chain of passes (list of flagged rows with a whole wave, dense
    accumulator, one thread per row).  Whatever pass forms a row: R = P^T exactly, A_c = R (A P) to rounding against scipy,
    the same bits from two setups, and the solve agrees with the direct one."""
    sysm = synthetic.layered_system(2, 300, 240, via_lattice=6)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    rng = np.random.default_rng(11)
    nv = sysm.n_vertices
    for h, deg in enumerate((13, 20, 40, 80, 200, 600)):
        hub = 300 * (17 + 31 * h) + 11 * h + 7
        far = rng.choice(np.arange(1, nv), size=deg, replace=False)
        els += [("R", int(hub), int(f), 0.05 + 0.01 * (k % 7)) for k, f in enumerate(far) if int(f) != hub]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    A = (-Lo[1:nv, 1:nv]).tocsr()
    A.sort_indices()
    b = -ro[1:nv]
    assert np.diff(A.indptr).max() > 500

    def hierarchy():
        d = ctx.csr_from_scipy(A)
        res = d.solve_spd(b, precond="amg")
        ops = [(d.amg_level(l, "A"), d.amg_level(l, "P"), d.amg_level(l, "R")) for l in range(res.levels - 1)]
        ops.append((d.amg_level(res.levels - 1, "A"), None, None))
        d.close()
        return res, ops
    res, ops = hierarchy()
    res2, ops2 = hierarchy()
    assert res.status == _hip.OK and res.levels >= 3 and res.rel_residual <= 1.1e-12
    assert np.array_equal(res.x, res2.x)
    for (Al, P, R), (Al2, P2, R2) in zip(ops, ops2):
        assert np.array_equal(Al.indptr, Al2.indptr) and np.array_equal(Al.indices, Al2.indices) and np.array_equal(Al.data, Al2.data)
        if P is not None:
            assert np.array_equal(P.data, P2.data) and np.array_equal(R.data, R2.data)
    for lvl in range(len(ops) - 1):
        Al, P, R = ops[lvl]
        Ac = ops[lvl + 1][0]
        assert abs(R - P.T).max() == 0.0 and R.has_sorted_indices and Ac.has_sorted_indices
        ref = (P.T @ Al @ P).tocsr()
        assert abs(Ac - ref).max() <= 1e-12 * abs(ref).max()
    import scipy.sparse.linalg as spla
    ref = spla.spsolve(A.tocsc(), b)
    assert np.abs(res.x - ref).max() <= REL_TOL * np.abs(ref).max()


def test_setup_on_one_stream_builds_the_same_hierarchy(ctx, switches):
    """PADNE_SETUP_ONE_STREAM=1 (a switch for kernel traces: standalone times of the setup's side kernels) queues the
    side work of the multigrid setup on the main stream: the same kernels in another order of execution, hence the same
    operators and the same solution bit for bit."""
    A, b, _, _, _ = layered_spd(2, 150, 130, 5)

    def hierarchy():
        d = ctx.csr_from_scipy(A)
        res = d.solve_spd(b, precond="amg")
        ops = [(d.amg_level(l, "A"), d.amg_level(l, "P")) for l in range(res.levels - 1)]
        d.close()
        return res, ops
    res_two, ops_two = hierarchy()
    switches.set("PADNE_SETUP_ONE_STREAM", "1")
    res_one, ops_one = hierarchy()
    assert res_one.levels == res_two.levels >= 2 and res_one.iterations == res_two.iterations
    assert np.array_equal(res_one.x, res_two.x)
    for (A1, P1), (A2, P2) in zip(ops_one, ops_two):
        for U, V in ((A1, A2), (P1, P2)):
            assert np.array_equal(U.indptr, V.indptr) and np.array_equal(U.indices, V.indices) and np.array_equal(U.data, V.data)


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_multigrid_preconditioner_is_symmetric_positive_definite(ctx, switches, precision):
    """The cycle runs on single-precision copies of its operators by default (PADNE_AMG_F64=1: double): a fixed
    linear SPD operator up to the rounding of the working precision."""
    if precision == "f64":
        switches.set("PADNE_AMG_F64", "1")
    tol = 1e-10 if precision == "f64" else 2e-5
    A, b, _, _, _ = layered_spd(2, 70, 60, 4)
    d = ctx.csr_from_scipy(A)
    rng = np.random.default_rng(5)
    r1, r2 = rng.uniform(-1, 1, (2, A.shape[0]))
    z1, z2 = d.amg_apply(r1), d.amg_apply(r2)
    assert abs(r2 @ z1 - r1 @ z2) <= tol * (np.linalg.norm(r1) * np.linalg.norm(z2))
    assert r1 @ z1 > 0 and r2 @ z2 > 0
    # linear: M(a r1 + b r2) = a M r1 + b M r2
    z12 = d.amg_apply(2.0 * r1 - 3.0 * r2)
    assert np.abs(z12 - (2.0 * z1 - 3.0 * z2)).max() <= tol * np.abs(z12).max()
    assert np.array_equal(d.amg_apply(r1), z1)                          # deterministic
    # both precisions precondition the same operator: M32 r = M64 r to single-precision rounding
    if precision == "f32":
        switches.set("PADNE_AMG_F64", "1")
        d64 = ctx.csr_from_scipy(A)
        assert np.abs(d64.amg_apply(r1) - z1).max() <= 2e-5 * np.abs(z1).max()


def test_search_direction_stored_in_single_precision_solves_the_same_system(switches):
    """On one GPU the multigrid-preconditioned loop keeps its search direction as p / ||b|| in single precision
    (`csr_spmv_kernel<SPMV_DOT, double, double, double, ..., float>` multiplies it in double; x += alpha p and r -= alpha A p
    use the very same stored vector, so b - A x is tracked to double rounding).  Against PADNE_PCG_P64=1 (p in double): the
    same iteration count (one either way), the same potentials to the solve tolerance, a true residual at the tolerance -- on
    a system with an x-window plan and on one without (gather path), and with right-hand sides of 1e-30 and 1e+30 A."""
    cases = [layered_spd(2, 70, 60, 4), layered_spd(3, 260, 200, 5)]

    def run(A, b):
        c = _hip.Context(0)
        try:
            d = c.csr_from_scipy(A)
            out = [d.solve_spd(b * s, precond="amg", rtol=1e-12) for s in (1.0, 1e-30, 1e30)]
            d.close()
        finally:
            c.close()
        return out
    for A, b, _, _, _ in cases:
        f32 = run(A, b)
        switches.set("PADNE_PCG_P64", "1")
        f64 = run(A, b)
        switches.unset("PADNE_PCG_P64")
        for a, c, s in zip(f32, f64, (1.0, 1e-30, 1e30)):
            assert a.precond_fallbacks == 0 and c.precond_fallbacks == 0
            assert abs(a.iterations - c.iterations) <= 1, (a.iterations, c.iterations)
            assert np.abs(a.x - c.x).max() <= 1e-9 * np.abs(c.x).max()
            assert np.linalg.norm(A @ a.x - b * s) <= 2e-12 * np.linalg.norm(b * s)
        assert not np.array_equal(f32[0].x, f64[0].x), "the switch changed nothing"


def test_x_formed_from_the_kept_search_directions_has_the_bits_of_the_running_update(switches):
    """The one-GPU loop keeps its search directions (single precision, a place per iteration) and forms
    x = x_0 + sum_j alpha_j p_j when the loop has ended (`pcg_x_flush_kernel`), each row's terms in the order of the
    iterations, instead of reading and writing x in every iteration.  Same products, same additions, same order: bit for bit
    the x of the running update (PADNE_PCG_NO_XHIST=1), the same iterations -- cold, from an initial guess, with a ring of
    eight places that wraps several times within a solve (PADNE_FORCE=xhist_small), at a tolerance close to what the
    recurrence reaches, and through further solves on the same context (the places are reused) -- on a system whose
    product takes the x-window path and on one without a plan (the gather path)."""
    A, b, _, _, _ = layered_spd(3, 200, 150, 5)
    rng = np.random.default_rng(8)
    x0 = rng.uniform(-1, 1, A.shape[0]) * 1e-3
    _x_from_kept_directions(switches, A, b, x0)
    # no x-window plan: a numbering that scatters the columns of every tile (the gather path of the product)
    perm = rng.permutation(A.shape[0])
    Ap = A[perm][:, perm].tocsr()
    Ap.sort_indices()
    _x_from_kept_directions(switches, Ap, b[perm], x0[perm])


def _x_from_kept_directions(switches, A, b, x0):
    def run():
        c = _hip.Context(0)
        try:
            d = c.csr_from_scipy(A)
            out = [d.solve_spd(b, precond="amg", rtol=1e-12), d.solve_spd(b, precond="amg", rtol=1e-12, x0=x0),
                   d.solve_spd(b, precond="amg", rtol=2e-13), d.solve_spd(2.0 * b, precond="amg", rtol=1e-12)]
            d.close()
        finally:
            c.close()
        return out
    kept = run()
    switches.set("PADNE_FORCE", "xhist_small")
    ring = run()
    switches.unset("PADNE_FORCE")
    switches.set("PADNE_PCG_NO_XHIST", "1")
    running = run()
    switches.unset("PADNE_PCG_NO_XHIST")
    assert min(k.iterations for k in kept) > 16, "every solve is meant to wrap the ring of eight at least twice"
    for k, g, r in zip(kept, ring, running):
        assert k.precond_fallbacks == 0 and k.status == r.status == g.status
        assert k.iterations == r.iterations == g.iterations and k.restarts == r.restarts == g.restarts
        assert np.array_equal(k.x, r.x) and np.array_equal(g.x, r.x)
    assert np.linalg.norm(A @ kept[0].x - b) <= 2e-12 * np.linalg.norm(b)


def test_warm_start_whose_residual_lies_thirty_orders_below_the_right_hand_side(switches):
    """The single-precision vectors of the loop (cycle input, z, the stored search direction) are kept in units of ||b||;
    from an initial guess whose residual is 1e-30 ||b|| they would be denormals or zero (p.q = 0: a breakdown where the
    double loop iterated).  The unit of a warm start is therefore ||r_0||.  Two islands: the guess solves the first one
    exactly (its right-hand side is the device's own product A x_0, so r_0 vanishes there bit for bit), the second one has
    x_0 = 0 and a right-hand side of 1e-30: the solve has to reduce that residual by twelve orders like a cold solve of the
    second island alone."""
    A1, b1, _, _, _ = layered_spd(1, 90, 70, 5)
    n1 = A1.shape[0]
    A = sp.block_diag([A1, A1]).tocsr()
    A.sort_indices()
    x1 = np.random.default_rng(3).uniform(-1, 1, n1)
    x0 = np.concatenate([x1, np.zeros(n1)])
    delta = 1e-30 * b1 / np.abs(b1).max()
    for p64 in (False, True):
        if p64:
            switches.set("PADNE_PCG_P64", "1")
        c = _hip.Context(0)
        try:
            d = c.csr_from_scipy(A)
            b = d.matvec(x0)
            assert not b[n1:].any() and np.abs(b[:n1]).max() > 1.0
            b[n1:] = delta
            ref = c.csr_from_scipy(A1).solve_spd(delta, rtol=1e-12, precond="amg")
            res = d.solve_spd(b, rtol=0.0, atol=1e-12 * np.linalg.norm(delta), x0=x0, precond="amg")
            assert res.status == _hip.OK and abs(res.iterations - ref.iterations) <= 3 and res.iterations > 10
            assert np.array_equal(res.x[:n1], x1)                    # nothing moves where the residual is zero
            assert np.abs(res.x[n1:] - ref.x).max() <= 1e-8 * np.abs(ref.x).max()
        finally:
            c.close()


def test_single_precision_cycle_is_independent_of_the_units_of_the_system(ctx):
    """The cycle input is normalised by ||b||, so right-hand sides of 1e-30 A or 1e+30 A converge like 1 A ones."""
    A, b, _, _, _ = layered_spd(2, 70, 60, 4)
    d = ctx.csr_from_scipy(A)
    base = d.solve_spd(b, precond="amg")
    for scale in (1e-30, 1e30):
        res = d.solve_spd(b * scale, precond="amg")
        assert res.precond_fallbacks == 0 and abs(res.iterations - base.iterations) <= 1
        assert np.abs(res.x / scale - base.x).max() <= 1e-9 * np.abs(base.x).max()


def test_sparse_products_of_the_setup_split_their_rows_when_the_slots_exceed_the_index_space(ctx, switches):
    """A*P of a 130 M-row Laplacian has more product slots than 32-bit offsets address: the product is then formed in
    row halves and stacked.  PADNE_FORCE=spgemm_split:<slots> lowers the limit so that a small system takes that path (several
    levels of recursion); hierarchy and solution must not change."""
    A, b, _, _, _ = layered_spd(2, 90, 80, 4)
    base = ctx.csr_from_scipy(A).solve_spd(b, precond="amg")
    switches.set("PADNE_FORCE", "spgemm_split:30000")
    d = ctx.csr_from_scipy(A)
    res = d.solve_spd(b, precond="amg")
    assert res.levels == base.levels and res.levels >= 2 and res.precond_fallbacks == 0
    assert abs(res.operator_complexity - base.operator_complexity) <= 1e-12
    assert abs(res.iterations - base.iterations) <= 1
    assert np.abs(res.x - base.x).max() <= 1e-9 * np.abs(base.x).max()


def test_transposes_of_the_setup_through_cursors_and_the_sort(ctx, switches):
    """The transposition of a prolongator places every entry in the order of its row (per column the list of the 64-row
    waves that hold it, inside a wave a mask of the rows in a hash table of the columns met): the rows of the restriction
    come out in column order.  Columns met by more waves than a list holds, and the rows of a wave with more entries than
    its table takes, go through a cursor per column and are sorted afterwards.  PADNE_FORCE=transpose_cursors sends every wave
    that way: the restriction operators -- and with them the hierarchy and the solve -- must be the same bit for bit."""
    A, b, _, _, _ = layered_spd(3, 150, 110, 5)
    base = ctx.csr_from_scipy(A).solve_spd(b, precond="amg")
    switches.set("PADNE_FORCE", "transpose_cursors")
    res = ctx.csr_from_scipy(A).solve_spd(b, precond="amg")
    assert res.levels == base.levels and res.levels >= 3 and res.precond_fallbacks == 0
    assert res.operator_complexity == base.operator_complexity and res.iterations == base.iterations
    assert np.array_equal(res.x, base.x)


def test_multigrid_and_jacobi_agree_with_the_direct_solve(ctx):
    A, b, Lo, ro, n = layered_spd(4, 120, 100, 6)
    v_ref = O.solve_system(Lo, ro)[0]
    d = ctx.csr_from_scipy(A)
    xa = d.solve_spd(b, precond="amg")
    xj = d.solve_spd(b, precond="jacobi")
    scale = np.abs(v_ref[:n]).max()
    assert np.abs(xa.x - v_ref[1:n]).max() <= REL_TOL * scale
    assert np.abs(xj.x - v_ref[1:n]).max() <= REL_TOL * scale
    assert xa.iterations * 10 < xj.iterations
    assert xa.rel_residual <= 1.1e-12 and xj.rel_residual <= 1.1e-12


# ---- post-processing ----------------------------------------------------------------------------

@pytest.mark.parametrize("name", NAMES)
def test_power_density_vs_golden(ctx, name):
    g = H.load_golden(name)
    ms = H.meshes_of(g)
    if not ms:
        return
    xy, tri, mvo, mto, sig = flat(ms)
    got = ctx.power_density(xy, tri, mvo, mto, sig, g["v"])
    # the same through the mesh an assembled system keeps on the device
    empty = np.zeros(0, dtype=np.int64)
    n_vert = len(xy)
    Ld = ctx.assemble_system(n_vert, xy, tri, mvo, mto, sig, empty, empty, np.zeros(0))
    assert np.array_equal(Ld.power_density(g["v"][:n_vert], len(tri)), got)
    Ld.close()
    want = np.concatenate([g[f"pow{i}"] for i in range(len(ms))])
    assert np.array_equal(got, want)
    for i, (pxy, ptri, s, _) in enumerate(ms):                        # and through the ZeroForm seam
        z = mesh.ZeroForm(mesh.Mesh(pxy, ptri))
        z.values = g[f"pot{i}"].copy()
        assert np.array_equal(solver.compute_power_density(z, s).values, g[f"pow{i}"])


@pytest.mark.parametrize("values,expected", [((5.0, 5.0, 5.0), (0.0, 0.0)), ((0.0, 1.0, 0.0), (1.0, 0.0)),
                                             ((0.0, 0.0, 1.0), (0.0, 1.0)), ((0.0, 1.0, 1.0), (1.0, 1.0))])
def test_triangle_gradient_known_answers(ctx, values, expected):
    vs = [mesh.Vertex(mesh.Point(0.0, 0.0)), mesh.Vertex(mesh.Point(1.0, 0.0)), mesh.Vertex(mesh.Point(0.0, 1.0))]
    g = solver.compute_triangle_gradient(vs, list(values))           # tests/test_solver.py:1042-1112
    assert g.dx == pytest.approx(expected[0], abs=1e-10) and g.dy == pytest.approx(expected[1], abs=1e-10)
    gx, gy = O.triangle_gradient(np.array([0.0, 0.0]), np.array([1.0, 0.0]), np.array([0.0, 1.0]), *values)
    assert g.dx == gx and g.dy == gy
    with pytest.raises(ValueError):
        solver.compute_triangle_gradient(vs[:2], [0.0, 1.0])


# ---- Problem-level fixtures produced by the reference's own post-meshing path (config C1 among them) ----------

class FixtureMesher:
    """Stands where the CGAL mesher stands: poly_to_mesh(geom, seeds) returns the fixture's mesh of that polygon."""

    def __init__(self, by_geom):
        self.by_geom = by_geom
        self.seeds = {}

    def poly_to_mesh(self, geom, seed_points=()):
        self.seeds[id(geom)] = list(seed_points)
        return self.by_geom[id(geom)]


@pytest.mark.parametrize("family", ["padne_amd", "lookalike"])
@pytest.mark.parametrize("name", H.problem_golden_names())
def test_problem_fixture_through_solve(ctx, name, family):
    """``solve(prob)`` on the Problem of a reference-generated fixture (``problem_c1`` = config C1 of BASELINE.json, the
    via_tht_4layer-like 4-layer board): numbering, assembly, solve and post-processing against what the reference's
    NodeIndexer.create / assemble_system / solve_system / produce_layer_solutions returned for the same Problem."""
    import test_host_logic as TH
    g = H.load_golden(name)
    P = problem if family == "padne_amd" else TH._lookalike_problem_module()
    prob, nodes, flat_elements = H.build_problem(g, P)
    ms = H.problem_meshes(g)
    # one polygon per mesh, in mesh order (meshes are listed layer by layer in these fixtures)
    by_geom, per_layer = {}, {}
    for xy, tri, layer in ms:
        per_layer.setdefault(layer, []).append(mesh.Mesh(xy, tri))
    layers = []
    for li, lay in enumerate(prob.layers):
        geoms = H.Geoms(len(per_layer.get(li, [])))
        for token, m in zip(geoms.geoms, per_layer.get(li, [])):
            by_geom[id(token)] = m
        layers.append(P.Layer(shape=geoms, name=lay.name, conductance=lay.conductance) if family == "padne_amd"
                      else type("LayerWithGeoms", (), dict(shape=geoms, geoms=geoms.geoms, name=lay.name,
                                                           conductance=lay.conductance))())
    # connections refer to the layer objects: rebuild the problem around the new layers
    remap = {id(old): new for old, new in zip(prob.layers, layers)}
    nets = [P.Network(connections=[P.Connection(layer=remap[id(c.layer)], point=c.point, node_id=c.node_id)
                                   for c in net.connections], elements=list(net.elements)) for net in prob.networks]
    prob = P.Problem(layers=layers, networks=nets)
    mesher = FixtureMesher(by_geom)
    n_disc = int(g.get("n_disc", 0))
    with warnings.catch_warnings():
        warnings.simplefilter("error", solver.SolverWarning)
        if n_disc == 0:
            sol = solver.solve(prob, mesher=mesher)
        else:
            # copper that nothing drives (solver.py:862-870: the connectivity pre-pass keeps it out of the system and hands it
            # to produce_layer_solutions as it is): the post-meshing entry point, with the meshes the fixture lists
            disc = [[] for _ in layers]
            for k in range(n_disc):
                disc[int(g[f"disc_layer{k}"])].append(mesh.Mesh(g[f"disc_xy{k}"], g[f"disc_tri{k}"]))
            handed = [m for li in range(len(layers)) for m in per_layer.get(li, [])]
            sol = solver.solve_meshed(prob, handed, [m[2] for m in ms], disconnected_meshes_by_layer=disc)
            for ls, dl in zip(sol.layer_solutions, disc):
                assert len(ls.disconnected_meshes) == len(dl) and all(a is b for a, b in zip(ls.disconnected_meshes, dl))
    n_vert = sum(len(m[0]) for m in ms)
    scale = np.abs(g["v"][:n_vert]).max()
    # (an island that carries no current -- a dead end of the board -- is an equipotential: its power density is the rounding
    # noise of its potentials, 1e-24 of the board's; the bar is relative to the board, not to the island)
    pow_scale = max(max(g[f"pow{i}"].max() for i in range(len(ms))), 1e-300)
    worst = 0.0
    for li, ls in enumerate(sol.layer_solutions):
        idx = [i for i, m in enumerate(ms) if m[2] == li]
        assert len(ls.potentials) == len(idx)
        for i, zf, tf in zip(idx, ls.potentials, ls.power_densities):
            worst = max(worst, np.abs(zf.values - g[f"pot{i}"]).max())
            ref_pow = g[f"pow{i}"]
            assert np.abs(tf.values - ref_pow).max() <= 1e-7 * pow_scale
    assert worst <= REL_TOL * scale, f"potentials differ from the reference's direct solve by {worst / scale:.2e}"
    assert abs(sol.solver_info.ground_node_current - float(g["ground_node_current"])) <= 1e-8 * np.abs(g["v"][n_vert:]).max()
    assert sol.solver_info.residual_norm < 1e-9                      # the reference's own bar, tests/test_solver.py:2083-2089
    if name == "problem_c1":
        vs = next(e for e in flat_elements if solver.element_kind(e) == "VoltageSource")
        ids = {int(k): int(v) for k, v in zip(g["node_ids"], g["node_global"])}
        inv = {v: k for k, v in nodes.items()}
        flat_v = np.concatenate([zf.values for ls in sol.layer_solutions for zf in ls.potentials])
        assert abs(flat_v[ids[inv[vs.p]]] - flat_v[ids[inv[vs.n]]] - 1.0) < 1e-3     # tests/test_solver.py:1205
        if family == "padne_amd":                                                    # (the look-alike classes are test-local)
            assert len(pickle.loads(pickle.dumps(sol)).layer_solutions) == 4         # tests/test_solver.py:2047-2080
        # every connection point reached the mesher as a seed of its layer's polygons (solver.py:solve step 3)
        assert sum(len(v) for v in mesher.seeds.values()) == len(g["connections"])


@pytest.mark.parametrize("route", ["object_walk", "soup_dict", "cgal_dict"])
def test_config_c1_through_the_reference_shaped_mesh_hand_off(ctx, route):
    """SURVEY 8 f3 on the GPU: ``solve_meshed`` receives what padne's ``solve()`` holds after meshing
    (``mesh.py:778-786``) -- half-edge mesh OBJECTS, not arrays -- and must return the reference's potentials for config
    C1.  ``object_walk``: ``Mesh.from_reference`` walks vertices / faces ((v3, v1, v2) per face, ``mesh.py:320-325``);
    ``soup_dict``: the mesher stub of INTEGRATION.md left CGAL's output dict on the mesh (``_padne_hip_soup``), no walk;
    ``cgal_dict``: ``Mesh.from_cgal_output`` on the dict ``padne._cgal.mesh`` returns (``_cgal.cpp:479-488``)."""
    g = H.load_golden("problem_c1")
    prob, nodes, flat_elements = H.build_problem(g, problem)
    ms = H.problem_meshes(g)
    layer_of = [m[2] for m in ms]
    handed = []
    for xy, tri, _ in ms:
        out = {"vertices": [(float(x), float(y)) for x, y in xy], "triangles": [tuple(int(i) for i in t) for t in tri]}
        if route == "object_walk":
            he = H.HalfEdgeLikeMesh(xy, tri)
            assert not hasattr(he, "points") and not hasattr(he, "triangles")
            handed.append(he)
        elif route == "soup_dict":
            handed.append(H.HalfEdgeLikeMesh(xy[:0], tri[:0], soup=out))      # the walk would find nothing: the dict is used
        else:
            handed.append(mesh.Mesh.from_cgal_output(out))
    with warnings.catch_warnings():
        warnings.simplefilter("error", solver.SolverWarning)
        sol = solver.solve_meshed(prob, handed, layer_of)
    n_vert = sum(len(m[0]) for m in ms)
    scale = np.abs(g["v"][:n_vert]).max()
    worst = 0.0
    for li, ls in enumerate(sol.layer_solutions):
        idx = [i for i, m in enumerate(ms) if m[2] == li]
        assert len(ls.potentials) == len(idx)
        for i, zf, tf in zip(idx, ls.potentials, ls.power_densities):
            assert np.array_equal(zf.mesh.points, ms[i][0]) and np.array_equal(zf.mesh.triangles, ms[i][1])
            worst = max(worst, np.abs(zf.values - g[f"pot{i}"]).max())
            assert np.abs(tf.values - g[f"pow{i}"]).max() <= 1e-7 * max(g[f"pow{i}"].max(), 1e-300)
    assert worst <= REL_TOL * scale, f"{route}: potentials differ from the reference's solve by {worst / scale:.2e}"
    assert sol.solver_info.residual_norm < 1e-9
    assert abs(sol.solver_info.ground_node_current - float(g["ground_node_current"])) <= 1e-8 * np.abs(g["v"][n_vert:]).max()


# ---- Problem-level drop-in (the reference's synthetic end-to-end tests) --------------------------

def strip_problem(n_src=5):
    layer = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 10, 1)), name="F.Cu", conductance=1.0)
    nets = []
    ys = np.linspace(0.0, 0.8, n_src)        # on mesh rows (h = 0.2) so that every pad snaps to its own vertex
    lefts = [problem.Connection(layer=layer, point=mesh.Point(0.0, float(y))) for y in ys]
    rights = [problem.Connection(layer=layer, point=mesh.Point(10.0, float(y))) for y in ys]
    for cl, cr in zip(lefts, rights):
        nets.append(problem.Network(connections=[cl, cr],
                                    elements=[problem.VoltageSource(p=cr.node_id, n=cl.node_id, voltage=1.0)]))
    for a, b in zip(lefts[1:], lefts[:-1]):                           # tie the pads of each side (0 V glue)
        ca = problem.Connection(layer=layer, point=a.point)
        cb = problem.Connection(layer=layer, point=b.point)
        nets.append(problem.Network(connections=[ca, cb],
                                    elements=[problem.VoltageSource(p=ca.node_id, n=cb.node_id, voltage=0.0)]))
    return problem.Problem(layers=[layer], networks=nets)


def test_linear_rectangle_end_to_end(ctx):
    """tests/test_solver.py:461-595: 10x1 strip, 1 V end to end -> V linear in x within 0.05."""
    prob = strip_problem()
    mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.2), jitter=0.15, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("error", solver.SolverWarning)
        sol = solver.solve(prob, mesher=mesher)
    ls = sol.layer_solutions[0]
    msh, pot, pw = ls.meshes[0], ls.potentials[0], ls.power_densities[0]
    assert np.all(np.isfinite(pot.values))
    assert np.abs(pot.values - msh.points[:, 0] / 10.0).max() < 0.05
    # uniform power density sigma*|E|^2 = 1 * 0.1^2 (tests/test_solver.py:1249-1321 analogue)
    area = np.array([f.area for f in msh.faces])
    assert abs((pw.values * area).sum() / area.sum() - 0.01) < 1e-3 * 0.01 * 50
    assert sol.solver_info.residual_norm < 1e-9
    assert abs(sol.solver_info.ground_node_current) < 1e-9
    blob = pickle.dumps(sol)                                          # cli.py:223-224: Solution must pickle
    assert np.array_equal(pickle.loads(blob).layer_solutions[0].potentials[0].values, pot.values)


def test_long_thin_trace_resistance_and_power_density(ctx):
    """tests/test_solver.py:1214-1321: a 100 mm x 0.2 mm trace of 35 um copper carries 1 A -> 0.24 V end to end,
    power density I^2 R / (L w) everywhere (5 % per face away from the pads, 0.1 % area-weighted)."""
    sigma = 2082.5                                                    # 5.95e4 S/mm * 0.035 mm
    layer = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 100, 0.2)), name="F.Cu", conductance=sigma)
    P = mesh.Point
    ys = (0.0, 0.1, 0.2)                                              # the whole width of each end is a pad
    left = [problem.Connection(layer=layer, point=P(0.0, y)) for y in ys]
    right = [problem.Connection(layer=layer, point=P(100.0, y)) for y in ys]
    nets = [problem.Network(connections=[left[1], right[1]],
                            elements=[problem.CurrentSource(f=left[1].node_id, t=right[1].node_id, current=1.0)])]
    for side in (left, right):                                        # 0 V glue across the pad
        for a, b in ((side[0], side[1]), (side[2], side[1])):
            ca, cb = problem.Connection(layer=layer, point=a.point), problem.Connection(layer=layer, point=b.point)
            nets.append(problem.Network(connections=[ca, cb],
                                        elements=[problem.VoltageSource(p=ca.node_id, n=cb.node_id, voltage=0.0)]))
    prob = problem.Problem(layers=[layer], networks=nets)
    # a regular grid: every angle <= 90 degrees, so the reference's |cot| equals cot and the uniform field is reproduced
    # exactly (a jittered grid has obtuse corners whose flipped cotangents cost ~5 %, the tolerance of the reference's
    # own strip test)
    mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.1), jitter=0.0)
    sol = solver.solve(prob, mesher=mesher)
    ls = sol.layer_solutions[0]
    msh, pot, pw = ls.meshes[0], ls.potentials[0], ls.power_densities[0]
    x = msh.points[:, 0]
    v_left, v_right = pot.values[x == 0.0].mean(), pot.values[x == 100.0].mean()
    r_expected = 100.0 / (sigma * 0.2)                                # 0.2401 ohm
    assert abs(abs(v_right - v_left) - r_expected) < 0.01 * r_expected
    area = np.array([f.area for f in msh.faces])
    p_expected = 1.0 ** 2 * r_expected / (100.0 * 0.2)
    cxa = msh.points[msh.triangles].mean(axis=1)[:, 0]
    inner = (cxa > 1.0) & (cxa < 99.0)
    assert np.abs(pw.values[inner] - p_expected).max() < 0.05 * p_expected
    assert abs((pw.values * area).sum() / area.sum() - p_expected) < 2e-3 * p_expected
    assert sol.solver_info.residual_norm < 1e-9


def test_floating_plane_stays_equipotential(ctx):
    """tests/test_solver.py:1664-1758: a plane that touches the circuit in a single node carries no current, so all
    its vertices sit at that node's potential (to 1e-10), whatever happens on the driven layer."""
    top = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 20, 10)), name="F.Cu", conductance=2082.5)
    bot = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 20, 10)), name="B.Cu", conductance=1041.25)
    P = mesh.Point
    a, b = problem.Connection(layer=top, point=P(1, 5)), problem.Connection(layer=top, point=P(19, 5))
    nets = [problem.Network(connections=[a, b], elements=[problem.VoltageSource(p=b.node_id, n=a.node_id, voltage=2.0)])]
    load_a, load_b = problem.Connection(layer=top, point=P(3, 2)), problem.Connection(layer=top, point=P(17, 8))
    nets.append(problem.Network(connections=[load_a, load_b],
                                elements=[problem.Resistor(a=load_a.node_id, b=load_b.node_id, resistance=1.0)]))
    via_t, via_b = problem.Connection(layer=top, point=P(10, 5)), problem.Connection(layer=bot, point=P(10, 5))
    nets.append(problem.Network(connections=[via_t, via_b],
                                elements=[problem.Resistor(a=via_t.node_id, b=via_b.node_id, resistance=1e-3)]))
    prob = problem.Problem(layers=[top, bot], networks=nets)
    mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.25), jitter=0.2, seed=4)
    sol = solver.solve(prob, mesher=mesher)
    top_s, bot_s = sol.layer_solutions
    vb = bot_s.potentials[0].values
    vt, mt = top_s.potentials[0].values, top_s.meshes[0]
    k = int(np.argmin(np.hypot(mt.points[:, 0] - 10.0, mt.points[:, 1] - 5.0)))
    assert np.ptp(vt) > 1.9                                            # the driven layer really carries the 2 V
    assert np.abs(vb - vt[k]).max() <= 1e-10 * max(1.0, abs(vt[k]))
    assert np.abs(bot_s.power_densities[0].values).max() <= 1e-12
    # the 2 V source enters the reduced right-hand side as L c (norm ~1e4 here): rtol 1e-12 of that
    assert sol.solver_info.rel_residual <= 1e-12 and sol.solver_info.residual_norm < 1e-9


def test_unterminated_current_loop_warns_about_the_ground_current(ctx):
    """tests/test_solver.py:1829-1833: a current source drives one island from another island that nothing else
    connects to it.  The reference warns "Ground node current is not zero" and still returns a Solution; so does this
    path -- the floating island is held at one vertex instead of making the solve singular."""
    top = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 6, 4), structured.Rect(10, 0, 15, 4)),
                        name="F.Cu", conductance=2082.5)
    P = mesh.Point
    f, t = problem.Connection(layer=top, point=P(1, 2)), problem.Connection(layer=top, point=P(13, 2))
    nets = [problem.Network(connections=[f, t], elements=[problem.CurrentSource(f=f.node_id, t=t.node_id, current=1.5)])]
    prob = problem.Problem(layers=[top], networks=nets)
    mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.25), jitter=0.2, seed=5)
    with pytest.warns(solver.SolverWarning, match="Ground node current is not zero"):
        sol = solver.solve(prob, mesher=mesher)
    assert abs(abs(sol.solver_info.ground_node_current) - 1.5) < 1e-9
    ls = sol.layer_solutions[0]
    assert len(ls.potentials) == 2 and all(np.all(np.isfinite(z.values)) for z in ls.potentials)
    # each island carries its 1.5 A between the pad and the node that holds it: potentials stay of the order I / sigma
    assert all(np.abs(z.values).max() < 1.5 / 2082.5 * 10 for z in ls.potentials)
    # with a return path (a second source closing the loop) the same problem is regular and silent
    f2, t2 = problem.Connection(layer=top, point=P(14, 3)), problem.Connection(layer=top, point=P(2, 3))
    nets.append(problem.Network(connections=[f2, t2],
                                elements=[problem.CurrentSource(f=f2.node_id, t=t2.node_id, current=1.5)]))
    with warnings.catch_warnings():
        warnings.simplefilter("error", solver.SolverWarning)
        sol = solver.solve(problem.Problem(layers=[top], networks=nets), mesher=mesher)
    assert abs(sol.solver_info.ground_node_current) < 1e-9 and sol.solver_info.residual_norm < 1e-9


def test_coaxial_structure_end_to_end(ctx):
    """tests/test_solver.py:597-751: V(r) = ln(9/r)/ln 9 within 0.03, rings equipotential to 1e-3."""
    layer = problem.Layer(shape=structured.Shapes.of(structured.Annulus(0, 0, 1.0, 9.0)), name="F.Cu", conductance=1.0)
    n_t = 96
    ang = np.arange(n_t) * 2 * np.pi / n_t
    inner = [problem.Connection(layer=layer, point=mesh.Point(float(np.cos(a)), float(np.sin(a)))) for a in ang]
    outer = [problem.Connection(layer=layer, point=mesh.Point(float(9 * np.cos(a)), float(9 * np.sin(a)))) for a in ang]
    nets = [problem.Network(connections=[inner[0], outer[0]],
                            elements=[problem.VoltageSource(p=inner[0].node_id, n=outer[0].node_id, voltage=1.0)])]
    for ring in (inner, outer):
        for a, b in zip(ring[1:], ring[:-1]):
            ca = problem.Connection(layer=layer, point=a.point)
            cb = problem.Connection(layer=layer, point=b.point)
            nets.append(problem.Network(connections=[ca, cb],
                                        elements=[problem.VoltageSource(p=ca.node_id, n=cb.node_id, voltage=0.0)]))
    prob = problem.Problem(layers=[layer], networks=nets)

    class M(structured.StructuredMesher):
        def poly_to_mesh(self, poly, seed_points=()):
            xy, tri = synthetic.annulus_mesh(1.0, 9.0, 33, n_t)
            return mesh.Mesh(xy, tri)
    sol = solver.solve(prob, mesher=M())
    msh, pot = sol.layer_solutions[0].meshes[0], sol.layer_solutions[0].potentials[0]
    rad = np.hypot(msh.points[:, 0], msh.points[:, 1])
    assert np.abs(pot.values - np.log(9 / rad) / np.log(9)).max() < 0.03
    ring = pot.values[n_t * 10:n_t * 11]
    assert ring.max() - ring.min() < 1e-3




# ---- multi-GPU code path on one GPU ---------------------------------------------------------------

def test_halo_and_rccl_path_with_one_rank_communicator():
    """Exercises pack -> ncclAllGather -> SpMV on exchange columns -> fold -> ncclAllReduce with a
    1-rank RCCL communicator: off-diagonal couplings to a subset of unknowns are re-routed through
    the exchange area, which must not change the solution."""
    xy, tri = synthetic.jittered_grid(70, 50, seed=6)
    A = (-2082.5 * O.laplace_operator(xy, tri).tocsr()[1:, 1:]).tocsr()
    n = A.shape[0]
    b = np.random.default_rng(1).uniform(-1, 1, n)
    x_ref = O.solve_system(sp.bmat([[A, None], [None, sp.identity(1)]]).tocsr(), np.concatenate([b, [0.0]]))[0][:n]
    export = np.arange(3, n, 7, dtype=np.int32)
    m = len(export) + 5                                        # padded segment, like max over ranks
    pos = -np.ones(n, dtype=np.int64)
    pos[export] = np.arange(len(export))
    C = A.tocoo()
    cols = C.col.astype(np.int64).copy()
    move = (pos[cols] >= 0) & (C.row != C.col)
    cols[move] = n + pos[cols[move]]
    A_ext = sp.coo_matrix((C.data, (C.row, cols)), shape=(n + m, n + m)).tocsr()
    ctx2 = _hip.Context(0)
    try:
        ctx2.comm_init(ctx2.comm_unique_id(), 0, 1)
        ctx2.set_halo(n, m, export)
        d = ctx2.csr_from_scipy(A_ext)
        res = d.solve_spd(b, rtol=1e-12, precond="jacobi")
        assert res.x.shape == (n,)
        assert np.abs(res.x - x_ref).max() <= REL_TOL * np.abs(x_ref).max()
        # multigrid on the attached owned x owned block (what a rank of a layer-partitioned run does)
        block = ctx2.csr_from_scipy(A)
        d.set_preconditioner_block(block)
        res_amg = d.solve_spd(b, rtol=1e-12, precond="amg")
        assert res_amg.levels >= 2 and res_amg.iterations * 10 < res.iterations
        assert np.abs(res_amg.x - x_ref).max() <= REL_TOL * np.abs(x_ref).max()
        d.set_preconditioner_block(None)
        with pytest.raises(ValueError):                        # without a block a halo matrix cannot be coarsened
            d.solve_spd(b, precond="amg")
        plain = ctx2.csr_from_scipy(A)
        ctx2.clear_halo()
        res2 = plain.solve_spd(b, rtol=1e-12, precond="jacobi")  # reductions still via RCCL (communicator set)
        assert np.abs(res2.x - x_ref).max() <= REL_TOL * np.abs(x_ref).max()
        assert abs(res2.iterations - res.iterations) <= 3
        res3 = plain.solve_spd(b, rtol=1e-12, precond="amg")
        assert np.abs(res3.x - x_ref).max() <= REL_TOL * np.abs(x_ref).max()
        d.close()
        plain.close()
        block.close()
    finally:
        ctx2.close()


# ---- unstructured meshes (what CGAL produces: arbitrary vertex order, variable valence) ---------------

def delaunay_mesh(n_points, seed, hole=True):
    import scipy.spatial
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 40, (n_points, 2))
    if hole:
        pts = pts[np.hypot(pts[:, 0] - 20, pts[:, 1] - 20) > 6.0]
    d = scipy.spatial.Delaunay(pts)
    tri = d.simplices.astype(np.int32)
    if hole:
        c = pts[tri].mean(axis=1)
        tri = tri[np.hypot(c[:, 0] - 20, c[:, 1] - 20) > 6.5]
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    cross = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    tri[cross < 0] = tri[cross < 0][:, [0, 2, 1]]              # counter-clockwise like CGAL
    used = np.unique(tri)
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[used] = np.arange(len(used))
    return pts[used], remap[tri].astype(np.int32)


def test_unstructured_delaunay_mesh_assembly_and_solve(ctx):
    xy, tri = delaunay_mesh(30000, seed=11)
    n = len(xy)
    O.check_manifold(n, tri)                                           # sanity of the generator
    sigma = 2082.5
    far = int(np.argmax(xy[:, 0] + xy[:, 1]))
    near = int(np.argmin(xy[:, 0] + xy[:, 1]))
    els = [("I", near, far, 1.0), ("R", near, far, 0.05)]
    Lo, ro = O.assemble_system([(xy, tri, sigma)], 0, els, 7)
    Lo.sort_indices()
    stamps = solver.StampList(n + 1)
    g = 1 / 0.05
    for (i, j, v) in ((near, near, -g), (near, far, g), (far, far, -g), (far, near, g)):
        stamps.add(i, j, v)
    r = np.zeros(n + 1)
    r[near] += 1.0
    r[far] -= 1.0
    solver.setup_ground_node(7, stamps, r)
    L = solver.assemble_from_arrays([mesh.Mesh(xy, tri)], [sigma], stamps, n)
    got = L.tocsr()
    assert H.same_structure(got, Lo) and np.array_equal(got.data, Lo.data)
    assert np.array_equal(r, ro)
    v_ref, gc, res = O.solve_system(Lo, ro)
    v, info = solver.solve_system(L, r)
    assert np.abs(v[:n] - v_ref[:n]).max() <= REL_TOL * np.abs(v_ref[:n]).max()
    assert info.residual_norm < 1e-9 and abs(info.ground_node_current) < 1e-9
    assert info.iterations < 80                                          # multigrid copes with the irregular mesh
    pd = ctx.power_density(xy, tri, [0, n], [0, len(tri)], [sigma], v_ref[:n])
    assert np.array_equal(pd, O.power_density(xy, tri, v_ref[:n], sigma))
    L.dev.close()


def _graded_two_layer_system(needles: bool):
    import scipy.spatial
    rng = np.random.default_rng(21)
    u = rng.uniform(0, 1, (60000, 2))
    if needles:     # different grading in x and y: needle triangles, cotangent weights spanning 11 decades
        pts = np.column_stack([30.0 * u[:, 0] ** 3, 30.0 * u[:, 1] ** 2.5])
    else:           # isotropic grading towards the origin (what a sizing field produces): spacing 1 : 300
        rad = 30.0 * u[:, 0] ** 2.5
        pts = np.column_stack([rad * np.cos(2 * np.pi * u[:, 1]), rad * np.sin(2 * np.pi * u[:, 1])])
    pts = np.unique(np.round(pts, 9), axis=0)
    tri = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    cross = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    tri = tri[np.abs(cross) > 1e-12]                                        # drop degenerate slivers of the hull
    cross = cross[np.abs(cross) > 1e-12]
    tri[cross < 0] = tri[cross < 0][:, [0, 2, 1]]
    used = np.unique(tri)
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[used] = np.arange(len(used))
    xy, tri = pts[used], remap[tri].astype(np.int32)
    n1 = len(xy)
    ms = [(xy, tri, 2082.5), (xy.copy(), tri.copy(), 52.0)]                 # 40x conductivity jump between the layers
    n = 2 * n1
    ties = np.random.default_rng(5).choice(n1, 25, replace=False)
    els = [("R", int(t), int(n1 + t), 2e-3) for t in ties]
    src, snk = int(np.argmin(xy.sum(axis=1))), int(n1 + np.argmax(xy.sum(axis=1)))
    els += [("I", src, snk, 3.0)]
    Lo, ro = O.assemble_system(ms, 0, els, 11)
    stamps = solver.StampList(n + 1)
    for t in ties:
        g = 1 / 2e-3
        for (i, j, v) in ((t, t, -g), (t, n1 + t, g), (n1 + t, n1 + t, -g), (n1 + t, t, g)):
            stamps.add(int(i), int(j), v)
    r = np.zeros(n + 1)
    r[src] += 3.0
    r[snk] -= 3.0
    solver.setup_ground_node(11, stamps, r)
    L = solver.assemble_from_arrays([mesh.Mesh(xy, tri), mesh.Mesh(xy.copy(), tri.copy())], [2082.5, 52.0], stamps, n)
    assert np.array_equal(r, ro)
    return xy, tri, n, Lo, ro, L, r


def test_strongly_graded_mesh_with_conductivity_jump(ctx):
    """CGAL sizing fields give meshes whose element size varies by orders of magnitude, and stacked layers differ in
    conductance: vertex spacing graded 1 : 300 towards a point, two layers with a 40x conductivity jump stitched by
    a few resistors.  Strip reordering, multigrid and the x-window plan must all cope."""
    xy, tri, n, Lo, ro, L, r = _graded_two_layer_system(needles=False)
    edge = np.linalg.norm(xy[tri[:, 0]] - xy[tri[:, 1]], axis=1)
    assert np.percentile(edge, 99) > 100 * np.percentile(edge, 1)           # really graded
    v_ref = O.solve_system(Lo, ro)[0]
    with warnings.catch_warnings():
        warnings.simplefilter("error", solver.SolverWarning)
        v, info = solver.solve_system(L, r)
        v2, info2 = solver.solve_system(L, r, reorder=True)
    scale = np.abs(v_ref[:n]).max()
    assert np.abs(v[:n] - v_ref[:n]).max() <= REL_TOL * scale
    assert np.abs(v2[:n] - v_ref[:n]).max() <= REL_TOL * scale
    assert info.residual_norm < 1e-9 and info.iterations < 80 and info2.iterations < 80
    L.dev.close()


def test_needle_triangles_stall_gracefully(ctx):
    """Needle triangles put cotangent weights of 1e13 next to weights of 1e2 (entry ratio 1e11): b - A x cannot be evaluated below
    ~1e-6 ||b|| in binary64, so no iteration can certify rtol = 1e-12.  Like the reference (which always returns the
    LU answer and reports its residual), solve_system returns its best iterate, reports the residual it reached and
    warns -- it does not raise."""
    xy, tri, n, Lo, ro, L, r = _graded_two_layer_system(needles=True)
    assert abs(Lo).max() > 1e10 * np.percentile(abs(Lo.data), 50)
    v_ref = O.solve_system(Lo, ro)[0]
    with pytest.warns(solver.SolverWarning, match="relative residual"):
        v, info = solver.solve_system(L, r)
    assert np.all(np.isfinite(v))
    assert info.residual_norm < 1e-4                                       # reported honestly, far above 1e-12
    assert np.abs(v[:n] - v_ref[:n]).max() <= 1e-3 * np.abs(v_ref[:n]).max()
    L.dev.close()


# ---- more Problem-level behaviour of the reference's solve() ---------------------------------------------

def two_island_problem():
    """Two copper islands on one layer and a second layer; a star of 1 mOhm resistors to an internal node
    (multi-pad terminal, kicad.py:535-556), a regulator between the islands, a load resistor."""
    top = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 6, 3), structured.Rect(8, 0, 14, 3)),
                        name="F.Cu", conductance=2082.5)
    bot = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 14, 3)), name="B.Cu", conductance=1041.25)
    P = mesh.Point
    # supply: 5 V source on the left island, its return on the bottom layer
    c_in_p = problem.Connection(layer=top, point=P(0.6, 1.5))
    c_in_n = problem.Connection(layer=bot, point=P(0.6, 1.5))
    supply = problem.Network(connections=[c_in_p, c_in_n],
                             elements=[problem.VoltageSource(p=c_in_p.node_id, n=c_in_n.node_id, voltage=5.0)])
    # regulator: senses the left island (s_f -> s_t draws gain * output current), drives the right island at 3.3 V
    c_sf = problem.Connection(layer=top, point=P(5.4, 1.5))
    c_st = problem.Connection(layer=bot, point=P(5.4, 1.5))
    c_vp = problem.Connection(layer=top, point=P(8.6, 1.5))
    c_vn = problem.Connection(layer=bot, point=P(8.6, 1.5))
    reg = problem.Network(connections=[c_sf, c_st, c_vp, c_vn],
                          elements=[problem.VoltageRegulator(v_p=c_vp.node_id, v_n=c_vn.node_id, s_f=c_sf.node_id,
                                                             s_t=c_st.node_id, voltage=3.3, gain=1.0)])
    # load: three pads on the right island tied to an internal node by 1 mOhm, then 2.2 Ohm to the bottom layer
    pads = [problem.Connection(layer=top, point=P(13.2, y)) for y in (0.6, 1.5, 2.4)]
    c_ret = problem.Connection(layer=bot, point=P(13.2, 1.5))
    hub = problem.NodeID()
    load = problem.Network(connections=pads + [c_ret],
                           elements=[problem.Resistor(a=c.node_id, b=hub, resistance=1e-3) for c in pads] +
                                    [problem.Resistor(a=hub, b=c_ret.node_id, resistance=2.2)])
    return problem.Problem(layers=[top, bot], networks=[supply, reg, load]), (c_vp, c_vn, c_sf, c_st, hub, c_ret, pads)


def test_regulator_star_and_multi_island_problem_end_to_end(ctx):
    prob, (c_vp, c_vn, c_sf, c_st, hub, c_ret, pads) = two_island_problem()
    mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.3), jitter=0.1, seed=1)
    sol = solver.solve(prob, mesher=mesher)
    assert len(sol.layer_solutions) == 2
    assert len(sol.layer_solutions[0].meshes) == 2 and len(sol.layer_solutions[1].meshes) == 1
    # 5 V forced across a 2 kS sheet drives kilo-amps: the residual still meets the reference's absolute 1e-9; the
    # ground current (a difference of such currents) is held relative to them
    amps = 5.0 * 2082.5
    assert sol.solver_info.residual_norm < 1e-9 and abs(sol.solver_info.ground_node_current) < 1e-9 * amps
    # rebuild the same system through the reference-shaped seams and check against the oracle's direct solve
    meshes = [m for ls in sol.layer_solutions for m in ls.meshes]
    m2l = [0, 0, 1]
    vi = solver.VertexIndexer.create(meshes)
    ni = solver.NodeIndexer.create(prob, meshes, m2l, vi, prob.networks)
    L, r = solver.assemble_system(prob, meshes, m2l, vi, prob.networks, ni)
    els = []
    for net in prob.networks:
        for e in net.elements:
            ix = ni.node_to_global_index
            if isinstance(e, problem.Resistor):
                els.append(("R", ix[e.a], ix[e.b], e.resistance))
            elif isinstance(e, problem.VoltageSource):
                els.append(("V", ix[e.p], ix[e.n], e.voltage, ni.extra_source_to_global_index[e]))
            elif isinstance(e, problem.VoltageRegulator):
                els.append(("REG", ix[e.v_p], ix[e.v_n], ix[e.s_f], ix[e.s_t], e.voltage, e.gain,
                            ni.extra_source_to_global_index[e]))
    sig = [prob.layers[l].conductance for l in m2l]
    Lo, ro = O.assemble_system([(m.points, m.triangles, s) for m, s in zip(meshes, sig)], ni.internal_node_count, els,
                               solver.find_best_ground_node_index(prob, ni))
    Lo.sort_indices()
    got = L.tocsr()
    assert H.same_structure(got, Lo) and np.array_equal(got.data, Lo.data) and np.array_equal(r, ro)
    v_ref, gc_ref, _ = O.solve_system(Lo, ro)
    v, info = solver.solve_system(L, r)
    n_pot = len(vi) + ni.internal_node_count
    assert np.abs(v[:n_pot] - v_ref[:n_pot]).max() <= REL_TOL * np.abs(v_ref[:n_pot]).max()
    assert np.abs(v[n_pot:] - v_ref[n_pot:]).max() <= 1e-7 * np.abs(v_ref[n_pot:]).max()
    ix = ni.node_to_global_index
    # physics: the regulator holds 3.3 V across its output pins, the load draws ~1.5 A, the input mirrors it
    assert abs((v[ix[c_vp.node_id]] - v[ix[c_vn.node_id]]) - 3.3) < 1e-9
    i_out = v[ni.extra_source_to_global_index[prob.networks[1].elements[0]]]
    assert abs(i_out - 3.3 / 2.2) < 0.01 * 3.3 / 2.2
    i_in = v[ni.extra_source_to_global_index[prob.networks[0].elements[0]]]
    assert abs(abs(i_in) - i_out) < 0.01 * i_out         # gain 1.0: the supply delivers the mirrored current
    assert i_in < 0 < i_out                               # MNA sign: a delivering source has a negative branch current
    L.dev.close()


def test_five_regulators_advance_in_lockstep_and_match_the_direct_solve(ctx, switches):
    """K regulators add K right-hand sides to a solve (the gain columns, `solver.py:512-538`).  With K + 1 >= 5 they advance
    in lockstep through the batched cycle (a group of up to eight, zero-padded), one pass over the matrix and the hierarchy
    per iteration for all of them; fewer are solved one after the other (a lockstep iteration costs about 2.4 single
    ones).  Five regulators across two layers of 90 x 80 vertices: potentials and branch currents against the reference's
    direct solve; `padne_ctx_lockstep_groups` (test header) tells which path a solve took."""
    rng = np.random.default_rng(11)
    meshes, offs = [], [0]
    for layer, (nx, ny) in enumerate(((90, 80), (90, 80))):
        xy, tri = synthetic.jittered_grid(nx, ny, seed=20 + layer)
        meshes.append((xy, tri, 2082.5))
        offs.append(offs[-1] + len(xy))
    n_vert = offs[-1]
    vert = lambda l: int(rng.integers(offs[l], offs[l + 1]))
    els = [("R", vert(0), vert(1), float(10 ** rng.uniform(-3, -1))) for _ in range(12)]      # vias
    els.append(("I", vert(0), vert(1), 1.5))
    used = set()

    def fresh(l):
        while True:
            v = vert(l)
            if v not in used:
                used.add(v)
                return v
    n_extra = 0
    for k in range(5):
        vp, vn, sf, st = fresh(0), fresh(1), fresh(0), fresh(1)
        els.append(("REG", vp, vn, sf, st, 1.0 + 0.5 * k, 0.6 + 0.1 * k, n_vert + n_extra))
        n_extra += 1
        els.append(("R", vp, vn, 1.0 + k))                  # a load, so that the regulator delivers something
    Lo, ro = O.assemble_system(meshes, 0, els, 0)
    v_ref, _, _ = O.solve_system(Lo, ro)
    groups_before = ctx.lockstep_groups()
    v, info = solver.solve_system(Lo, ro)
    assert ctx.lockstep_groups() == groups_before + 1      # the six right-hand sides went as ONE zero-padded group
    n_pot = n_vert
    assert np.abs(v[:n_pot] - v_ref[:n_pot]).max() <= REL_TOL * np.abs(v_ref[:n_pot]).max()
    assert np.abs(v[n_pot:] - v_ref[n_pot:]).max() <= 1e-7 * np.abs(v_ref[n_pot:]).max()
    assert info.residual_norm < 1e-9
    switches.set("PADNE_NO_BATCH", "1")
    v1, info1 = solver.solve_system(Lo, ro)
    assert ctx.lockstep_groups() == groups_before + 1      # ... and one at a time with the batched path switched off
    assert np.abs(v1[:n_pot] - v_ref[:n_pot]).max() <= REL_TOL * np.abs(v_ref[:n_pot]).max()
    assert np.abs(v1 - v).max() <= 1e-9 * np.abs(v_ref).max()      # the grouping does not show beyond the tolerance


@pytest.mark.parametrize("n_reg", [1, 2, 3])
def test_one_to_three_regulators_in_the_narrow_lockstep_widths(ctx, switches, n_reg):
    """One to three regulators are two to four right-hand sides (`solver.py:512-538`).  The lockstep kernels exist in
    widths 2 and 4 as well as 8; measured, only exactly four right-hand sides gain from them (3.3
    against 4.0 single solves; two cost 2.3 against 2.0 -- profiles/r04_lockstep_widths.json), so that is what the default
    does: three regulators go as ONE group of width 4, one or two regulators one right-hand side at a time.
    PADNE_LOCKSTEP_NARROW=2 sends those through width 2 / a zero-padded width 4 too: potentials and regulator currents
    of every path against the reference's direct solve and against each other."""
    rng = np.random.default_rng(31 + n_reg)
    meshes, offs = [], [0]
    for layer, (nx, ny) in enumerate(((90, 80), (80, 90))):
        xy, tri = synthetic.jittered_grid(nx, ny, seed=50 + layer)
        meshes.append((xy, tri, 2082.5))
        offs.append(offs[-1] + len(xy))
    n_vert = offs[-1]
    p0 = iter(rng.permutation(offs[1]))
    p1 = iter(offs[1] + rng.permutation(offs[2] - offs[1]))
    els = [("R", int(next(p0)), int(next(p1)), float(10 ** rng.uniform(-3, -1))) for _ in range(12)]
    els.append(("I", int(next(p0)), int(next(p1)), 1.5))
    for k in range(n_reg):
        vp, vn, sf, st = int(next(p0)), int(next(p1)), int(next(p0)), int(next(p1))
        els.append(("REG", vp, vn, sf, st, 1.0 + 0.5 * k, 0.6 + 0.1 * k, n_vert + k))
        els.append(("R", vp, vn, 1.0 + k))
    Lo, ro = O.assemble_system(meshes, 0, els, 0)
    v_ref, _, _ = O.solve_system(Lo, ro)

    def check(v, info):
        assert np.abs(v[:n_vert] - v_ref[:n_vert]).max() <= REL_TOL * np.abs(v_ref[:n_vert]).max()
        assert np.abs(v[n_vert:] - v_ref[n_vert:]).max() <= 1e-7 * np.abs(v_ref[n_vert:]).max()
        assert info.residual_norm < 1e-9
    before = ctx.lockstep_groups()
    v, info = solver.solve_system(Lo, ro)
    check(v, info)
    assert ctx.lockstep_groups() == before + (1 if n_reg == 3 else 0)
    switches.set("PADNE_LOCKSTEP_NARROW", "2")
    before = ctx.lockstep_groups()
    v2, info2 = solver.solve_system(Lo, ro)
    check(v2, info2)
    assert ctx.lockstep_groups() == before + 1             # width 2, or width 4 (zero-padded for two regulators)
    switches.set("PADNE_LOCKSTEP_NARROW", "0")
    before = ctx.lockstep_groups()
    v0, info0 = solver.solve_system(Lo, ro)
    check(v0, info0)
    assert ctx.lockstep_groups() == before                 # everything one at a time
    assert np.abs(v2 - v0).max() <= 1e-9 * np.abs(v_ref).max() and np.abs(v - v0).max() <= 1e-9 * np.abs(v_ref).max()


def test_more_regulators_than_the_former_cap_of_the_device_plan(ctx):
    """`padne_kkt_solve` took at most 64 extra right-hand sides; the host path it replaced, like the
    reference (`solver.py:512-538`), takes any number.  66 regulators on two small layers: 67 right-hand sides in groups
    of eight, potentials and regulator currents against the reference's direct solve."""
    rng = np.random.default_rng(23)
    meshes, offs = [], [0]
    for layer in range(2):
        xy, tri = synthetic.jittered_grid(60, 50, seed=40 + layer)
        meshes.append((xy, tri, 2082.5))
        offs.append(offs[-1] + len(xy))
    n_vert = offs[-1]
    picks = iter(rng.permutation(offs[1]))
    picks1 = iter(offs[1] + rng.permutation(offs[2] - offs[1]))
    els = [("R", int(next(picks)), int(next(picks1)), float(10 ** rng.uniform(-3, -1))) for _ in range(10)]
    els.append(("I", int(next(picks)), int(next(picks1)), 1.5))
    K = 66
    for k in range(K):
        vp, vn, sf, st = int(next(picks)), int(next(picks1)), int(next(picks)), int(next(picks1))
        els.append(("REG", vp, vn, sf, st, 1.0 + 0.01 * k, 0.3 + 0.005 * k, n_vert + k))
        els.append(("R", vp, vn, 1.0 + 0.1 * k))
    Lo, ro = O.assemble_system(meshes, 0, els, 0)
    v_ref, _, _ = O.solve_system(Lo, ro)
    v, info = solver.solve_system(Lo, ro)
    assert np.abs(v[:n_vert] - v_ref[:n_vert]).max() <= REL_TOL * np.abs(v_ref[:n_vert]).max()
    assert np.abs(v[n_vert:] - v_ref[n_vert:]).max() <= 1e-7 * np.abs(v_ref[n_vert:]).max()
    assert info.residual_norm < 1e-9


# ---- size-independent properties at BASELINE scale (direct solve unaffordable there) --------------------

@pytest.fixture(scope="module")
def c3_system(ctx):
    """Config C3 of BASELINE.json: 4 layers of 1118x1118, N = 5 M, via rings, assembled on the device."""
    sysm = synthetic.config("C3")
    nv = sysm.n_vertices
    N = nv + 1
    xy, tri, mvo, mto, sig = flat(sysm.meshes)
    a, b, rr = sysm.resistors
    gg = 1 / rr
    rows = np.concatenate([np.stack([a, a, b, b], 1).reshape(-1), [N - 1, sysm.ground]])
    cols = np.concatenate([np.stack([a, b, b, a], 1).reshape(-1), [sysm.ground, N - 1]])
    vals = np.concatenate([np.stack([-gg, gg, -gg, gg], 1).reshape(-1), [1.0, 1.0]])
    Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    from padne_amd.reduction import Constraint, KKTLayout
    L = solver.SystemMatrix(Ld, KKTLayout(size=N, n_potential=nv,
                                          constraints=[Constraint(index=N - 1, p=sysm.ground, n=-1, value=0.0)]))
    yield sysm, L
    Ld.close()


def _rhs(sysm, pairs):
    r = np.zeros(sysm.n_vertices + 1)
    for f, t, i in pairs:
        r[f] += i
        r[t] -= i
    return r


def test_full_size_residual_ground_current_and_symmetry(ctx, c3_system):
    sysm, L = c3_system
    nv = sysm.n_vertices
    f, t, i = int(sysm.current_sources[0][0]), int(sysm.current_sources[1][0]), 1.0
    r = _rhs(sysm, [(f, t, i)])
    v, info = solver.solve_system(L, r)
    assert np.all(np.isfinite(v))
    assert info.residual_norm < 1e-9                                   # tests/test_solver.py:2083-2089
    assert abs(info.ground_node_current) < 1e-9                        # solver.py:880-888 (no SolverWarning)
    assert 5 < info.iterations < 100
    # current conservation through every layer pair: the via rings carry exactly the injected 1 A
    a, b, rr = sysm.resistors
    n_per = nv // len(sysm.meshes)
    for l in range(len(sysm.meshes) - 1):
        sel = (a // n_per == l) & (b // n_per == l + 1)
        # CurrentSource(f, t, I) raises V_t above V_f (tests/test_solver.py:82-83): the amp flows from t's layer to f's
        assert abs(np.sum((v[b[sel]] - v[a[sel]]) / rr[sel]) - 1.0) < 1e-7
    # the potential extremes sit at the source and the sink (discrete maximum principle of an M-matrix)
    assert int(np.argmax(v[:nv])) == t and int(np.argmin(v[:nv])) == f
    # matrix symmetry on random probes: x.(L y) == y.(L x) on the potential block
    rng = np.random.default_rng(0)
    x, y = rng.uniform(-1, 1, (2, nv + 1))
    x[-1] = y[-1] = 0.0
    assert abs(x @ (L @ y) - y @ (L @ x)) <= 1e-9 * abs(x @ (L @ y))


def test_full_size_superposition_and_reciprocity(ctx, c3_system):
    """tests/test_solver.py:1449-1564 (superposition) on the 5 M-node system, plus reciprocity of the
    transfer resistance, both without a direct solve."""
    sysm, L = c3_system
    nv = sysm.n_vertices
    n_per = nv // len(sysm.meshes)
    p1 = (3 * 1118 + 7, 2 * n_per + 500 * 1118 + 40)
    p2 = (n_per + 900 * 1118 + 900, 3 * n_per + 100 * 1118 + 1000)
    v1, _ = solver.solve_system(L, _rhs(sysm, [(p1[0], p1[1], 1.0)]))
    v2, _ = solver.solve_system(L, _rhs(sysm, [(p2[0], p2[1], 2.5)]))
    v12, _ = solver.solve_system(L, _rhs(sysm, [(p1[0], p1[1], 1.0), (p2[0], p2[1], 2.5)]))
    scale = np.abs(v12[:nv]).max()
    assert np.abs(v12[:nv] - (v1[:nv] + v2[:nv])).max() <= 1e-9 * scale
    # reciprocity: voltage across pair 2 per amp into pair 1 == voltage across pair 1 per amp into pair 2
    z21 = (v1[p2[0]] - v1[p2[1]]) / 1.0
    z12 = (v2[p1[0]] - v2[p1[1]]) / 2.5
    assert abs(z21 - z12) <= 1e-9 * max(abs(z21), abs(v1[:nv]).max())


def _check_against_the_direct_solve_samples(name, v):
    """tests/golden/direct_<config>.npz: a few thousand potentials of the reference's own solve call
    (scipy.sparse.linalg.spsolve on the un-reduced system, solver.py:772-775) at FULL size, produced once on a GPU box host
    by scripts/direct_full.py (392 s for C3, 696 s for C4 on one EPYC core; profiles/r02_direct_full.json).  North star:
    potentials within 1e-8 relative of the reference's direct solve."""
    import os
    path = os.path.join(H.GOLDEN, f"direct_{name}.npz")
    assert os.path.exists(path), f"{path} missing (scripts/direct_full.py writes it)"
    g = np.load(path)
    assert int(g["n_vertices"]) == len(v)
    err = np.abs(v[g["index"]] - g["potential"]).max() / float(g["max_abs_potential"])
    assert err <= REL_TOL, f"{name}: potentials differ from the reference's direct solve by {err:.2e} relative"
    assert float(g["residual_norm"]) < 1e-9


def test_config_c2_assembly_bit_for_bit_and_solve_at_full_size(ctx):
    """Config C2 of BASELINE.json at full size (one layer of 1000x1000, N = 1 M, 7 non-zeros per row), pinned
    INDEPENDENTLY of the device on both halves of the north star:
      * assembly: the device-assembled system (``asm_*`` kernels, mesh generated on the host) equals the CPU oracle's
        ``assemble_system`` (``mesh.py:124-139``, ``solver.py:171-213, 563-575`` restated) in structure and BIT FOR BIT
        in every value, at 1 M rows / 7 M entries;
      * solve: potentials within 1e-8 of ``tests/golden/direct_C2.npz`` -- samples of the reference's solve call
        (``tocsc + spsolve``, ``solver.py:772-775``) on the ORACLE-assembled matrix (``scripts/direct_cpu.py``, no device
        code on that side) -- and the reference's absolute residual bar ``||L v - r|| < 1e-9``
        (``tests/test_solver.py:2083-2089``)."""
    sysm = synthetic.config("C2")
    nv = sysm.n_vertices
    N = nv + 1
    xy, tri, mvo, mto, sig = flat(sysm.meshes)
    rows = np.array([N - 1, sysm.ground], dtype=np.int64)
    cols = np.array([sysm.ground, N - 1], dtype=np.int64)
    vals = np.array([1.0, 1.0])
    assert len(sysm.resistors[0]) == 0                                 # one layer: no via rings
    Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    els = [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
    Lo.sort_indices()
    Lg = Ld.to_scipy()
    assert Lg.shape == Lo.shape == (N, N) and Lg.nnz == Lo.nnz and Lo.nnz > 6.9e6
    assert np.array_equal(Lg.indptr, Lo.indptr) and np.array_equal(Lg.indices, Lo.indices)
    assert np.array_equal(Lg.data, Lo.data), "device assembly of C2 differs from the oracle"
    del Lg, Lo
    from padne_amd.reduction import Constraint, KKTLayout
    L = solver.SystemMatrix(Ld, KKTLayout(size=N, n_potential=nv,
                                          constraints=[Constraint(index=N - 1, p=sysm.ground, n=-1, value=0.0)]))
    v, info = solver.solve_system(L, ro)
    Ld.close()
    assert info.residual_norm < 1e-9 and abs(info.ground_node_current) < 1e-9
    assert 10 < info.iterations < 60 and info.rel_residual <= 1.1e-12
    _check_against_the_direct_solve_samples("C2", v[:nv])
    f, t = int(sysm.current_sources[0][0]), int(sysm.current_sources[1][0])
    assert int(np.argmax(v[:nv])) == t and int(np.argmin(v[:nv])) == f


def test_headline_config_at_full_size(ctx, switches):
    """Config C4 of BASELINE.json itself (8 layers of 1118x1118, N = 10 M, the bench workload), through the
    properties that need no direct solve: the windowed and the gather path of the product agree bit for bit, the
    mesh rows annihilate constants, the solve reaches the requested residual, conserves the current through
    every layer pair, puts the potential extremes at source and sink, and is linear in the right-hand side."""
    sysm = synthetic.config("C4")
    nv = sysm.n_vertices
    N = nv + 1
    xy, tri, mvo, mto, sig = flat(sysm.meshes)
    a, b, rr = sysm.resistors
    gg = 1 / rr
    rows = np.stack([a, a, b, b], 1).reshape(-1)
    cols = np.stack([a, b, b, a], 1).reshape(-1)
    vals = np.stack([-gg, gg, -gg, gg], 1).reshape(-1)
    Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    del xy, tri
    imap = np.arange(N, dtype=np.int32)
    imap[sysm.ground] = -1
    imap[imap > sysm.ground] -= 1
    imap[N - 1] = -1
    A = Ld.reduce(imap, nv - 1, -1.0)
    n = A.shape[0]
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, n)
    y_win = A.matvec(x)
    # constants: every row of the Laplacian part sums to zero; with the ground column removed A 1 is the ground
    # vertex's couplings, i.e. non-zero only in its handful of neighbours
    ones = A.matvec(np.ones(n))
    assert np.count_nonzero(np.abs(ones) > 1e-9 * np.abs(y_win).max()) <= 16
    switches.set("PADNE_NO_XWINDOW", "1")
    A_gather = Ld.reduce(imap, nv - 1, -1.0)
    assert np.array_equal(A_gather.matvec(x), y_win)
    A_gather.close()
    switches.unset("PADNE_NO_XWINDOW")
    Ld.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    f, t = int(sysm.current_sources[0][0]), int(sysm.current_sources[1][0])
    full = np.zeros(nv)
    full[f] += 1.0
    full[t] -= 1.0
    res = A.solve_spd(-full[keep], rtol=1e-12, precond="amg")
    assert res.status == _hip.OK and res.levels >= 4 and 10 < res.iterations < 60 and res.rel_residual <= 1.1e-12
    v = np.zeros(nv)
    v[keep] = res.x
    n_per = nv // len(sysm.meshes)
    for l in range(len(sysm.meshes) - 1):                    # the via rings carry exactly the injected 1 A
        sel = (a // n_per == l) & (b // n_per == l + 1)
        assert abs(np.sum((v[b[sel]] - v[a[sel]]) / rr[sel]) - 1.0) < 1e-7
    assert int(np.argmax(v)) == t and int(np.argmin(v)) == f
    _check_against_the_direct_solve_samples("C4", v)
    res2 = A.solve_spd(-2.5 * full[keep], rtol=1e-12, precond="amg")
    assert np.abs(res2.x - 2.5 * res.x).max() <= 1e-9 * np.abs(res2.x).max()
    A.close()


def test_config_c5_at_full_size(ctx):
    """Config C5 of BASELINE.json at full size: 8 current-source configurations on the 4-layer N = 5 M mesh, advanced in
    lockstep.  Every column reaches the tolerance; column 0 is the C3 right-hand side and is checked against the sampled
    potentials of the reference's direct solve; every column equals its own one-at-a-time solve."""
    sysm, xy_d, tri_d = synthetic.config_on_device(ctx, "C5")
    nv = sysm.n_vertices
    N = nv + 1
    a, b, rr = sysm.resistors
    gg = 1 / rr
    rows = np.stack([a, a, b, b], 1).reshape(-1)
    cols = np.stack([a, b, b, a], 1).reshape(-1)
    vals = np.stack([-gg, gg, -gg, gg], 1).reshape(-1)
    sig = np.array([m[2] for m in sysm.meshes])
    Ld = ctx.assemble_system(N, xy_d, tri_d, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
    xy_d.free()
    tri_d.free()
    imap = np.arange(N, dtype=np.int32)
    imap[sysm.ground] = -1
    imap[imap > sysm.ground] -= 1
    imap[N - 1] = -1
    A = Ld.reduce(imap, nv - 1, -1.0)
    Ld.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    src, snk = synthetic.multi_rhs_pairs(sysm, 8)
    src[0], snk[0] = int(sysm.current_sources[0][0]), int(sysm.current_sources[1][0])     # column 0 = config C3
    B = np.zeros((8, len(keep)))
    for k in range(8):
        full = np.zeros(nv)
        full[src[k]] += 1.0
        full[snk[k]] -= 1.0
        B[k] = -full[keep]
    res = A.solve_spd(B, rtol=1e-12, precond="amg")
    assert res.status == _hip.OK and res.rel_residual <= 1.1e-12 and res.precond_fallbacks == 0
    v0 = np.zeros(nv)
    v0[keep] = res.x[0]
    _check_against_the_direct_solve_samples("C3", v0)
    for k in (0, 5):
        one = A.solve_spd(B[k], rtol=1e-12, precond="amg")
        assert np.abs(one.x - res.x[k]).max() <= 1e-9 * np.abs(one.x).max()
        vk = np.zeros(nv)
        vk[keep] = res.x[k]
        assert int(np.argmax(vk)) == snk[k] and int(np.argmin(vk)) == src[k]              # maximum principle
    A.close()


def test_headline_config_on_eight_ranks(ctx):
    """Config C4 as BASELINE.json runs it -- one layer per rank, 8 ranks -- on ONE GPU through the in-process team:
    the layer partition, three exchanged multigrid levels and the single-reduction loop at full size.  Checked against
    the sampled potentials of the reference's direct solve; iteration count as on one GPU (28) within the spread seen
    for 2-8 ranks."""
    sysm = synthetic.config("C4")
    v, iters, res = run_team(sysm, 8, "amg")
    _check_against_the_direct_solve_samples("C4", v)
    assert res.rel_residual <= 1.1e-12 and res.levels >= 4 and 20 <= iters <= 45


@pytest.mark.parametrize("shape,jitter", [((128, 64), True), ((129, 127), True), ((127, 129), False), ((256, 256), False),
                                           ((255, 257), True), ((64, 2), False), ((2, 64), True), ((91, 90), False)])
def test_assembly_at_the_edges_of_its_tiles_and_chunks(ctx, shape, jitter):
    """The row kernel works in tiles of 128 rows and chunks of 64 tiles (8192 rows), its scan hands offsets from tile to
    tile: vertex counts on, one below and one above those boundaries, plus one extra unknown behind the vertices.  The
    unjittered grids have right angles -- exact zero weights that are not stored, so the rows are SHORTER than their
    fans suggest and only the in-kernel scan knows where they go.  Structure and values bit for bit against the oracle."""
    nx, ny = shape
    if jitter:
        xy, tri = synthetic.jittered_grid(nx, ny, seed=nx * 1000 + ny)
    else:
        gx, gy = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64), indexing="xy")
        xy = np.stack([gx.ravel(), gy.ravel()], 1) * 0.25
        q = (np.arange(ny - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).ravel()
        tri = np.concatenate([np.stack([q, q + 1, q + nx + 1], 1), np.stack([q, q + nx + 1, q + nx], 1)]).astype(np.int32)
    n = len(xy)
    els = [("R", 3, n - 2, 0.5), ("R", n // 2, n, 2.0), ("R", n, 7, 1.0)]      # two stamps on vertices, one internal node
    Lo, _ = O.assemble_system([(xy, tri, 2082.5)], 1, els, 0)
    Lo.sort_indices()
    N = Lo.shape[0]
    rows, cols, vals = [], [], []
    for _, a, b, res in els:
        g = 1.0 / res
        rows += [a, a, b, b]; cols += [a, b, b, a]; vals += [-g, g, -g, g]
    rows += [N - 1, 0]; cols += [0, N - 1]; vals += [1.0, 1.0]
    L = ctx.assemble_system(N, xy, tri.astype(np.int32), np.array([0, n], np.int64), np.array([0, len(tri)], np.int64),
                            np.array([2082.5]), np.array(rows, np.int64), np.array(cols, np.int64), np.array(vals))
    got = L.to_scipy()
    L.close()
    assert H.same_structure(got, Lo) and np.array_equal(got.data.view(np.int64), Lo.data.view(np.int64))
    if not jitter:
        assert got.nnz < 7 * n                                 # the diagonals of the squares carry no weight


def test_assemblies_running_side_by_side_on_one_gpu_are_the_serial_result(ctx):
    """The row kernel finds its offsets with a scan that runs INSIDE it (workgroups publish counts, a scanner wave the
    offsets; `asm_rows_in_place`).  Six contexts assemble six different systems on the same GPU at the same time --
    their workgroups share the chip, none of the kernels has it to itself -- and each result is the one the same
    context produces alone, bit for bit; four times, so that a wrong order of two memory operations -- or a workgroup that
    waits for a tile it holds itself -- has chances to show."""
    import threading
    # (four large systems -- a thousand tiles each -- and two of a few tiles, whose workgroups draw most of their tickets late)
    systems = [synthetic.layered_system(3, 260 + 7 * k, 230 + 5 * k, via_lattice=9) for k in range(4)]
    systems += [synthetic.layered_system(2, 40 + 3 * k, 30 + k, via_lattice=3) for k in range(2)]

    def arrays(sysm):
        N = sysm.n_vertices + 1
        a, b, r = sysm.resistors                             # stamps in the reference's order (solver.py:475-492, 558-560)
        g = 1.0 / r
        rows = np.concatenate([np.stack([a, a, b, b], 1).reshape(-1), [N - 1, sysm.ground]])
        cols = np.concatenate([np.stack([a, b, b, a], 1).reshape(-1), [sysm.ground, N - 1]])
        vals = np.concatenate([np.stack([-g, g, -g, g], 1).reshape(-1), [1.0, 1.0]])
        return N, (rows, cols, vals)

    def assemble(c, sysm, prepared):
        N, (rows, cols, vals) = prepared
        xy = np.concatenate([m[0] for m in sysm.meshes])
        tri = np.concatenate([m[1] for m in sysm.meshes]).astype(np.int32)
        mvo = np.cumsum([0] + [len(m[0]) for m in sysm.meshes]).astype(np.int64)
        mto = np.cumsum([0] + [len(m[1]) for m in sysm.meshes]).astype(np.int64)
        sig = np.array([m[2] for m in sysm.meshes], dtype=np.float64)
        L = c.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
        A = L.to_scipy()
        L.close()
        return A

    prepared = [arrays(sm) for sm in systems]
    alone = [assemble(ctx, sm, p) for sm, p in zip(systems, prepared)]
    for _ in range(2):
        out, errors = [None] * len(systems), []

        def worker(k):
            c = None
            try:
                c = _hip.Context(0)
                out[k] = assemble(c, systems[k], prepared[k])
            except BaseException as exc:
                errors.append((k, exc))
            finally:
                if c is not None:                            # stream, pinned buffer, pool: back at once, not at interpreter exit
                    c.close()
        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(len(systems))]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=120)
        stuck = [k for k, th in enumerate(threads) if th.is_alive()]
        assert not stuck, f"assemblies {stuck} did not come back (their kernels still hold the GPU)"
        assert not errors, errors
        for A, B in zip(alone, out):
            assert B is not None
            assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            assert np.array_equal(A.data.view(np.int64), B.data.view(np.int64))


@pytest.mark.parametrize("case", ["grid_with_stamps", "delaunay_fans"])
def test_second_path_of_the_row_kernel_is_the_first_bit_for_bit(ctx, switches, case):
    """When the in-kernel scan of `asm_rows_in_place` gives up (a chip shared with other work can starve its scanner until
    a bounded wait runs out) the host builds the rows again in two passes -- lengths, an ordinary scan, fill -- instead of
    failing the assembly.  PADNE_FORCE=asm_two_pass takes that path at once: structure and
    values of the single-pass result and of the oracle bit for bit, right angles (rows shorter than their fans), stamps,
    an internal node, fans of up to 12 triangles and a hub among them; the counter of the test header shows that the
    second path really ran."""
    if case == "grid_with_stamps":
        nx, ny = 257, 131
        gx, gy = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64), indexing="xy")
        xy = np.stack([gx.ravel(), gy.ravel()], 1) * 0.25
        xy[nx + 1::2] += np.random.default_rng(4).uniform(-0.03, 0.03, xy[nx + 1::2].shape)   # every other vertex off the lattice
        q = (np.arange(ny - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).ravel()
        tri = np.concatenate([np.stack([q, q + 1, q + nx + 1], 1), np.stack([q, q + nx + 1, q + nx], 1)]).astype(np.int32)
        xy[:nx] = np.stack([np.arange(nx) * 0.25, np.zeros(nx)], 1)                            # (the first line keeps right angles)
    else:
        from scipy.spatial import Delaunay
        rng = np.random.default_rng(11)
        pts = rng.uniform(0, 40, (9000, 2))
        hub = np.array([[20.0, 20.0]]) + 1e-3 * np.stack([np.cos(np.arange(30) * 2 * np.pi / 30), np.sin(np.arange(30) * 2 * np.pi / 30)], 1)
        xy = np.concatenate([pts, [[20.0, 20.0]], hub])                                      # a vertex with ~30 triangles around it
        tri = Delaunay(xy).simplices.astype(np.int32)
        a, b, c = xy[tri[:, 0]], xy[tri[:, 1]], xy[tri[:, 2]]
        cross = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
        tri[cross < 0] = tri[cross < 0][:, [0, 2, 1]]
    n = len(xy)
    els = [("R", 3, n - 2, 0.5), ("R", n // 2, n, 2.0), ("R", n, 7, 1.0)]
    Lo, _ = O.assemble_system([(xy, tri, 2082.5)], 1, els, 0)
    Lo.sort_indices()
    N = Lo.shape[0]
    rows, cols, vals = [], [], []
    for _, a_, b_, res in els:
        g = 1.0 / res
        rows += [a_, a_, b_, b_]; cols += [a_, b_, b_, a_]; vals += [-g, g, -g, g]
    rows += [N - 1, 0]; cols += [0, N - 1]; vals += [1.0, 1.0]

    def assemble():
        L = ctx.assemble_system(N, xy, tri, np.array([0, n], np.int64), np.array([0, len(tri)], np.int64),
                                np.array([2082.5]), np.array(rows, np.int64), np.array(cols, np.int64), np.array(vals))
        got = L.to_scipy()
        L.close()
        return got
    first = assemble()
    before = _hip.asm_second_path_count()
    switches.set("PADNE_FORCE", "asm_two_pass")
    second = assemble()
    assert _hip.asm_second_path_count() == before + 1
    for got in (first, second):
        assert H.same_structure(got, Lo) and np.array_equal(got.data.view(np.int64), Lo.data.view(np.int64))


def test_randomised_assembly_is_the_oracle_bit_for_bit(ctx):
    """scripts/fuzz_assembly.py: random Delaunay meshes (holes, fans with more triangles around a vertex than the
    incidence lists hold), random conductances and stamps, a hub row -- structure and values identical to the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_assembly", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_assembly.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(16, seed0=5, verbose=False) > 12          # the sweep reached vertices beyond the in-LDS path


def test_incidence_pass_through_the_hash_table_of_wide_vertex_ranges(ctx, switches):
    """The incidence pass counts the corners of a workgroup's triangles in LDS, indexed by vertex while the vertices lie
    within 4096 of each other, through a hash table of the vertices met when they do not (unordered triangle lists, mesh
    lines longer than that).  PADNE_FORCE=asm_hash sends every workgroup through the hash table: the same matrices bit for bit."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_assembly", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_assembly.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    switches.set("PADNE_FORCE", "asm_hash")
    assert fz.run(10, seed0=9, verbose=False) > 12


def test_randomised_systems_against_the_direct_solve(ctx):
    """scripts/fuzz_parity.py: random multi-layer systems with vias, internal nodes, current sources, forests of
    voltage sources and regulators, in the reference's KKT layout; the product path against the reference's direct
    solve on the same matrix.  (The script runs hundreds of cases; a fixed dozen here.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst_v, worst_i = fuzz.run(12, seed0=3, verbose=False)
    assert worst_v <= REL_TOL and worst_i <= 1e-7


def test_locality_reordering_is_transparent(ctx):
    """A scattered vertex numbering is solved in Z-order internally; v comes back in the caller's numbering."""
    xy, tri = delaunay_mesh(20000, seed=3, hole=False)
    n = len(xy)
    from padne_amd import reduction as red
    assert red.ordering_is_scattered(tri, n)
    stamps = solver.StampList(n + 1)
    r = np.zeros(n + 1)
    r[5] += 1.0
    r[n - 9] -= 1.0
    solver.setup_ground_node(11, stamps, r)
    L = solver.assemble_from_arrays([mesh.Mesh(xy, tri)], [2082.5], stamps, n)
    v_plain, i_plain = solver.solve_system(L, r, reorder=False)
    v_auto, i_auto = solver.solve_system(L, r)                      # picks the Z-order by itself
    v_forced, _ = solver.solve_system(L, r, reorder=True)
    scale = np.abs(v_plain[:n]).max()
    assert np.abs(v_auto[:n] - v_plain[:n]).max() <= 1e-9 * scale
    assert np.array_equal(v_auto, v_forced)
    assert i_auto.residual_norm < 1e-9 and i_plain.residual_norm < 1e-9
    L.dev.close()


def test_strip_numbering_on_the_device_is_the_hosts_permutation(ctx):
    """The locality ordering of the reduced unknowns (mesh, strip, x) is made on the device from the mesh the system was
    assembled from (``padne_kkt_create``, flags bit 0: per-mesh bounding boxes of the owners, 64-bit keys, one stable
    key-value sort).  It must be the permutation of ``reduction.apply_locality_ordering`` (numpy sorts over N-element
    arrays: 0.09 s of a 0.14 s ``solve_system`` call at 2 M vertices): the reduced matrices of the two plans are compared
    bit for bit -- two shuffled Delaunay meshes of different size, a voltage source that ties a vertex of one to a vertex
    of the other, an internal node, the ground."""
    from padne_amd import reduction as red_mod
    rng = np.random.default_rng(12)
    m1 = delaunay_mesh(9000, seed=5, hole=True)
    m2 = delaunay_mesh(5000, seed=6, hole=False)
    meshes = [mesh.Mesh(*m1), mesh.Mesh(m2[0] * 0.7 + 3.0, m2[1])]
    n1, n2 = len(meshes[0].points), len(meshes[1].points)
    nv = n1 + n2
    n_pot = nv + 1                                                    # one internal node
    N = n_pot + 2                                                     # a source current, the ground row
    stamps = solver.StampList(N)
    r = np.zeros(N)
    hub = nv
    for a in (17, n1 + 40, n1 + 333):                                 # the internal node hangs on three resistors
        g = 1 / 0.25
        stamps.add(a, a, -g); stamps.add(a, hub, g); stamps.add(hub, hub, -g); stamps.add(hub, a, g)
    iv = n_pot
    p_, n_ = 4321, n1 + 1234                                          # voltage source across the two meshes
    stamps.add(iv, p_, 1.0); stamps.add(iv, n_, -1.0); stamps.add(p_, iv, 1.0); stamps.add(n_, iv, -1.0)
    r[iv] = 0.5
    stamps.constraints.append(red_mod.Constraint(index=iv, p=p_, n=n_, value=0.5))
    r[100] += 1.0
    r[n1 + 7] -= 1.0
    solver.setup_ground_node(3, stamps, r)
    L = solver.assemble_from_arrays(meshes, [2082.5, 1041.25], stamps, n_pot)
    red = red_mod.build_reduction(L.layout)
    on_device = _hip.KktPlan(L.dev, n_pot, red.elim, red.tied, red.n_free, strip_order=True)
    red_mod.apply_locality_ordering(red, L.xy, L.mesh_offsets)
    on_host = _hip.KktPlan(L.dev, n_pot, red.elim, red.tied, red.n_free, index_map=red.index_map)
    A_dev, A_host = on_device.reduced_matrix().to_scipy(), on_host.reduced_matrix().to_scipy()
    assert A_dev.shape == A_host.shape == (red.n_free, red.n_free)
    assert np.array_equal(A_dev.indptr, A_host.indptr) and np.array_equal(A_dev.indices, A_host.indices)
    assert np.array_equal(A_dev.data, A_host.data)
    # and it is a band numbering: the x-window plan of the SpMV engages on it, which it does not in the mesher's numbering
    plain = _hip.KktPlan(L.dev, n_pot, red.elim, red.tied, red.n_free)
    bw = lambda A: float(np.mean(np.abs(A.tocoo().row - A.tocoo().col)))
    assert bw(A_dev) < 0.05 * bw(plain.reduced_matrix().to_scipy())
    v, info = solver.solve_system(L, r)                               # the whole call, through the device ordering
    v_ref = O.solve_system(L.tocsr(), r)[0]
    # (the 0.5 V source across two 2 kS sheets drives kiloamperes: 1e-9 absolute is the rounding floor of this system)
    assert np.abs(v - v_ref).max() <= REL_TOL * np.abs(v_ref).max() and info.rel_residual < 2e-12 and info.residual_norm < 1e-8
    for pl in (on_device, on_host, plain):
        pl.close()
    L.close()


def test_multigrid_with_hubs_and_strong_lumped_couplings(ctx):
    """Star hubs (hundreds of 1 mOhm resistors into one internal node, kicad.py:535-556) and very stiff
    layer-to-layer links next to the sheet Laplacian: rows of very different length and weight."""
    rng = np.random.default_rng(8)
    parts = [synthetic.jittered_grid(90, 70, seed=1), synthetic.jittered_grid(60, 50, seed=2)]
    ms = [(parts[0][0], parts[0][1], 2082.5), (parts[1][0], parts[1][1], 300.0)]
    n0, n1 = len(parts[0][0]), len(parts[1][0])
    nv = n0 + n1
    hub_a, hub_b = nv, nv + 1
    els = [("R", int(p), hub_a, 1e-3) for p in rng.choice(n0, 400, replace=False)]
    els += [("R", int(n0 + p), hub_b, 1e-3) for p in rng.choice(n1, 250, replace=False)]
    els += [("R", hub_a, hub_b, 0.5), ("R", 17, n0 + 23, 1e-6), ("I", 5, n0 + n1 - 3, 2.0)]
    Lo, ro = O.assemble_system(ms, 2, els, 0)
    v_ref, _, _ = O.solve_system(Lo, ro)
    v, info = solver.solve_system(Lo, ro)
    n_pot = nv + 2
    assert np.abs(v[:n_pot] - v_ref[:n_pot]).max() <= REL_TOL * np.abs(v_ref[:n_pot]).max()
    assert info.residual_norm < 1e-9
    A = (-Lo[1:n_pot, 1:n_pot]).tocsr()
    d = ctx.csr_from_scipy(A)
    res = d.solve_spd(-ro[1:n_pot], precond="amg")
    assert res.levels >= 2 and res.iterations < 200


# ---- the multi-rank solver with world > 1 on one GPU (in-process team instead of RCCL) ----------------------

def run_team(sysm, world, precond, block=False):
    """One thread per rank, each with its own context on GPU 0; returns (global potentials, iterations)."""
    import threading
    from padne_amd import distributed
    team = _hip.LocalTeam(world)
    out = [None] * world
    errors = []

    def rank_main(rank):
        try:
            c = _hip.Context(0)
            plan = distributed.build_layer_partition(sysm, rank, world)
            ds = distributed.DistributedSolver(c, plan, team=team, block_preconditioner=block)
            res = ds.solve(rtol=1e-12, precond=precond)
            sol = ds.solution()
            # the same solve again with the hierarchy in place: what ONE solve's iterations cost in collectives
            # (counters are per process: every rank of the team adds to them)
            c0 = c.comm_call_counts()[0]
            res2 = ds.solve(rtol=1e-12, precond=precond)
            c1 = c.comm_call_counts()[0]
            assert res2.iterations == res.iterations and np.array_equal(ds.solution(), sol)
            res.collectives = [b - a for a, b in zip(c0, c1)]
            res.split_tiles = [ds.A.split_tiles(-1)] + ([ds.A.split_tiles(1)] if precond == "amg" and not block else [])
            out[rank] = (plan, sol, res, ds.owned_reduced_global)
        except BaseException as exc:                              # wake the peers: they wait for this rank in a collective
            errors.append((rank, exc))
            team.abort()
    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(o is not None for o in out), "a rank did not finish"
    v = np.zeros(sysm.n_vertices)
    for plan, xs, _, idx in out:
        v[idx] = xs
    its = {o[2].iterations for o in out}
    assert len(its) == 1, f"ranks disagree on the iteration count: {its}"
    return v, its.pop(), out[0][2]


@pytest.mark.parametrize("world,precond", [(2, "jacobi"), (2, "amg"), (4, "amg"), (8, "amg"), (2, "amg-block"),
                                           (8, "amg-block")])
def test_layer_partitioned_solver_with_several_ranks_on_one_gpu(world, precond):
    block = precond == "amg-block"
    precond = precond.split("-")[0]
    sysm = synthetic.layered_system(8, 90, 70, via_lattice=5)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
    v_ref = O.solve_system(Lo, ro)[0][:sysm.n_vertices]
    v, iters, res = run_team(sysm, world, precond, block=block)
    assert np.abs(v - v_ref).max() <= REL_TOL * np.abs(v_ref).max()
    assert res.rel_residual <= 1.1e-12
    if precond == "amg":
        # one hierarchy over all ranks converges like the single-GPU one; block-Jacobi pays for the dropped couplings
        assert res.levels >= 2 and iters < (400 if block else 60)
        # Collectives of one solve, per rank (the counters saw all ranks between two barriers; ranks are not
        # synchronised at the read-out, hence the slack of one rank's worth).  Single-reduction CG: ONE all-reduce of
        # three doubles and ONE exchange of z per iteration (+ start, true-residual check); inside the cycle two float
        # exchanges per row-partitioned level -- one on the last of them, which computes its neighbours' corrected values
        # from the tail solution -- and the gather of the tail; none at all for block-Jacobi.
        n_ar, n_ag64, n_ag32, n_p2p = [c / world for c in res.collectives]
        # iterations are queued four at a time between two looks at the status word; + cycle and product of the start
        its = 4 * ((iters + 3) // 4) + 1
        assert n_ar <= its + 3 + 2, (n_ar, iters)         # 1 per iteration; ||b||, true residual
        if block:
            assert n_ag32 == 0 and n_p2p <= its + 1 + 2   # the exchange of z per iteration, A x of the true residual
        else:
            # the halo exchanges are peer-to-peer stores into the other ranks' mailboxes, not
            # collectives.  What is left per iteration: ONE all-reduce and the gather of the tail's right-hand side
            # (<= 4 was the bar); the four exchanges (z; the pre-smoothed iterate of both partitioned levels; the
            # corrected iterate of the first) are stores + one arrival each.
            assert n_ag64 <= 2, (n_ag64, iters)
            assert n_ag32 <= its + 2, (n_ag32, iters)
            assert n_ar + n_ag64 + n_ag32 <= 2 * its + 8, (res.collectives, iters)
            assert n_p2p <= (1 + 2 * 2 - 1) * its + 4, (n_p2p, iters)


def test_peer_to_peer_halo_stores_are_the_all_gather_bit_for_bit(switches):
    """The same solve with the halo exchanged by peer stores (default with the in-process team) and by all-gathers
    (PADNE_NO_P2P=1): the values that arrive are the same, so iterations and potentials are bit-identical; only the kind of
    communication differs (calls[3] against calls[1] / calls[2] of padne_comm_call_counts).  (Both runs with one launch per
    product: the interior / boundary split, which regroups the dot-product partials, is only built where the exchange
    overlaps the interior tiles, i.e. not for the all-gather form.)"""
    sysm = synthetic.layered_system(8, 90, 70, via_lattice=5)
    switches.set("PADNE_NO_SPLIT", "1")
    v_p2p, it_p2p, res_p2p = run_team(sysm, 4, "amg")
    switches.set("PADNE_NO_P2P", "1")
    v_ag, it_ag, res_ag = run_team(sysm, 4, "amg")
    assert it_p2p == it_ag and np.array_equal(v_p2p, v_ag)
    assert res_ag.collectives[3] == 0 and res_p2p.collectives[3] > 0
    # every exchange that was an all-gather is a peer-to-peer exchange now
    moved = (res_ag.collectives[1] + res_ag.collectives[2]) - (res_p2p.collectives[1] + res_p2p.collectives[2])
    assert moved == res_p2p.collectives[3], (res_ag.collectives, res_p2p.collectives)
    assert res_ag.split_tiles[0] == (0, 0)                # no overlap, no split


def test_products_split_into_interior_and_boundary_tiles_around_the_exchange(switches):
    """every product of a row-partitioned level that follows a halo exchange is launched in two parts
    -- the 64-row tiles whose columns are all owned while the exchange is under way, the tiles that read an exchange slot
    once it has landed (csr_build_split_plan).  Same products, same sums per row; only the grouping of the dot-product
    partials differs from the one-launch form (PADNE_NO_SPLIT=1): same iteration count, potentials equal to rounding, and
    the split run itself is bitwise reproducible."""
    sysm = synthetic.layered_system(8, 150, 120, via_lattice=6)        # 18 000 rows per layer: 282 tiles per rank at world 8
    v_split, it_split, res_split = run_team(sysm, 4, "amg")
    v_again, it_again, _ = run_team(sysm, 4, "amg")
    assert it_again == it_split and np.array_equal(v_again, v_split)
    (int0, bnd0), (int1, bnd1) = res_split.split_tiles            # the fine operator and the first coarse one are split
    assert bnd0 > 0 and int0 > 4 * bnd0 and int0 + bnd0 == (36000 + 63) // 64
    assert bnd1 > 0 and int1 > 0
    switches.set("PADNE_NO_SPLIT", "1")
    v_one, it_one, res_one = run_team(sysm, 4, "amg")
    assert res_one.split_tiles[0] == (0, 0)
    assert abs(it_split - it_one) <= 1
    assert np.abs(v_split - v_one).max() <= 1e-10 * np.abs(v_one).max()
    assert res_split.rel_residual <= 1.1e-12 and res_one.rel_residual <= 1.1e-12


def test_last_partitioned_level_computes_its_neighbours_from_the_tail(switches):
    """Up-leg of the last row-partitioned level: the other ranks' values after the coarse correction, x1 + P e, are
    computed locally (x1 came with the down-leg exchange, the remote rows of P with the setup, e is the tail solution
    every rank holds) instead of exchanged.  Same arithmetic as on the owning rank: bit-identical potentials, the same
    iterations, one float exchange less per cycle than with PADNE_AMG_EXCHANGE_ALL=1."""
    sysm = synthetic.layered_system(8, 90, 70, via_lattice=5)
    v_new, it_new, res_new = run_team(sysm, 4, "amg")
    switches.set("PADNE_AMG_EXCHANGE_ALL", "1")
    v_old, it_old, res_old = run_team(sysm, 4, "amg")
    assert it_new == it_old
    assert np.array_equal(v_new, v_old)
    its = 4 * ((it_new + 3) // 4) + 1
    # (float exchanges: all-gathers with PADNE_NO_P2P=1, peer-to-peer exchanges otherwise)
    saved = ((res_old.collectives[2] + res_old.collectives[3]) - (res_new.collectives[2] + res_new.collectives[3])) / 4
    assert its - 1 <= saved <= its + 1, (saved, its, res_old.collectives, res_new.collectives)


def test_single_reduction_cg_is_the_textbook_iteration(ctx, switches):
    """The rearranged loop of the row-partitioned runs (one global reduction per iteration) on ONE GPU against the
    textbook loop: same iterates up to rounding, same iteration count (+-1), same answer as the direct solve."""
    A, b, Lo, ro, n = layered_spd(4, 120, 100, 6)
    d = ctx.csr_from_scipy(A)
    ref = d.solve_spd(b, precond="amg")
    switches.set("PADNE_CG_SINGLE_REDUCTION", "1")
    sr = d.solve_spd(b, precond="amg")
    again = d.solve_spd(b, precond="amg")
    assert sr.status == _hip.OK and abs(sr.iterations - ref.iterations) <= 1 and sr.rel_residual <= 1.1e-12
    assert np.array_equal(sr.x, again.x)                               # bitwise reproducible like the other loop
    assert np.abs(sr.x - ref.x).max() <= 1e-10 * np.abs(ref.x).max()
    import scipy.sparse.linalg as spla
    direct = spla.splu(A.tocsc()).solve(b)
    assert np.abs(sr.x - direct).max() <= REL_TOL * np.abs(direct).max()
    warm = d.solve_spd(b, precond="amg", x0=sr.x)
    assert warm.iterations <= 2
    with pytest.raises(_hip.NotConvergedError):
        d.solve_spd(b, precond="amg", max_iter=3)
    ind = ctx.csr_from_scipy(sp.csr_matrix(A - 2.0 * sp.diags(A.diagonal())))   # indefinite: breakdown, not a hang
    with pytest.raises(_hip.HipError):
        ind.solve_spd(b, precond="amg")


def test_a_rank_that_fails_does_not_leave_its_peers_waiting():
    """a rank that leaves a team collective early (here: its driver raises before the first solve) must not
    block the others for ever -- the team is aborted and their collectives return E_COMM."""
    import threading
    from padne_amd import distributed
    sysm = synthetic.layered_system(4, 40, 30, via_lattice=3)
    team = _hip.LocalTeam(2)
    seen = [None, None]

    def rank_main(rank):
        try:
            c = _hip.Context(0)
            plan = distributed.build_layer_partition(sysm, rank, 2)
            ds = distributed.DistributedSolver(c, plan, team=team)
            if rank == 1:
                raise RuntimeError("rank 1 gives up before the solve")
            ds.solve(rtol=1e-12, precond="amg")
            seen[rank] = "finished"
        except BaseException as exc:
            seen[rank] = exc
            team.abort()
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is still waiting for its failed peer"
    assert isinstance(seen[1], RuntimeError)
    assert isinstance(seen[0], _hip.HipError) and seen[0].code == _hip.E_COMM


def _two_layer_problem(n_layers=2):
    P = mesh.Point
    layers = [problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, 30, 20)), name=f"L{i}", conductance=2082.5 / (i + 1))
              for i in range(n_layers)]
    nets = []
    rv = synthetic.via_ring_resistance(0.5)
    for (x, y) in ((6.0, 5.0), (22.0, 14.0), (15.0, 10.0)) if n_layers > 1 else ():
        conns, els = [], []
        for k in range(16):
            px, py = x + 0.15 * np.cos(k * np.pi / 8), y + 0.15 * np.sin(k * np.pi / 8)
            a = problem.Connection(layer=layers[0], point=P(px, py))
            b = problem.Connection(layer=layers[-1], point=P(px, py))
            conns += [a, b]
            els.append(problem.Resistor(a=a.node_id, b=b.node_id, resistance=rv))
        nets.append(problem.Network(connections=conns, elements=els))
    f = problem.Connection(layer=layers[0], point=P(2.0, 3.0))
    t = problem.Connection(layer=layers[-1], point=P(27.0, 17.0))
    hub = problem.NodeID()                                   # an internal node between two resistors
    nets.append(problem.Network(connections=[f, t], elements=[problem.CurrentSource(f=f.node_id, t=t.node_id, current=2.0)]))
    la, lb = problem.Connection(layer=layers[0], point=P(10.0, 16.0)), problem.Connection(layer=layers[-1], point=P(20.0, 4.0))
    nets.append(problem.Network(connections=[la, lb], elements=[problem.Resistor(a=la.node_id, b=hub, resistance=0.3),
                                                                problem.Resistor(a=hub, b=lb.node_id, resistance=0.2)]))
    # a resistor that lands on the vertex that will be the ground (unknown 0 = the corner (0, 0) of layer 0)
    g0, g1 = problem.Connection(layer=layers[0], point=P(0.0, 0.0)), problem.Connection(layer=layers[-1], point=P(29.0, 1.0))
    nets.append(problem.Network(connections=[g0, g1], elements=[problem.Resistor(a=g0.node_id, b=g1.node_id, resistance=5.0)]))
    return problem.Problem(layers=layers, networks=nets)


def solve_on_team(prob, mesher, world):
    """``solver.solve(prob, partition=...)`` on ``world`` threads, one context each on GPU 0: the Solutions of all ranks."""
    import threading
    from padne_amd import distributed
    team = _hip.LocalTeam(world)
    barrier = threading.Barrier(world)
    board = [None] * world
    sols, errors = [None] * world, []

    def gather_for(rank):
        def gather(obj):
            board[rank] = obj
            barrier.wait(timeout=120)
            out = list(board)
            barrier.wait(timeout=120)
            return out
        return gather

    def rank_main(rank):
        try:
            solver.set_context(_hip.Context(0))
            part = distributed.Partition(rank=rank, world=world, team=team, gather=gather_for(rank))
            sols[rank] = solver.solve(prob, mesher=mesher, partition=part)
        except BaseException as exc:
            errors.append((rank, exc))
            team.abort()
            barrier.abort()
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errors, errors
    assert all(sol is not None for sol in sols)
    return sols, team


@pytest.mark.parametrize("n_layers,world", [(2, 2), (1, 2), (2, 4)])
def test_solve_with_the_rows_dealt_to_several_ranks(ctx, n_layers, world):
    """``solve(prob, partition=...)``: a Problem (not only a SyntheticSystem) through the row-partitioned path -- by layer
    (2 layers, 2 ranks), by strips of a layer (1 layer on 2 ranks, 2 layers on 4), with an internal node and a resistor
    on the ground vertex -- gives every rank the Solution of the single-GPU solve.  Also with a VOLTAGE SOURCE between the
    layers (its two terminals become one unknown, owned by one rank)."""
    from padne_amd import distributed
    prob = _two_layer_problem(n_layers)
    mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=0.25), jitter=0.2, seed=6)
    a, b = prob.networks[-1].connections
    ca, cb = problem.Connection(layer=a.layer, point=a.point), problem.Connection(layer=b.layer, point=b.point)
    with_source = problem.Problem(layers=prob.layers, networks=prob.networks + [problem.Network(
        connections=[ca, cb], elements=[problem.VoltageSource(p=ca.node_id, n=cb.node_id, voltage=0.25)])])
    for pr, zero_ground_current in ((prob, True), (with_source, True)):
        ref = solver.solve(pr, mesher=mesher)
        sols, team = solve_on_team(pr, mesher, world)
        scale = max(np.abs(z.values).max() for ls in ref.layer_solutions for z in ls.potentials)
        for sol in sols:
            for ls, lr in zip(sol.layer_solutions, ref.layer_solutions):
                for z, zr, pw, pr_ in zip(ls.potentials, lr.potentials, ls.power_densities, lr.power_densities):
                    assert np.abs(z.values - zr.values).max() <= 1e-9 * scale
                    assert np.abs(pw.values - pr_.values).max() <= 1e-7 * max(pr_.values.max(), 1e-300)
            # (without sources the ground current is the sum of the injected currents, exactly 0 here; with them it is
            # recovered from the residual row of the ground like on one GPU: zero to the accuracy of the solve)
            assert abs(sol.solver_info.ground_node_current) < (1e-12 if pr is prob else 1e-9)
            assert sol.solver_info.residual_norm < 1e-9 and sol.solver_info.iterations < 80


@pytest.mark.parametrize("name,world", [("problem_c1", 2), ("problem_c1", 4), ("problem_mixed", 2), ("problem_two_planes", 2),
                                        ("problem_many_meshes", 4), ("problem_simple_trace", 2)])
def test_problem_fixtures_with_sources_on_several_ranks(ctx, name, world):
    """``solve(prob, partition=...)`` on the reference-generated Problem fixtures: config C1 of BASELINE.json (the
    via_tht_4layer-like board: four layers, via rings, three resistors, one 1 V source) and the mixed network (resistors,
    current source, voltage source, a REGULATOR: one extra solve with the same matrix).  Every rank gets the potentials
    AND the ground current the reference's own direct solve returned for the Problem (``tests/golden/problem_*.npz``)."""
    g = H.load_golden(name)
    prob, nodes, flat_elements = H.build_problem(g, problem)
    ms = H.problem_meshes(g)
    by_geom, per_layer = {}, {}
    for xy, tri, layer in ms:
        per_layer.setdefault(layer, []).append(mesh.Mesh(xy, tri))
    layers = []
    for li, lay in enumerate(prob.layers):
        geoms = H.Geoms(len(per_layer.get(li, [])))
        for token, m in zip(geoms.geoms, per_layer.get(li, [])):
            by_geom[id(token)] = m
        layers.append(problem.Layer(shape=geoms, name=lay.name, conductance=lay.conductance))
    remap = {id(old): new for old, new in zip(prob.layers, layers)}
    nets = [problem.Network(connections=[problem.Connection(layer=remap[id(c.layer)], point=c.point, node_id=c.node_id)
                                         for c in net.connections], elements=list(net.elements)) for net in prob.networks]
    prob = problem.Problem(layers=layers, networks=nets)
    sols, _ = solve_on_team(prob, FixtureMesher(by_geom), world)
    n_vert = sum(len(m[0]) for m in ms)
    scale = np.abs(g["v"][:n_vert]).max()
    bar = REL_TOL
    for sol in sols:
        for li, ls in enumerate(sol.layer_solutions):
            idx = [i for i, m in enumerate(ms) if m[2] == li]
            for i, zf, tf in zip(idx, ls.potentials, ls.power_densities):
                assert np.abs(zf.values - g[f"pot{i}"]).max() <= bar * scale
                assert np.abs(tf.values - g[f"pow{i}"]).max() <= 1e-7 * max(max(g[f"pow{k}"].max() for k in range(len(ms))), 1e-300)
        assert abs(sol.solver_info.ground_node_current - float(g["ground_node_current"])) <= 1e-8 * np.abs(g["v"][n_vert:]).max()
        assert sol.solver_info.residual_norm < 1e-9


def test_row_partitioned_hierarchy_with_several_exchanged_levels(switches):
    """Force three row-partitioned levels (default: everything below 262144 unknowns is gathered)."""
    switches.set("PADNE_AMG_GATHER_N", "700")
    sysm = synthetic.layered_system(8, 90, 70, via_lattice=5)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
    v_ref = O.solve_system(Lo, ro)[0][:sysm.n_vertices]
    for world in (2, 4):
        v, iters, res = run_team(sysm, world, "amg")
        assert np.abs(v - v_ref).max() <= REL_TOL * np.abs(v_ref).max()
        assert res.levels >= 4 and iters < 60


def test_batched_right_hand_sides_in_lockstep(ctx, switches):
    """Config C5 at test size: 11 current-source configurations = one lockstep group of 8 + 3 solved one at a time
    (and 14 = 8 + a padded group of 6); every column must match the direct solve and the one-at-a-time path."""
    A, b, Lo, ro, n = layered_spd(4, 120, 100, 6)
    rng = np.random.default_rng(9)
    k = 11
    B = np.zeros((k, A.shape[0]))
    for c in range(k):
        f, t = rng.choice(A.shape[0], 2, replace=False)
        B[c, f] += 1.0 + c
        B[c, t] -= 1.0 + c
    B[3] *= 1e-20                                  # wildly different scales in one group
    B[5] = 0.0                                     # and a zero right-hand side (converged at once)
    d = ctx.csr_from_scipy(A)
    res = d.solve_spd(B, precond="amg")
    assert res.rel_residual <= 1.1e-12 and res.precond_fallbacks == 0
    switches.set("PADNE_NO_BATCH", "1")
    seq = d.solve_spd(B, precond="amg")
    import scipy.sparse.linalg as spla
    lu = spla.splu(A.tocsc())
    for c in range(k):
        ref = lu.solve(B[c])
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(res.x[c] - ref).max() <= REL_TOL * scale
        assert np.abs(res.x[c] - seq.x[c]).max() <= REL_TOL * scale
    assert np.all(res.x[5] == 0.0)
    # the lockstep group does not cost more iterations per column than the one-at-a-time path
    assert res.iterations <= seq.iterations + 8
    switches.unset("PADNE_NO_BATCH")
    # a lockstep group started from a good guess stops at once; from a perturbed guess it still converges
    warm = d.solve_spd(B[:8], precond="amg", x0=res.x[:8])
    assert warm.iterations <= 8 and warm.rel_residual <= 1.1e-12
    rough = d.solve_spd(B[:8], precond="amg", x0=res.x[:8] * (1.0 + 1e-3))
    assert rough.rel_residual <= 1.1e-12 and rough.iterations < res.iterations
    assert np.abs(rough.x - res.x[:8]).max() <= REL_TOL * np.abs(res.x[:8]).max()
    # the loop of rounds 3-4 (z in double, x updated beside r: PADNE_PCG_P64=1) reaches the same solutions in as many iterations
    switches.set("PADNE_PCG_P64", "1")
    old = d.solve_spd(B[:8], precond="amg")
    switches.unset("PADNE_PCG_P64")
    new = d.solve_spd(B[:8], precond="amg")
    assert old.rel_residual <= 1.1e-12 and abs(old.iterations - new.iterations) <= 8
    for c in range(8):
        assert np.abs(old.x[c] - new.x[c]).max() <= REL_TOL * max(np.abs(new.x[c]).max(), 1e-300)
    # a guess whose residual lies thirty orders below the right-hand sides: the single-precision vectors of the loop are
    # in units of that first residual, not of ||b|| (floats of 1e-30 ||b|| would be denormals or zero)
    tiny = d.solve_spd(B[:8], precond="amg", x0=res.x[:8] * (1.0 + 1e-30))
    assert tiny.status == _hip.OK and tiny.rel_residual <= 1.1e-12
    B14 = np.vstack([B, B[:3] * 0.5])
    res14 = d.solve_spd(B14, precond="amg")
    assert res14.rel_residual <= 1.1e-12
    for c in range(14):
        ref = lu.solve(B14[c])
        assert np.abs(res14.x[c] - ref).max() <= REL_TOL * max(np.abs(ref).max(), 1e-300)


def test_x_window_tiles_and_gather_tiles_in_one_product(ctx, switches):
    """A scan-line mesh matrix large enough for the x-window plan, with lumped couplings to far-away unknowns in some
    rows (those tiles keep the gather path): the product is bit-identical to scipy's and to the plan-free kernel."""
    sysm = synthetic.layered_system(2, 260, 260, via_lattice=6)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    Lo, _ = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-Lo[1:n, 1:n]).tocsr()
    A.sort_indices()
    assert A.shape[0] > 65536
    x = np.random.default_rng(0).uniform(-1, 1, A.shape[0])
    y_plan = ctx.csr_from_scipy(A).matvec(x)
    switches.set("PADNE_NO_XWINDOW", "1")
    y_gather = ctx.csr_from_scipy(A).matvec(x)
    assert np.array_equal(y_plan, A @ x)
    assert np.array_equal(y_plan, y_gather)
    # random band matrices: three bands of varying width (some tiles fit runs of 72, some need 128, some none),
    # ragged rows, empty rows, 1 % of the rows with far-away columns, a rectangular shape
    switches.unset("PADNE_NO_XWINDOW")
    rng = np.random.default_rng(33)
    for n, ncols, half in ((70001, 70001, 20), (90000, 90500, 45), (66000, 66000, 70)):
        rows, cols = [], []
        for off in (-311, 0, 297):
            for d in range(-half, half + 1):
                keep = rng.random(n) < 0.12
                r = np.flatnonzero(keep)
                c = r + off + d
                ok = (c >= 0) & (c < ncols)
                rows.append(r[ok]); cols.append(c[ok])
        far = rng.choice(n, n // 100, replace=False)
        rows.append(far); cols.append(rng.integers(0, ncols, len(far)))
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        rows, cols = rows[rows % 997 != 0], cols[rows % 997 != 0]                # some empty rows
        M = sp.csr_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, ncols))
        M.sum_duplicates(); M.sort_indices()
        xv = rng.uniform(-1, 1, ncols)
        assert np.array_equal(ctx.csr_from_scipy(M).matvec(xv), M @ xv)


def test_nearest_vertex_matches_the_kd_tree(ctx):
    """Connection snapping (solver.py:425): the device brute force returns what the reference's KD-tree returns."""
    import scipy.spatial
    rng = np.random.default_rng(17)
    for n in (1, 5, 4096, 4097, 300001):
        pts = rng.uniform(-50, 50, (n, 2))
        q = np.vstack([rng.uniform(-60, 60, (257, 2)), pts[rng.integers(0, n, 7)]])        # some exactly on a vertex
        got = ctx.nearest_vertex(pts, q)
        _, ref = scipy.spatial.KDTree(pts, leafsize=32).query(q, k=1)
        assert np.array_equal(got, ref)
    # ties: the smallest index wins
    pts = np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 0.0]])
    assert list(ctx.nearest_vertex(pts, np.array([[1.0, 0.0], [0.0, 0.0]]))) == [0, 0]
    # a via in the centre of a grid cell is equidistant from its four corners: documented rule = smallest index
    gx, gy = np.meshgrid(np.arange(4.0), np.arange(3.0), indexing="xy")
    grid = np.stack([gx.reshape(-1), gy.reshape(-1)], axis=1)
    assert list(ctx.nearest_vertex(grid, np.array([[1.5, 0.5], [2.5, 1.5]]))) == [1, 6]
    # non-finite coordinates are an argument error, not an index of INT64_MAX
    for bad in (np.nan, np.inf):
        with pytest.raises(ValueError, match="finite"):
            ctx.nearest_vertex(pts, np.array([[bad, 0.0]]))
        with pytest.raises(ValueError, match="finite"):
            ctx.nearest_vertex(np.array([[0.0, bad], [1.0, 1.0]]), np.array([[0.5, 0.5]]))


def test_connections_equidistant_from_several_vertices_snap_like_the_kd_tree(ctx):
    """a connection exactly midway between vertices of an UNJITTERED grid (a via in the centre of a
    cell, a pad on the middle of an edge) above the 50 000-vertex switch to the device brute force.  The device reports
    how many vertices sit at exactly the minimum distance; ``NodeIndexer.create`` re-resolves those queries -- and only
    those -- with the reference's KD-tree (solver.py:389-392, 425), so the snapped node is the reference's."""
    import scipy.spatial
    nx, ny, h = 320, 260, 0.5                                          # 83 200 vertices > NEAREST_ON_DEVICE_FROM
    assert nx * ny >= solver.NEAREST_ON_DEVICE_FROM
    xy, tri = synthetic.jittered_grid(nx, ny, h, jitter=0.0)
    rng = np.random.default_rng(8)
    cx, cy = rng.integers(1, nx - 2, 40), rng.integers(1, ny - 2, 40)
    centres = np.stack([(cx + 0.5) * h, (cy + 0.5) * h], axis=1)       # 4 equidistant corners
    edges = np.stack([(cx + 0.5) * h, cy * h], axis=1)                 # 2 equidistant end points
    unique = np.stack([(cx + 0.3) * h, (cy + 0.2) * h], axis=1)        # one nearest vertex
    q = np.vstack([centres, edges, unique])
    got, ties = ctx.nearest_vertex(xy, q, with_ties=True)
    assert list(ties[:40]) == [4] * 40 and list(ties[40:80]) == [2] * 40 and list(ties[80:]) == [1] * 40
    tree = scipy.spatial.KDTree(xy, leafsize=32)
    _, ref = tree.query(q, k=1)
    assert np.array_equal(got[80:], ref[80:])                          # unique nearest vertex: the same everywhere
    d_got = np.hypot(*(xy[got] - q).T)
    d_ref = np.hypot(*(xy[ref] - q).T)
    assert np.array_equal(d_got, d_ref)                                # ties: another vertex at the same distance ...
    assert np.any(got[:80] != ref[:80])                                # ... and not always the tree's (smallest index here)
    # through the seam: the node numbering is the reference's
    layer = problem.Layer(shape=H.Geoms(1), name="F.Cu", conductance=1.0)
    conns = [problem.Connection(layer=layer, point=H.XY(x, y)) for x, y in q]
    els = [problem.Resistor(a=conns[k].node_id, b=conns[k + 1].node_id, resistance=1.0) for k in range(0, len(conns) - 1, 2)]
    prob = problem.Problem(layers=[layer], networks=[problem.Network(connections=conns, elements=els)])
    m = mesh.Mesh(xy, tri)
    ni = solver.NodeIndexer.create(prob, [m], [0], solver.VertexIndexer.create([m]), prob.networks)
    snapped = np.array([ni.node_to_global_index[c.node_id] for c in conns])
    assert np.array_equal(snapped, ref)


def test_spmm8_columns_are_bitwise_the_single_vector_products(ctx):
    """8 interleaved right-hand sides through one pass over the matrix; ragged rows, empty rows, long rows."""
    rng = np.random.default_rng(3)
    for M in (H.random_csr(5000, 5000, 7, 21), H.random_csr(777, 900, 40, 22), H.random_csr(130, 64, 300, 23)):
        d = ctx.csr_from_scipy(M)
        X = rng.uniform(-1, 1, (M.shape[1], 8))
        Y = d.matmat8(X)
        for j in range(8):
            yj = d.matvec(np.ascontiguousarray(X[:, j]))
            assert np.array_equal(Y[:, j], yj)
            assert np.array_equal(yj, M @ X[:, j])            # and both equal scipy's CSR product bit for bit


def test_relabel_with_injective_maps_is_the_slot_path_bit_for_bit(ctx, switches):
    """Eliminations and permutations (injective row and column maps) are relabelled directly -- count, scan, write,
    rows re-sorted by their new columns -- instead of through slots and a merge: same matrix, bit for bit, including
    rows longer than the in-LDS sort, dropped rows / columns and exact zeros."""
    rng = np.random.default_rng(21)
    M = H.random_csr(5000, 5000, 9, 3).tolil()
    M[17, :40] = rng.uniform(-1, 1, 40)                      # a long row
    M[:60, 23] = rng.uniform(-1, 1, (60, 1))                 # a long column
    M = M.tocsr()
    perm = rng.permutation(5000).astype(np.int32)
    drop = rng.choice(5000, 300, replace=False)
    rmap = perm.copy()
    rmap[drop] = -1
    rmap[rmap >= 0] = np.argsort(np.argsort(rmap[rmap >= 0])).astype(np.int32)      # compress to 0..n-1, order scrambled
    n_out = int((rmap >= 0).sum())
    cmap = np.arange(5000, dtype=np.int32)
    cmap[drop[:100]] = -1
    cmap[cmap >= 0] = np.arange((cmap >= 0).sum(), dtype=np.int32)                 # monotone column map
    cases = [("reduce", lambda m: m.reduce(rmap, n_out, -1.0)),
             ("relabel", lambda m: m.relabel(rmap, n_out, cmap, int((cmap >= 0).sum()), 2.5))]
    bad = rmap.copy()
    bad[7] = n_out                    # out of range: rejected (checked on the device copy of the map)
    with pytest.raises(ValueError):
        ctx.csr_from_scipy(M).reduce(bad, n_out, 1.0)
    Mz = M.copy()
    Mz.data[::37] = 0.0               # explicit zeros: no path stores them (the direct one notices and hands over to the slots)
    for src in (M, Mz):
        d = ctx.csr_from_scipy(src)
        for name, fn in cases:
            direct = fn(d).to_scipy()
            switches.set("PADNE_FORCE", "relabel_slots")
            slots = fn(d).to_scipy()
            switches.unset("PADNE_FORCE")
            assert direct.shape == slots.shape and np.array_equal(direct.indptr, slots.indptr), name
            assert np.array_equal(direct.indices, slots.indices) and np.array_equal(direct.data, slots.data), name
            assert direct.has_sorted_indices and direct.nnz > 0 and np.all(direct.data != 0.0)


def test_maps_that_only_drop_indices_are_relabelled_by_a_copy_with_holes(ctx, switches):
    """The reduction to the potential block drops indices and keeps the order of the rest (solver.py:544-560: no ground
    vertex, no multiplier row): count, scan, copy -- no histogram, no sort.  Against scipy and against the slot path, bit for
    bit: one hole, holes at both ends, scattered holes, a run of holes longer than the look-ahead of the map test (which the
    general path takes), rows of 0 to 40 entries, explicit zeros; the map as a host array and resident on the device."""
    rng = np.random.default_rng(5)
    n = 7000
    M = H.random_csr(n, n, 7, 31).tolil()
    M[33, 100:140] = rng.uniform(-1, 1, 40)
    M[70, 200:1500] = rng.uniform(-1, 1, 1300)      # a 64-row tile of several passes (512 entries each) of the wave kernels
    M[700:703, 0:400] = rng.uniform(-1, 1, (3, 400))
    M = M.tocsr()
    M.sort_indices()

    def cmap_of(drop):
        m = np.arange(n, dtype=np.int32)
        m[list(drop)] = -1
        keep = m >= 0
        m[keep] = np.arange(int(keep.sum()), dtype=np.int32)
        return m, keep
    drops = {"ground": [0], "last": [n - 1], "both ends": [0, n - 1], "scattered": sorted(rng.choice(n, 200, replace=False)),
             "long rows": [70, 701, 250, 1499],
             "long run": list(range(500, 600)), "none": []}
    Mz = M.copy()
    Mz.data[::41] = 0.0
    for src in (M, Mz):
        d = ctx.csr_from_scipy(src)
        for name, drop in drops.items():
            m, keep = cmap_of(drop)
            n_out = int(keep.sum())
            ref = (-1.0 * src[keep][:, keep]).tocsr()
            ref.eliminate_zeros()
            ref.sort_indices()
            host = d.reduce(m, n_out, -1.0).to_scipy()
            dm = ctx.to_device(m)
            dev = d.reduce(dm, n_out, -1.0).to_scipy()
            dm.free()
            switches.set("PADNE_FORCE", "relabel_slots")
            slots = d.reduce(m, n_out, -1.0).to_scipy()
            switches.set("PADNE_FORCE", "relabel_lanes")      # a lane per row (round 5) instead of a wave per 64 rows
            lanes = d.reduce(m, n_out, -1.0).to_scipy()
            switches.unset("PADNE_FORCE")
            for got in (host, dev, slots, lanes):
                assert got.shape == ref.shape and got.has_sorted_indices, name
                assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices), name
                assert np.array_equal(got.data, ref.data), name
    for far in (n, 2 * n, 2**31 - 1):     # out of range, by one and by far (nothing may be written through such an entry)
        bad, _ = cmap_of([3])
        bad[9] = far
        with pytest.raises(ValueError):
            ctx.csr_from_scipy(M).reduce(bad, n - 1, 1.0)
    # the C ABI takes rows whose columns do not ascend (padne_csr_from_host does not demand scipy's canonical form): a copy
    # would keep their order, so such a matrix goes through the general path and comes out sorted
    from padne_amd import _hip
    U = M.copy()
    for r in (5, 33, n - 1):
        k0, k1 = U.indptr[r], U.indptr[r + 1]
        U.indices[k0:k1] = U.indices[k0:k1][::-1].copy()
        U.data[k0:k1] = U.data[k0:k1][::-1].copy()
    indptr, indices, data = U.indptr.astype(np.int32), U.indices.astype(np.int32), U.data.astype(np.float64)
    h = _hip._P()
    _hip._check(ctx._lib.padne_csr_from_host(ctx._h, n, n, _hip._ptr(indptr, _hip._PI32), _hip._ptr(indices, _hip._PI32),
                                             _hip._ptr(data, _hip._PF64), _hip.C.byref(h)))
    du = _hip.CsrMatrix(ctx, h)
    m, keep = cmap_of(drops["scattered"])
    got = du.reduce(m, int(keep.sum()), -1.0).to_scipy()
    ref = (-1.0 * M[keep][:, keep]).tocsr()
    ref.sort_indices()
    assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
    assert np.array_equal(got.data, ref.data)


def test_relabel_and_vstack_against_scipy(ctx):
    rng = np.random.default_rng(12)
    M = H.random_csr(300, 420, 9, 12)
    d = ctx.csr_from_scipy(M)
    rmap = rng.integers(-1, 120, 300).astype(np.int32)
    cmap = rng.integers(-1, 200, 420).astype(np.int32)
    out = d.relabel(rmap, 120, cmap, 200, -2.0).to_scipy()
    R = sp.csr_matrix((np.ones((rmap >= 0).sum()), (np.flatnonzero(rmap >= 0), rmap[rmap >= 0])), shape=(300, 120))
    Cm = sp.csr_matrix((np.ones((cmap >= 0).sum()), (np.flatnonzero(cmap >= 0), cmap[cmap >= 0])), shape=(420, 200))
    ref = (-2.0 * (R.T @ M @ Cm)).tocsr()
    assert out.shape == (120, 200)
    assert abs(out - ref).max() <= 1e-12 * abs(ref).max()
    B = H.random_csr(77, 420, 5, 13)
    st = d.vstack(ctx.csr_from_scipy(B)).to_scipy()
    ref2 = sp.vstack([M, B]).tocsr()
    assert st.shape == ref2.shape and (st != ref2).nnz == 0
