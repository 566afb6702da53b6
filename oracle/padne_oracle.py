"""CPU oracle for the padne solver hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy/scipy restatement of the arithmetic of the reference
(``atx/padne``, ``padne/solver.py`` + ``padne/mesh.py``) on flat arrays.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; nothing under ``padne_amd/`` does, and the product path
raises when the HIP library is missing instead of falling back to anything here.

Pinning (see ``tests/test_oracle_golden.py``, ``tests/golden/``): every
function below is checked against golden vectors produced by running the
*reference's own functions* in the build container
(``tests/golden/make_golden.py`` via ``oracle/ref_loader.py``) and against the
analytic known-answer cases of the reference's test-suite
(``tests/test_solver.py:65-147, 776-852, 923-971, 1042-1112``).

Arithmetic notes (what "parity" means function by function):

* per-triangle cotangent weight follows ``HalfEdge.cotan`` (``mesh.py:124-139``)
  operation by operation: ``abs(dot/cross)/2`` in IEEE double, *with* the
  ``abs()`` (obtuse corners give positive weight) -- bitwise identical.
* an off-diagonal entry is the sum of at most two such terms, which is
  commutative, so it is bitwise identical to ``laplace_operator``
  (``solver.py:171-213``); exact zeros are dropped (``solver.py:187-190``).
* the diagonal is ``-(w_1 + w_2 + ...)``; the reference accumulates in
  half-edge *orbit* order (``mesh.py:78-84``), this oracle (and the HIP kernel)
  in ascending column order.  The two differ by at most a few ulp; the pin
  test uses rtol 1e-14 for the diagonal and exact equality elsewhere.
* lumped stamps (``solver.py:469-541``), ground (``solver.py:544-560``) are
  applied sequentially in element order, like the reference.
* the solve is the reference's own call, ``scipy.sparse.linalg.spsolve`` on
  the CSC matrix (``solver.py:772-775``); scipy is a third-party dependency of
  the reference (``pyproject.toml:46``, ``scipy>=1.15.0``), present here as
  scipy 1.15.3.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

DTYPE = np.float64  # solver.py:21

# --------------------------------------------------------------------------
# mesh.py:124-139  HalfEdge.cotan, restated per triangle corner
# --------------------------------------------------------------------------


def triangle_corner_cot_half(xy: np.ndarray, tri: np.ndarray) -> np.ndarray:
    """|cot(theta_o)|/2 for the three edges of every triangle.

    Returns ``c[t, e]`` for edge ``e`` = (tri[t,e], tri[t,(e+1)%3]) whose
    opposite corner is ``tri[t,(e+2)%3]``.  Same operations, same order as
    ``mesh.py:136-138``: vi = p_i - p_o, vk = p_k - p_o,
    abs(vi.dot(vk) / (vi ^ vk)) / 2.
    """
    xy = np.asarray(xy, dtype=DTYPE).reshape(-1, 2)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    out = np.empty((tri.shape[0], 3), dtype=DTYPE)
    with np.errstate(divide="ignore", invalid="ignore"):
        for e in range(3):
            i = tri[:, e]
            k = tri[:, (e + 1) % 3]
            o = tri[:, (e + 2) % 3]
            vix = xy[i, 0] - xy[o, 0]
            viy = xy[i, 1] - xy[o, 1]
            vkx = xy[k, 0] - xy[o, 0]
            vky = xy[k, 1] - xy[o, 1]
            dot = vix * vkx + viy * vky          # Vector.dot  mesh.py:24-25
            cross = vix * vky - viy * vkx        # Vector.__xor__  mesh.py:41-43
            out[:, e] = np.abs(dot / cross) / 2
    return out


def check_manifold(n_vert: int, tri: np.ndarray) -> None:
    """Raise ``ValueError("Non-manifold mesh")`` like ``mesh.py:342-343``.

    The reference builds half-edges keyed by directed vertex pair
    (``mesh.py:266-297``) and rejects a vertex that is the origin of more than
    one boundary half-edge.  On arrays: the boundary half-edges are the
    reversed directed triangle edges that have no triangle of their own.  A
    directed edge used by two triangles is rejected as well (the reference
    silently overwrites the face of the shared half-edge and then trips over
    the same check).
    """
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    if tri.size == 0:
        return
    u = tri.reshape(-1)
    v = tri[:, [1, 2, 0]].reshape(-1)
    key = u * n_vert + v
    rkey = v * n_vert + u
    ks = np.sort(key)
    if np.any(ks[1:] == ks[:-1]):
        raise ValueError("Non-manifold mesh")
    has_twin = np.isin(rkey, key)
    borig = v[~has_twin]                     # origin of the boundary half-edge v->u
    if borig.size and np.bincount(borig).max() > 1:
        raise ValueError("Non-manifold mesh")


# --------------------------------------------------------------------------
# solver.py:171-213  laplace_operator
# --------------------------------------------------------------------------


def laplace_operator(xy: np.ndarray, tri: np.ndarray, validate: bool = True) -> sp.coo_matrix:
    """Mesh-local cotangent Laplacian, reference sign (L_ik=+w, L_ii=-sum w).

    COO without duplicates, rows in ascending (row, col) order with the
    diagonal included; exact-zero weights are not stored.  ``validate=False`` skips the manifold test of
    ``Mesh.from_triangle_soup`` (for a piece of a mesh whose outer ring of vertices has incomplete fans, as the
    tests of the strip partition assemble; the reference itself always validates).
    """
    xy = np.asarray(xy, dtype=DTYPE).reshape(-1, 2)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    n = xy.shape[0]
    if validate:
        check_manifold(n, tri)
    c = triangle_corner_cot_half(xy, tri)
    # each triangle edge contributes to both directed entries (i,k) and (k,i)
    i = tri.reshape(-1)
    k = tri[:, [1, 2, 0]].reshape(-1)
    w = c.reshape(-1)
    rows = np.concatenate([i, k])
    cols = np.concatenate([k, i])
    vals = np.concatenate([w, w])
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    # merge the (at most two) contributions of an undirected edge:  0. + c1 + c2
    first = np.ones(rows.shape[0], dtype=bool)
    first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    idx = np.flatnonzero(first)
    cnt = np.diff(np.append(idx, rows.shape[0]))
    if cnt.size and cnt.max() > 2:
        raise ValueError("Non-manifold mesh")
    wsum = vals[idx].copy()
    two = cnt == 2
    wsum[two] = vals[idx[two]] + vals[idx[two] + 1]
    r_u, c_u = rows[idx], cols[idx]
    keep = wsum != 0                          # solver.py:187-190
    r_u, c_u, wsum = r_u[keep], c_u[keep], wsum[keep]
    # diagonal: -(w_1 + w_2 + ...) sequentially in ascending column order
    diag = np.zeros(n, dtype=DTYPE)
    if r_u.size:
        # sequential (left-to-right) accumulation, vectorised over rows
        start = np.flatnonzero(np.r_[True, r_u[1:] != r_u[:-1]])
        length = np.diff(np.append(start, r_u.size))
        acc = np.zeros(start.size, dtype=DTYPE)
        for j in range(int(length.max())):
            m = length > j
            acc[m] = acc[m] - wsum[start[m] + j]     # diag[i] -= ratio  solver.py:203
        diag[r_u[start]] = acc
    rows_all = np.concatenate([r_u, np.arange(n)])
    cols_all = np.concatenate([c_u, np.arange(n)])
    vals_all = np.concatenate([wsum, diag])
    order = np.lexsort((cols_all, rows_all))
    return sp.coo_matrix((vals_all[order], (rows_all[order], cols_all[order])),
                         shape=(n, n), dtype=DTYPE)


# --------------------------------------------------------------------------
# solver.py:469-560, 563-575, 748-812   assemble_system on flat arrays
# --------------------------------------------------------------------------
# Lumped elements are passed as plain tuples in network/element order:
#   ("R", a, b, resistance)                      problem.Resistor
#   ("I", f, t, current)                         problem.CurrentSource
#   ("V", p, n, voltage, i_v)                    problem.VoltageSource
#   ("REG", v_p, v_n, s_f, s_t, voltage, gain, i_v)   problem.VoltageRegulator
# where a, b, ... are *global* unknown indices and i_v is the index of the
# element's extra current unknown (NodeIndexer numbering, solver.py:441-460).


def assemble_system(meshes, n_internal: int, elements, i_gnd: int):
    """Return ``(L_csr, r)`` in the reference's layout.

    ``meshes``: list of ``(xy, tri, conductance)``; vertices are numbered mesh
    by mesh (``VertexIndexer``, ``solver.py:221-229``).  N = n_vertices +
    n_internal + n_extra + 1 (``solver.py:757-760``).
    """
    n_vert = sum(np.asarray(m[0]).reshape(-1, 2).shape[0] for m in meshes)
    n_extra = sum(1 for e in elements if e[0] in ("V", "REG"))
    N = n_vert + n_internal + n_extra + 1
    rows, cols, vals = [], [], []
    off = 0
    for xy, tri, sigma in meshes:                       # solver.py:563-575
        Lm = laplace_operator(xy, tri)
        rows.append(Lm.row.astype(np.int64) + off)
        cols.append(Lm.col.astype(np.int64) + off)
        vals.append(sigma * Lm.data)                    # conductance * laplace_operator(msh)
        off += Lm.shape[0]
    if rows:
        L = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                          shape=(N, N), dtype=DTYPE).tolil()
    else:
        L = sp.lil_matrix((N, N), dtype=DTYPE)
    r = np.zeros(N, dtype=DTYPE)
    for e in elements:                                  # solver.py:469-541
        kind = e[0]
        if kind == "R":
            _, a, b, res = e
            L[a, a] -= 1 / res
            L[a, b] += 1 / res
            L[b, b] -= 1 / res
            L[b, a] += 1 / res
        elif kind == "I":
            _, f, t, cur = e
            r[f] += cur
            r[t] += -cur
        elif kind == "V":
            _, p, n, volt, iv = e
            L[iv, p] = 1
            L[iv, n] = -1
            r[iv] = volt
            L[p, iv] = 1
            L[n, iv] = -1
        elif kind == "REG":
            _, vp, vn, sf, st, volt, gain, iv = e
            L[iv, vp] = 1
            L[iv, vn] = -1
            L[vp, iv] = 1
            L[vn, iv] = -1
            r[iv] += volt
            L[sf, iv] += gain
            L[st, iv] += -gain
        else:
            raise NotImplementedError(f"Unsupported node type {e}")
    L[-1, i_gnd] = 1                                    # solver.py:558-560
    L[i_gnd, -1] = 1
    r[-1] = 0
    return L.tocsr(), r


# --------------------------------------------------------------------------
# solver.py:398-466, 671-686   node numbering and ground choice, on tables
# --------------------------------------------------------------------------
# A network is ``dict(connections=[(layer, x, y, node)], elements=[(kind, t..., values...)])`` with ``node`` /
# terminals small integers that identify network nodes (the reference's opaque ``NodeID``); kinds as above but
# with node ids for terminals and no ``i_v``.

_N_TERMINALS = {"R": 2, "I": 2, "V": 2, "REG": 4}


def node_indexer_create(layer_points: dict, layer_gidx: dict, n_vertices: int, networks):
    """``NodeIndexer.create`` (``solver.py:398-466``): snap every connection to the nearest vertex of its layer
    (``scipy.spatial.KDTree(leafsize=32).query(k=1)``, as ``:389-392, 424``), number the nodes no connection
    names after the vertices, then one extra unknown per voltage source / regulator, network by network.

    Returns ``(node_to_global, extra_index_per_element, internal_node_count)``.  Internal nodes of ONE network
    are numbered in order of first appearance; the reference iterates a ``set`` there (``problem.py:86-96``), so
    with several internal nodes in a network its order is address-dependent (any order is the same system up to
    a permutation)."""
    import scipy.spatial
    trees = {l: scipy.spatial.KDTree(np.asarray(p, dtype=DTYPE).reshape(-1, 2), leafsize=32)
             for l, p in layer_points.items()}
    node_to_global: dict = {}
    for net in networks:
        for layer, x, y, node in net["connections"]:
            _, k = trees[int(layer)].query((x, y), k=1)
            g = int(layer_gidx[int(layer)][k])
            if node in node_to_global and node_to_global[node] != g:
                raise ValueError("Duplicate connection vertices found, this should not happen.")
            node_to_global[node] = g
    i_at = int(n_vertices)
    internal = 0
    for net in networks:
        for e in net["elements"]:
            for t in e[1:1 + _N_TERMINALS[e[0]]]:
                if t not in node_to_global:
                    node_to_global[t] = i_at
                    i_at += 1
                    internal += 1
    extra = []
    for net in networks:
        for e in net["elements"]:
            if e[0] in ("V", "REG"):
                extra.append(i_at)
                i_at += 1
            else:
                extra.append(-1)
    return node_to_global, extra, internal


def find_best_ground_node_index(networks, node_to_global) -> int:
    """``solver.py:671-686``: negative terminal of the voltage source with the highest voltage, else unknown 0."""
    best, ground = float("-inf"), 0
    for net in networks:
        for e in net["elements"]:
            if e[0] == "V" and e[3] > best:
                best, ground = e[3], node_to_global[e[2]]
    return ground


def globalise_elements(networks, node_to_global, extra):
    """Elements of all networks in stamping order with global unknown indices: the input of assemble_system."""
    out, k = [], 0
    for net in networks:
        for e in net["elements"]:
            nt = _N_TERMINALS[e[0]]
            row = (e[0],) + tuple(node_to_global[t] for t in e[1:1 + nt]) + tuple(e[1 + nt:])
            if extra[k] >= 0:
                row = row + (extra[k],)
            out.append(row)
            k += 1
    return out


def solve_system(L, r):
    """``solver.py:767-780``: spsolve on CSC, residual norm, ground current."""
    L_csc = sp.csc_matrix(L)
    v = spla.spsolve(L_csc, r)
    residual_norm = float(np.linalg.norm(L_csc @ v - r))
    return v, float(v[-1]), residual_norm


# --------------------------------------------------------------------------
# solver.py:689-745   triangle gradient and power density
# --------------------------------------------------------------------------


def face_vertex_order(tri: np.ndarray) -> np.ndarray:
    """Vertex order in which the reference visits a face.

    ``Face.edge`` ends up as the *last* interior half-edge created for the
    triangle (``mesh.py:320-325``: the (v3, v1) edge), so ``face.vertices``
    yields (v3, v1, v2).
    """
    tri = np.asarray(tri).reshape(-1, 3)
    return tri[:, [2, 0, 1]]


def triangle_gradient(p1, p2, p3, f1, f2, f3):
    """``compute_triangle_gradient`` (``solver.py:689-725``), vectorised, same operation order."""
    x1, y1 = p1[..., 0], p1[..., 1]
    x2, y2 = p2[..., 0], p2[..., 1]
    x3, y3 = p3[..., 0], p3[..., 1]

    def interpolate(x, y):
        D = (y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3)
        l1 = ((y2 - y3) * (x - x3) + (x3 - x2) * (y - y3)) / D
        l2 = ((y3 - y1) * (x - x3) + (x1 - x3) * (y - y3)) / D
        l3 = 1 - l1 - l2
        return l1 * f1 + l2 * f2 + l3 * f3

    with np.errstate(divide="ignore", invalid="ignore"):
        gx = interpolate(x1 + 1, y1) - f1
        gy = interpolate(x1, y1 + 1) - f1
    return gx, gy


def power_density(xy, tri, values, conductivity: float) -> np.ndarray:
    """``compute_power_density`` (``solver.py:728-745``): p = (sigma*E) . E per face."""
    xy = np.asarray(xy, dtype=DTYPE).reshape(-1, 2)
    values = np.asarray(values, dtype=DTYPE)
    fv = face_vertex_order(tri)
    gx, gy = triangle_gradient(xy[fv[:, 0]], xy[fv[:, 1]], xy[fv[:, 2]],
                               values[fv[:, 0]], values[fv[:, 1]], values[fv[:, 2]])
    jx = gx * conductivity                              # J = E * conductivity
    jy = gy * conductivity
    return jx * gx + jy * gy                            # J.dot(E)


# --------------------------------------------------------------------------
# Restatement of the *device* algorithm on the CPU (test-only): Jacobi-PCG on
# the reduced SPD system, used by the gloo world_size-2 tests to check the
# partition / halo plan and by the parity tests for iteration-level debugging.
# --------------------------------------------------------------------------


def pcg_jacobi(A, b, rtol=1e-12, atol=0.0, max_iter=100000, x0=None, dot=None):
    """Textbook Jacobi-preconditioned CG; ``dot`` lets a caller plug an all-reduce."""
    A = sp.csr_matrix(A)
    dot = dot or (lambda u, v: float(u @ v))
    dinv = 1.0 / A.diagonal()
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    z = dinv * r
    p = z.copy()
    rz = dot(r, z)
    bnorm = np.sqrt(dot(b, b))
    tol = max(rtol * bnorm, atol)
    it = 0
    rnorm = np.sqrt(dot(r, r))
    while rnorm > tol and it < max_iter:
        q = A @ p
        alpha = rz / dot(p, q)
        x += alpha * p
        r -= alpha * q
        z = dinv * r
        rz_new = dot(r, z)
        p = z + (rz_new / rz) * p
        rz = rz_new
        rnorm = np.sqrt(dot(r, r))
        it += 1
    return x, it, rnorm / bnorm if bnorm > 0 else rnorm
