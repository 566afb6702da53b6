"""Shared helpers for the test-suite: golden fixtures <-> oracle / product inputs."""
from __future__ import annotations

import glob
import os

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
KIND = {0: "R", 1: "I", 2: "V", 3: "REG"}
NARGS = {"R": 3, "I": 3, "V": 4, "REG": 7}
INT_ARGS = {"R": 2, "I": 2, "V": 2, "REG": 4}


def golden_names():
    """Unknown-level fixtures (lumped elements given by global unknown index)."""
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
                  if not n.startswith(("problem_", "direct_")))


def problem_golden_names():
    """Problem-level fixtures (layers, connections by coordinates, networks): the reference's whole post-meshing path."""
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "problem_*.npz")))


PKIND = {0: "R", 1: "I", 2: "V", 3: "REG"}
PTERMS = {"R": 2, "I": 2, "V": 2, "REG": 4}


def problem_networks(g):
    """Networks of a problem-level fixture as the oracle's tables: dict(connections=[(layer, x, y, node)],
    elements=[(kind, terminal nodes..., values...)])."""
    n_net = int(max(g["connections"][:, 0].max(initial=-1), g["pelements"][:, 0].max(initial=-1))) + 1
    nets = [dict(connections=[], elements=[]) for _ in range(n_net)]
    for ni, layer, x, y, node in g["connections"]:
        nets[int(ni)]["connections"].append((int(layer), float(x), float(y), int(node)))
    for row in g["pelements"]:
        kind = PKIND[int(row[1])]
        nt = PTERMS[kind]
        terms = tuple(int(t) for t in row[2:2 + nt])
        vals = (float(row[6]),) if kind != "REG" else (float(row[6]), float(row[7]))
        nets[int(row[0])]["elements"].append((kind, *terms, *vals))
    return nets


def problem_meshes(g):
    return [(g[f"xy{i}"], g[f"tri{i}"], int(g[f"layer{i}"])) for i in range(int(g["n_mesh"]))]


class XY:
    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)


class Geoms:
    def __init__(self, n=1):
        self.geoms = tuple(object() for _ in range(n))


def build_problem(g, P):
    """The Problem of a problem-level fixture from the classes of module ``P`` (padne_amd.problem, the reference's
    padne.problem, or any look-alike).  Returns (problem, networks, elements in stamping order)."""
    layers = [P.Layer(shape=Geoms(1), name=f"L{i}", conductance=float(s)) for i, s in enumerate(g["layer_sigma"])]
    nodes, flat, networks = {}, [], []
    for net in problem_networks(g):
        conns = []
        for layer, x, y, node in net["connections"]:
            c = P.Connection(layer=layers[layer], point=XY(x, y))
            nodes[node] = c.node_id
            conns.append(c)
        els = []
        for e in net["elements"]:
            t = [nodes.setdefault(k, P.NodeID()) for k in e[1:1 + PTERMS[e[0]]]]
            if e[0] == "R":
                el = P.Resistor(a=t[0], b=t[1], resistance=e[3])
            elif e[0] == "I":
                el = P.CurrentSource(f=t[0], t=t[1], current=e[3])
            elif e[0] == "V":
                el = P.VoltageSource(p=t[0], n=t[1], voltage=e[3])
            else:
                el = P.VoltageRegulator(v_p=t[0], v_n=t[1], s_f=t[2], s_t=t[3], voltage=e[5], gain=e[6])
            els.append(el)
            flat.append(el)
        networks.append(P.Network(connections=conns, elements=els))
    return P.Problem(layers=layers, networks=networks), nodes, flat


class HalfEdgeLikeMesh:
    """Duck-typed stand-in for the reference's half-edge ``padne.mesh.Mesh`` as ``solve()`` hands it on
    (``mesh.py:778-786``): ``vertices`` = objects with ``.p.x``, ``.p.y`` and ``.i``; ``faces`` = objects whose
    ``.vertices`` iterate in the order the reference visits a face, (v3, v1, v2) (``mesh.py:320-325``).  No ``points`` /
    ``triangles`` arrays: ``Mesh.from_reference`` has to walk the objects.  ``soup``: what the mesher stub of
    INTEGRATION.md leaves next to the half-edge mesh (``_padne_hip_soup``: the CGAL output dict)."""

    class _V:
        __slots__ = ("p", "i")

        def __init__(self, x, y, i):
            self.p, self.i = XY(x, y), i

    class _F:
        __slots__ = ("vertices",)

        def __init__(self, vs):
            self.vertices = vs

    def __init__(self, xy, tri, soup=None):
        self.vertices = [self._V(float(x), float(y), i) for i, (x, y) in enumerate(xy)]
        self.faces = [self._F((self.vertices[c], self.vertices[a], self.vertices[b])) for a, b, c in
                      (tuple(int(k) for k in t) for t in tri)]
        if soup is not None:
            self._padne_hip_soup = soup


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))


def elements_of(g):
    out = []
    for row in g["elements"]:
        kind = KIND[int(row[0])]
        args = list(row[1:1 + NARGS[kind]])
        ni = INT_ARGS[kind]
        vals = [int(a) for a in args[:ni]] + [float(a) for a in args[ni:]]
        if kind in ("V", "REG"):
            vals[-1] = int(vals[-1])
        out.append((kind, *vals))
    return out


def meshes_of(g):
    return [(g[f"xy{i}"], g[f"tri{i}"], float(g[f"sigma{i}"]), int(g[f"layer{i}"])) for i in range(int(g["n_mesh"]))]


def golden_L(g):
    N = int(g["N"])
    return sp.csr_matrix((g["L_data"], g["L_indices"].astype(np.int32), g["L_indptr"].astype(np.int32)), shape=(N, N))


def golden_lap(g, i):
    n = len(g[f"xy{i}"])
    return sp.csr_matrix((g[f"lap{i}_data"], g[f"lap{i}_indices"].astype(np.int32), g[f"lap{i}_indptr"].astype(np.int32)),
                         shape=(n, n))


def same_structure(A, B):
    A = sp.csr_matrix(A)
    B = sp.csr_matrix(B)
    A.sort_indices()
    B.sort_indices()
    return A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)


def offdiag_and_diag(A):
    A = sp.csr_matrix(A)
    d = A.diagonal()
    off = A - sp.diags(d)
    off.eliminate_zeros()
    return off.tocsr(), d


# ---- product-side construction from a golden spec ------------------------------------------------

def product_system(g):
    """Build (meshes, conductances, StampList, r, n_potential) the way padne_amd.solver would."""
    from padne_amd import mesh as pmesh, solver
    ms = meshes_of(g)
    meshes = [pmesh.Mesh(xy, tri) for xy, tri, _, _ in ms]
    sig = [m[2] for m in ms]
    n_vert = sum(len(m.points) for m in meshes)
    n_pot = n_vert + int(g["n_internal"])
    N = int(g["N"])
    stamps = solver.StampList(N)
    r = np.zeros(N)
    from padne_amd.reduction import Constraint
    for e in elements_of(g):
        k = e[0]
        if k == "R":
            _, a, b, res = e
            gg = 1 / res
            stamps.add(a, a, -gg); stamps.add(a, b, gg); stamps.add(b, b, -gg); stamps.add(b, a, gg)
        elif k == "I":
            _, f, t, cur = e
            r[f] += cur
            r[t] += -cur
        elif k == "V":
            _, p, n, u, iv = e
            stamps.add(iv, p, 1.0); stamps.add(iv, n, -1.0); r[iv] = u
            stamps.add(p, iv, 1.0); stamps.add(n, iv, -1.0)
            stamps.constraints.append(Constraint(index=iv, p=p, n=n, value=u))
        elif k == "REG":
            _, vp, vn, sf, st, u, gain, iv = e
            stamps.add(iv, vp, 1.0); stamps.add(iv, vn, -1.0); stamps.add(vp, iv, 1.0); stamps.add(vn, iv, -1.0)
            r[iv] += u
            stamps.add(sf, iv, gain); stamps.add(st, iv, -gain)
            stamps.constraints.append(Constraint(index=iv, p=vp, n=vn, value=u, gamma={sf: gain, st: -gain}))
    solver.setup_ground_node(int(g["ground"]), stamps, r)
    return meshes, sig, stamps, r, n_pot


def random_csr(n_rows, n_cols, per_row, seed):
    """Random CSR with ~per_row sorted, unique column indices per row (cheap for any shape)."""
    rng = np.random.default_rng(seed)
    k = min(per_row, n_cols)
    lens = rng.integers(0, 2 * k + 1, size=n_rows).clip(0, n_cols)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = rng.integers(0, n_cols, size=int(indptr[-1]))
    rows = np.repeat(np.arange(n_rows), lens)
    key = np.unique(rows.astype(np.int64) * n_cols + cols)
    rows, cols = key // n_cols, key % n_cols
    data = rng.uniform(-1, 1, size=len(key))
    return sp.csr_matrix((data, (rows, cols)), shape=(n_rows, n_cols))
