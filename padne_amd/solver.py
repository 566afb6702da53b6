"""MI355X drop-in for the hot path of ``padne/solver.py``.

Same public names, argument meaning and error behaviour as the reference
(``solver.py:24-52, 171-229, 350-615, 671-902``); the arithmetic runs in
``libpadne_hip.so``:

=====================================  ====================================================
reference (file:line)                  here
=====================================  ====================================================
HalfEdge.cotan  mesh.py:124-139        ``asm_fill_tri`` kernel (per triangle, once)
laplace_operator  solver.py:171-213    ``padne_assemble_system`` (sigma = 1, no stamps)
process_mesh_laplace_operators :563    ``padne_assemble_system`` (all meshes in one launch)
stamp_network_into_system :469-541     host emits COO stamps in element order; the device
setup_ground_node :544-560             merges them after the mesh terms, in stamp order
solve_system :767-780 (SuperLU)        reduction to SPD (reduction.py) + Jacobi-PCG kernels,
                                       multipliers recovered from device residual products
produce_layer_solutions :578-615       numpy slice per mesh (contiguous blocks) + power kernel
compute_power_density :728-745         ``power_density_kernel``
=====================================  ====================================================

There is no CPU fallback: every entry point that computes raises
``_hip.HipUnavailableError`` when the library or the GPU is missing.
"""
from __future__ import annotations

import logging
import threading
import warnings
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp
import scipy.spatial

from . import _hip, mesh, problem
from .reduction import (Constraint, KKTLayout, Reduction, SingularSystemError, build_reduction,
                        floating_component_pins, infer_layout)

log = logging.getLogger(__name__)

DTYPE = np.float64

# tolerance of the iterative solve: ||b - A y|| <= RTOL * ||b||   
RTOL = 1e-12
MAX_ITER = 200000

_default = threading.local()        # one context per host thread: a context (stream, pools) is not shared between threads


def get_context() -> _hip.Context:
    """Device context of the calling thread (GPU 0 unless ``set_context`` was called in this thread)."""
    ctx = getattr(_default, "ctx", None)
    if ctx is None:
        ctx = _default.ctx = _hip.Context(0)
    return ctx


def set_context(ctx: Optional[_hip.Context]) -> None:
    _default.ctx = ctx


NEAREST_ON_DEVICE_FROM = 50000     # vertices of a layer from which connections are snapped on the device


class SolverWarning(Warning):
    """Non-fatal oddity of the problem (e.g. non-zero ground current), ``solver.py:24-30``."""


@dataclass(frozen=True)
class SolverInfo:
    ground_node_current: float   # ~0 for a well-posed problem
    residual_norm: float         # ||L v - r||_2 on the ORIGINAL (un-reduced) system
    # extras (not in the reference; default so positional construction stays compatible)
    iterations: int = 0
    rel_residual: float = 0.0
    solve_seconds: float = 0.0


@dataclass
class LayerSolution:
    meshes: list
    potentials: list
    power_densities: list = field(default_factory=list)
    disconnected_meshes: list = field(default_factory=list)


@dataclass
class Solution:
    problem: problem.Problem
    layer_solutions: list
    solver_info: SolverInfo


# --------------------------------------------------------------------------------------------
# index bookkeeping (host): VertexIndexer, NodeIndexer
# --------------------------------------------------------------------------------------------


class VertexIndexer:
    """Global numbering: one contiguous block per mesh, in mesh order (``solver.py:216-229``).

    The reference materialises a list and a dict with one entry per vertex; here the numbering
    is the offset table, and the two containers are built lazily for code that indexes them.
    """

    def __init__(self, sizes: Sequence[int] = ()):
        self.offsets = np.concatenate([[0], np.cumsum(np.asarray(list(sizes), dtype=np.int64))]).astype(np.int64)
        self._g2v = None
        self._v2g = None

    @classmethod
    def create(cls, meshes) -> "VertexIndexer":
        return cls([len(m.vertices) for m in meshes])

    def __len__(self) -> int:
        return int(self.offsets[-1])

    def global_index(self, mesh_idx: int, vertex_idx: int) -> int:
        return int(self.offsets[mesh_idx] + vertex_idx)

    @property
    def global_index_to_vertex_index(self) -> list:
        if self._g2v is None:
            self._g2v = [(m, v) for m in range(len(self.offsets) - 1)
                         for v in range(int(self.offsets[m + 1] - self.offsets[m]))]
        return self._g2v

    @property
    def mesh_vertex_index_to_global_index(self) -> dict:
        if self._v2g is None:
            self._v2g = {mv: g for g, mv in enumerate(self.global_index_to_vertex_index)}
        return self._v2g


@dataclass
class NodeIndexer:
    node_to_global_index: dict = field(default_factory=dict)
    extra_source_to_global_index: dict = field(default_factory=dict)
    internal_node_count: int = 0

    @classmethod
    def create(cls, prob, meshes, mesh_index_to_layer_index, vindex: VertexIndexer,
               filtered_networks) -> "NodeIndexer":
        """``solver.py:398-466``: snap connections to the nearest vertex of their layer, then number
        internal nodes and one current unknown per voltage source / regulator."""
        points = {}
        gidx = {}
        for layer_i in range(len(prob.layers)):
            blocks, ids = [], []
            for mesh_i, msh in enumerate(meshes):
                if mesh_index_to_layer_index[mesh_i] != layer_i or len(msh.vertices) == 0:
                    continue
                blocks.append(msh.points)
                ids.append(np.arange(len(msh.points), dtype=np.int64) + vindex.offsets[mesh_i])
            if not blocks:
                continue
            points[layer_i] = np.concatenate(blocks)
            gidx[layer_i] = np.concatenate(ids)
        # all connections of a layer are snapped together: small layers through a KD-tree exactly like the
        # reference (solver.py:356-396, leafsize=32, k=1); from NEAREST_ON_DEVICE_FROM vertices on, by brute force
        # on the device, where the tree build alone would cost more than the whole linear solve
        wanted = {}
        for network in filtered_networks:
            for conn in network.connections:
                wanted.setdefault(prob.layers.index(conn.layer), []).append((conn.point.x, conn.point.y))
        snapped = {}
        for layer_i, pts in wanted.items():
            q = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
            if len(points[layer_i]) >= NEAREST_ON_DEVICE_FROM:
                k, ties = get_context().nearest_vertex(points[layer_i], q, with_ties=True)
                tied = np.flatnonzero(ties > 1)
                if len(tied):
                    # several vertices at exactly the minimum distance (a connection midway between vertices of a regular
                    # grid): the reference takes whichever its KD-tree meets first (solver.py:389-392, 425).  Only then is
                    # the tree built, and only those queries go through it
                    _, kt = scipy.spatial.KDTree(points[layer_i], leafsize=32).query(q[tied], k=1)
                    k = np.array(k, dtype=np.int64)
                    k[tied] = kt
            else:
                _, k = scipy.spatial.KDTree(points[layer_i], leafsize=32).query(q, k=1)
            snapped[layer_i] = iter(np.asarray(k, dtype=np.int64))
        node_to_global = {}
        for network in filtered_networks:
            for conn in network.connections:
                layer_i = prob.layers.index(conn.layer)
                k = next(snapped[layer_i])
                g = int(gidx[layer_i][k])
                node = conn.node_id
                if node in node_to_global and node_to_global[node] != g:
                    raise ValueError("Duplicate connection vertices found, this should not happen.")
                node_to_global[node] = g
        i_at = len(vindex)
        internal = 0
        for network in filtered_networks:
            for node in network.nodes:
                if node not in node_to_global:
                    node_to_global[node] = i_at
                    i_at += 1
                    internal += 1
        extra = {}
        for network in filtered_networks:
            for elem in network.elements:
                if elem.extra_variable_count > 1:
                    raise NotImplementedError("Extra variable count > 1 not supported yet")
                for _ in range(elem.extra_variable_count):
                    extra[elem] = i_at
                    i_at += 1
        return cls(node_to_global_index=node_to_global, extra_source_to_global_index=extra,
                   internal_node_count=internal)


# --------------------------------------------------------------------------------------------
# stamps: the host only *lists* them; the device adds them up
# --------------------------------------------------------------------------------------------


class StampList:
    """Write-only stand-in for the reference's ``lil_matrix`` during stamping.

    ``L[i, j] += v`` style updates (``solver.py:475-538, 558-560``) are recorded as COO triples in
    the order they were issued; the device merge adds duplicates in that order.
    """

    def __init__(self, n: int):
        self.shape = (n, n)
        self.rows: list = []
        self.cols: list = []
        self.vals: list = []
        self.constraints: list = []          # filled by stamp_network_into_system / setup_ground_node

    def add(self, i: int, j: int, v: float) -> None:
        n = self.shape[0]
        if i < 0:
            i += n
        if j < 0:
            j += n
        self.rows.append(int(i))
        self.cols.append(int(j))
        self.vals.append(float(v))

    def arrays(self):
        return (np.asarray(self.rows, dtype=np.int64), np.asarray(self.cols, dtype=np.int64),
                np.asarray(self.vals, dtype=np.float64))


def _stamp(L, i, j, v):
    if isinstance(L, StampList):
        L.add(i, j, v)
    else:                       # any matrix with item assignment (e.g. a scipy lil_matrix)
        L[i, j] = L[i, j] + v


_ELEMENT_KINDS = ("Resistor", "CurrentSource", "VoltageSource", "VoltageRegulator")


def element_kind(element) -> Optional[str]:
    """Which of the reference's four lumped elements ``element`` is (``problem.py:98-171``), by class *name* along its
    MRO: the seam receives padne's own ``padne.problem`` objects, which are not instances of the
    classes in :mod:`padne_amd.problem`, so an ``isinstance`` test against those would reject every real problem."""
    for cls in type(element).__mro__:
        if cls.__name__ in _ELEMENT_KINDS:
            return cls.__name__
    return None


def stamp_network_into_system(network, node_indexer: NodeIndexer, L, r: np.ndarray) -> None:
    """MNA stamps of one network, same entries in the same order as ``solver.py:469-541``."""
    idx = node_indexer.node_to_global_index
    for element in network.elements:
        kind = element_kind(element)
        if kind == "Resistor":
            a, b = idx[element.a], idx[element.b]
            g = 1 / element.resistance
            _stamp(L, a, a, -g)
            _stamp(L, a, b, g)
            _stamp(L, b, b, -g)
            _stamp(L, b, a, g)
        elif kind == "CurrentSource":
            r[idx[element.f]] += element.current
            r[idx[element.t]] += -element.current
        elif kind == "VoltageSource":
            p, n = idx[element.p], idx[element.n]
            iv = node_indexer.extra_source_to_global_index[element]
            _stamp(L, iv, p, 1.0)
            _stamp(L, iv, n, -1.0)
            r[iv] = element.voltage
            _stamp(L, p, iv, 1.0)
            _stamp(L, n, iv, -1.0)
            if isinstance(L, StampList):
                L.constraints.append(Constraint(index=iv, p=p, n=n, value=float(element.voltage)))
        elif kind == "VoltageRegulator":
            vp, vn = idx[element.v_p], idx[element.v_n]
            sf, st = idx[element.s_f], idx[element.s_t]
            iv = node_indexer.extra_source_to_global_index[element]
            _stamp(L, iv, vp, 1.0)
            _stamp(L, iv, vn, -1.0)
            _stamp(L, vp, iv, 1.0)
            _stamp(L, vn, iv, -1.0)
            r[iv] += element.voltage
            _stamp(L, sf, iv, element.gain)
            _stamp(L, st, iv, -element.gain)
            if isinstance(L, StampList):
                gamma: dict = {}
                gamma[sf] = gamma.get(sf, 0.0) + element.gain
                gamma[st] = gamma.get(st, 0.0) - element.gain
                L.constraints.append(Constraint(index=iv, p=vp, n=vn, value=float(element.voltage),
                                                gamma={k: v for k, v in gamma.items() if v != 0.0}))
        else:
            raise NotImplementedError(f"Unsupported node type {element}")


def setup_ground_node(i_gnd: int, L, r: np.ndarray) -> None:
    """``solver.py:544-560``: last row/column = ground-current unknown."""
    _stamp(L, -1, i_gnd, 1.0)
    _stamp(L, i_gnd, -1, 1.0)
    r[-1] = 0
    if isinstance(L, StampList):
        L.constraints.append(Constraint(index=L.shape[0] - 1, p=int(i_gnd), n=-1, value=0.0))


def find_best_ground_node_index(prob, node_indexer: NodeIndexer) -> int:
    """``solver.py:671-686``: the ``n`` terminal of the highest-voltage source, else unknown 0."""
    best, ground = float("-inf"), 0
    for network in prob.networks:
        for element in network.elements:
            if element_kind(element) == "VoltageSource" and element.voltage > best:
                best = element.voltage
                ground = node_indexer.node_to_global_index[element.n]
    return ground


def allocate_system(vindex: VertexIndexer, node_indexer: NodeIndexer):
    """``solver.py:748-764``: N = vertices + internal nodes + extra currents + 1 (ground)."""
    N = len(vindex) + node_indexer.internal_node_count + len(node_indexer.extra_source_to_global_index) + 1
    log.info(f"System matrix size: {N}x{N} variables")
    return StampList(N), np.zeros(N, dtype=DTYPE)


# --------------------------------------------------------------------------------------------
# device-resident system matrix
# --------------------------------------------------------------------------------------------


class SystemMatrix:
    """The assembled ``L`` (reference layout and sign), resident on the GPU.

    Quacks enough like the ``lil_matrix`` the reference returns from ``assemble_system`` for the
    callers on the seam: ``shape``, ``tocsr()/tocsc()/tolil()/toarray()/todense()``, ``L[i, j]``,
    ``L @ v``.  Carries the KKT layout so that ``solve_system`` does not have to re-derive it.
    """

    def __init__(self, dev: _hip.CsrMatrix, layout: Optional[KKTLayout], xy=None, tri=None, mesh_offsets=None,
                 links=None):
        self.dev = dev
        self.layout = layout
        self.shape = dev.shape
        self._host = None
        # geometry of the mesh unknowns (optional): lets solve_system pick a cache-friendly internal ordering
        self.xy, self.tri, self.mesh_offsets = xy, tri, mesh_offsets
        # pairs of potentials coupled by lumped stamps (optional): lets solve_system find floating copper
        self.links = links
        self._plans = {}            # device plans of solve_system by the structure of the reduction (see there)

    @property
    def nnz(self) -> int:
        return self.dev.nnz

    def close(self):
        """Release the device memory of the system and of the solve plans kept with it."""
        for plan in self._plans.values():
            plan.close()
        self._plans.clear()
        self.dev.close()

    def tocsr(self):
        if self._host is None:
            self._host = self.dev.to_scipy()
        return self._host

    def tocsc(self):
        return self.tocsr().tocsc()

    def tolil(self):
        return self.tocsr().tolil()

    def tocoo(self):
        return self.tocsr().tocoo()

    def toarray(self):
        return self.tocsr().toarray()

    def todense(self):
        return self.tocsr().todense()

    def __getitem__(self, key):
        return self.tocsr()[key]

    def __matmul__(self, v):
        return self.dev.matvec(np.asarray(v, dtype=DTYPE))


def _flatten_meshes(meshes, conductances):
    xy = np.concatenate([m.points for m in meshes]) if meshes else np.zeros((0, 2))
    tri = np.concatenate([m.triangles for m in meshes]) if meshes else np.zeros((0, 3), np.int32)
    mvo = np.concatenate([[0], np.cumsum([len(m.points) for m in meshes])]).astype(np.int64)
    mto = np.concatenate([[0], np.cumsum([len(m.triangles) for m in meshes])]).astype(np.int64)
    return xy, tri, mvo, mto, np.asarray(conductances, dtype=np.float64)


def laplace_operator(msh: mesh.Mesh) -> sp.coo_matrix:
    """Mesh-local cotangent Laplacian (``solver.py:171-213``), computed on the device."""
    ctx = get_context()
    n = len(msh.vertices)
    xy, tri, mvo, mto, sig = _flatten_meshes([msh], [1.0])
    empty = np.zeros(0, dtype=np.int64)
    dev = ctx.assemble_system(n, xy, tri, mvo, mto, sig, empty, empty, np.zeros(0))
    out = dev.to_scipy().tocoo()
    dev.close()
    return out


def process_mesh_laplace_operators(meshes, conductances, vindex: VertexIndexer, L) -> None:
    """``solver.py:563-575``.  With a StampList the mesh terms are not listed at all: the device
    assembles them straight from the triangles (see ``assemble_system``); this function exists for
    callers that stamp into a host matrix."""
    if isinstance(L, StampList):
        L.meshes = (list(meshes), list(conductances))
        return
    for mesh_i, (msh, conductance) in enumerate(zip(meshes, conductances)):
        L_msh = (conductance * laplace_operator(msh)).tocoo()
        off = int(vindex.offsets[mesh_i])
        for i, j, v in zip(L_msh.row, L_msh.col, L_msh.data):
            L[off + i, off + j] += v


def assemble_system(prob, meshes, mesh_index_to_layer_index, vindex: VertexIndexer, filtered_networks,
                    node_indexer: NodeIndexer):
    """``solver.py:783-812``: allocate, mesh Laplacians, network stamps, ground -> ``(L, r)``.

    ``L`` is a :class:`SystemMatrix` on the device (convertible with ``.tocsr()`` / ``.tolil()``)."""
    conductances = [prob.layers[mesh_index_to_layer_index[i]].conductance for i in range(len(meshes))]
    stamps, r = allocate_system(vindex, node_indexer)
    for network in filtered_networks:
        stamp_network_into_system(network, node_indexer, stamps, r)
    setup_ground_node(find_best_ground_node_index(prob, node_indexer), stamps, r)
    L = assemble_from_arrays(meshes, conductances, stamps, n_potential=len(vindex) + node_indexer.internal_node_count)
    return L, r


def assemble_from_arrays(meshes, conductances, stamps: StampList, n_potential: int) -> SystemMatrix:
    ctx = get_context()
    xy, tri, mvo, mto, sig = _flatten_meshes(meshes, conductances)
    rows, cols, vals = stamps.arrays()
    dev = ctx.assemble_system(stamps.shape[0], xy, tri, mvo, mto, sig, rows, cols, vals)
    layout = KKTLayout(size=stamps.shape[0], n_potential=n_potential, constraints=list(stamps.constraints))
    off = (rows < cols) & (cols < n_potential)
    links = np.unique(np.stack([rows[off], cols[off]], axis=1), axis=0) if off.any() else np.zeros((0, 2), np.int64)
    return SystemMatrix(dev, layout, xy=xy, tri=tri, mesh_offsets=mvo, links=links)


# --------------------------------------------------------------------------------------------
# solve
# --------------------------------------------------------------------------------------------


# The reference judges a solve by the ABSOLUTE residual of the whole system, ||L v - r||_2 < 1e-9
# (tests/test_solver.py:2083-2089).  The rows of the reduced system are rows of that system (the multiplier rows
# hold exactly by construction), so its residual is driven to a quarter of that bar whenever 1e-12 ||b|| is looser
# than that -- a voltage source across a copper plane puts kiloamperes into b.  Below RTOL_FLOOR nothing is gained:
# that is what evaluating b - A y in binary64 can resolve (the solve stops at that floor and reports what it reached).
# The rule itself is applied where ||b|| is known: on the device (padne_kkt_solve, abs_residual_target), and from the
# globally reduced norm in the row-partitioned path (distributed.solve_partitioned).
ABS_RESIDUAL_TARGET = 2.5e-10
RTOL_FLOOR = 2e-15


STALL_WARN_ABOVE = 1e-9


def _warn_if_stalled(res, rtol: float) -> None:
    """The reference's direct solve always returns *an* answer and reports its quality through
    SolverInfo.residual_norm; an iteration that stalls above the requested tolerance does the same, with a warning in
    the reference's own soft-failure style (solver.py:880-888) when the residual is worse than 1e-9 relative (matrices
    with entry ratios beyond 1e12, e.g. needle triangles).  A stall between the requested 1e-12 and 1e-9 is the
    rounding floor of evaluating b - A x for a system whose solution is large against its right-hand side (a layer
    held at hundreds of volts through a weak link: 1 of 1000 random systems of scripts/fuzz_parity.py, potentials
    still within 4e-11 of the direct solve); it is reported in SolverInfo.residual_norm and not warned about."""
    if res.status != _hip.OK and not res.rel_residual <= max(rtol, STALL_WARN_ABOVE):
        warnings.warn(f"iterative solve stopped at a relative residual of {res.rel_residual:.2e} "
                      f"(requested {rtol:.1e}) after {res.iterations} iterations", SolverWarning)


def solve_system(L, r: np.ndarray, *, rtol: float = RTOL, reorder=None, n_potential: Optional[int] = None):
    """Solve ``L v = r`` and return ``(v, SolverInfo)`` like ``solver.py:767-780``.

    ``L`` is a :class:`SystemMatrix` from :func:`assemble_system`, or any scipy sparse matrix in the
    reference's layout (it is uploaded and its multiplier structure inferred; ``n_potential`` = number of
    potential unknowns in front of the multiplier block, if the caller knows it).  ``reorder``: None = solve in
    a band numbering when the mesh numbering is scattered (decided from the triangles), True / False = force.

    Copper that nothing ties to the ground node (the reference's matrix is singular there, its LU returns rounding
    noise for those potentials) is held at 0 V at one vertex; ``ground_node_current`` is, as in the reference, the net
    current injected into the grounded component (``tests/test_solver.py:1829-1833``: non-zero for an unterminated
    current loop).
    """
    ctx = get_context()
    r = np.ascontiguousarray(r, dtype=DTYPE)
    if isinstance(L, SystemMatrix):
        dev, layout = L.dev, L.layout
        owned = False
        Lc = None
    else:
        Lc = sp.csr_matrix(L)
        Lc.sum_duplicates()
        Lc.eliminate_zeros()
        Lc.sort_indices()
        dev, layout, owned = ctx.csr_from_scipy(Lc), None, True
        layout = infer_layout(Lc, r, n_potential)
    if layout is None or not layout.constraints:
        raise SingularSystemError("system has no ground constraint")
    # multiplier rows take their right-hand side from r (solver.py:505, 530, 560)
    for cst in layout.constraints:
        cst.value = float(r[cst.index])
    ground_p = layout.ground_constraint.p
    if Lc is not None:
        pins = floating_component_pins(layout.n_potential, ground_p, layout.constraints, matrix=Lc)
    elif L.links is not None and L.mesh_offsets is not None:
        pins = floating_component_pins(layout.n_potential, ground_p, layout.constraints,
                                       mesh_offsets=L.mesh_offsets, links=L.links)
    else:
        pins = []
    if pins:
        log.info(f"{len(pins)} floating component(s) held at 0 V at unknown(s) {pins[:8]}")
    # O(#constraints): what the reduction eliminates, ties and knows; the index map itself is made on the device
    red: Reduction = build_reduction(layout, pins)
    want_reorder = False
    if isinstance(L, SystemMatrix) and L.xy is not None and reorder is not False:
        # CGAL numbers vertices in insertion order; when neighbours are far apart in the numbering the SpMV
        # gathers miss the caches, so the reduced system is solved in a band numbering by horizontal strips
        # (internal: v comes back unpermuted)
        from .reduction import ordering_is_scattered
        if reorder is True:
            want_reorder = True
        else:
            # (a property of the mesh, not of the right-hand side: looked at once per assembled system -- 4 ms of a 34 ms
            #  call at 20 M triangles otherwise)
            if getattr(L, "_scattered", None) is None:
                L._scattered = bool(ordering_is_scattered(L.tri, len(L.xy)))
            want_reorder = L._scattered
    # the plan -- index map, A = -P^T L P and its multigrid hierarchy on the device -- depends on the STRUCTURE of the
    # reduction only (the values of the sources enter through c and r): kept with the assembled system, so a second
    # right-hand side on the same system finds everything in place, like a second solve with a kept factorisation
    key = (red.elim.tobytes(), tuple(red.tied), bool(want_reorder))
    plan = L._plans.get(key) if isinstance(L, SystemMatrix) else None
    if plan is None:
        try:
            if want_reorder:
                # the strip numbering is made on the device from the mesh that was assembled (padne_kkt_create, flags bit 0);
                # key fields that do not fit (65535 meshes / strips) fall back to the host's sort of the same keys -- on THAT
                # refusal only: any other invalid argument is the caller's error and is raised as it is
                try:
                    plan = _hip.KktPlan(dev, layout.n_potential, red.elim, red.tied, red.n_free, strip_order=True)
                except ValueError as exc:
                    if "the host orders this system" not in str(exc) and "beyond the key fields" not in str(exc):
                        raise
                    from .reduction import apply_locality_ordering
                    apply_locality_ordering(red, L.xy, L.mesh_offsets)
                    plan = _hip.KktPlan(dev, layout.n_potential, red.elim, red.tied, red.n_free, index_map=red.index_map)
            else:
                plan = _hip.KktPlan(dev, layout.n_potential, red.elim, red.tied, red.n_free)
        except BaseException:
            if owned:                                 # a scipy matrix uploaded for this call: it must not outlive a failed plan
                dev.close()
            raise
        if isinstance(L, SystemMatrix):
            for old in L._plans.values():             # one structure at a time: a plan holds GBs at N = 10 M
                old.close()
            L._plans.clear()
            L._plans[key] = plan
    try:
        members = sorted({int(x) for mem, cons, _ in red.groups if cons for x in mem})
        probes, res = plan.solve(r, red.known, [dict(cst.gamma) for cst in red.regulators], members, rtol=rtol,
                                 max_iter=MAX_ITER, abs_residual_target=ABS_RESIDUAL_TARGET)
        _warn_if_stalled(res, rtol)
        at = {x: k for k, x in enumerate(members)}

        class _Rows:                                   # rho_x of a member x, for Reduction.multipliers
            def __init__(self, i_vec):
                self.i_vec = i_vec

            def __getitem__(self, x):
                k = at[int(x)]
                # rho(v + sum_k i_k Z_k) = rho(v) - sum_k i_k (L Z_k): the device returned both at the members
                return probes[0, k] - sum(self.i_vec[j] * probes[1 + j, k] for j in range(len(self.i_vec)))
        K = len(red.regulators)
        keys = [cst.index for cst in red.regulators]
        i_reg = np.zeros(K)
        if K:
            # y = y0 + sum_k i_k z_k with A z_k = P^T gamma_k:  row x reads L_x.v + gamma_k[x] i_k = r_x, so
            # summing a group's rows gives  -P^T L P y = -P^T (r - L c) + sum_k i_k P^T gamma_k
            def currents_for(i_vec):
                return red.multipliers(_Rows(i_vec), dict(zip(keys, i_vec)))
            base = currents_for(np.zeros(K))
            F0 = np.array([base[k] for k in keys])
            J = np.zeros((K, K))
            for k in range(K):
                e = np.zeros(K)
                e[k] = 1.0
                ck = currents_for(e)
                J[:, k] = np.array([ck[q] for q in keys]) - F0
            i_reg = np.linalg.solve(np.eye(K) - J, F0)
        mult_known = dict(zip(keys, i_reg))
        mult = {idx: val for idx, val in red.multipliers(_Rows(i_reg), mult_known).items() if idx >= 0}
        # (negative index: the current through the pin of a floating component, not an unknown of the system)
        v, residual_norm = plan.finish(i_reg, mult)
    finally:
        if owned:
            plan.close()
            dev.close()
    info = SolverInfo(ground_node_current=float(v[-1]), residual_norm=float(residual_norm),
                      iterations=int(res.iterations), rel_residual=float(res.rel_residual), solve_seconds=float(res.seconds))
    return v, info


# --------------------------------------------------------------------------------------------
# post-processing
# --------------------------------------------------------------------------------------------


def compute_triangle_gradient(vertices, values) -> mesh.Vector:
    """Gradient of the linear interpolant on one triangle (``solver.py:689-725``), on the device."""
    if len(vertices) != 3 or len(values) != 3:
        raise ValueError("Vertices and values must be of length 3 for a triangle")
    ctx = get_context()
    # the kernel visits a face as (tri[2], tri[0], tri[1]); feed (v2, v3, v1) so it sees (v1, v2, v3)
    v1, v2, v3 = vertices
    xy = np.array([[v2.p.x, v2.p.y], [v3.p.x, v3.p.y], [v1.p.x, v1.p.y]], dtype=DTYPE)
    pot = np.array([values[1], values[2], values[0]], dtype=DTYPE)
    gx, gy = ctx_face_gradient(ctx, xy, np.array([[0, 1, 2]], np.int32), pot)
    return mesh.Vector(float(gx[0]), float(gy[0]))


def ctx_face_gradient(ctx, xy, tri, pot):
    one = np.array([0, len(xy)], dtype=np.int64)
    onet = np.array([0, len(tri)], dtype=np.int64)
    return ctx.face_gradient(xy, tri, one, onet, pot)


def compute_power_density(voltage: mesh.ZeroForm, conductivity: float) -> mesh.TwoForm:
    """Per-face ``sigma |grad V|^2`` (``solver.py:728-745``)."""
    ctx = get_context()
    msh = voltage.mesh
    out = mesh.TwoForm(msh)
    if len(msh.triangles):
        xy, tri, mvo, mto, sig = _flatten_meshes([msh], [conductivity])
        out.values = ctx.power_density(xy, tri, mvo, mto, sig, voltage.values)
    return out


def produce_layer_solutions(layers, vindex: VertexIndexer, meshes, mesh_index_to_layer_index, v: np.ndarray,
                            disconnected_meshes_by_layer, system: Optional["SystemMatrix"] = None) -> list:
    """``solver.py:578-615``.  Each mesh's unknowns are one contiguous block of ``v``, so the scatter
    is a slice; the power densities of all meshes come from one kernel launch."""
    ctx = get_context()
    sig = [layers[mesh_index_to_layer_index[i]].conductance for i in range(len(meshes))]
    power_all = None
    n_tri = sum(len(m.triangles) for m in meshes)
    if meshes and n_tri:
        if system is not None and system.tri is not None and len(system.tri) == n_tri:
            # the system was assembled from these meshes: they are still on the device, only the potentials travel
            power_all = system.dev.power_density(v[:len(vindex)], n_tri)
        else:
            xy, tri, mvo, mto, sg = _flatten_meshes(meshes, sig)
            power_all = ctx.power_density(xy, tri, mvo, mto, sg, v[:len(vindex)])
    toff = np.concatenate([[0], np.cumsum([len(m.triangles) for m in meshes])]).astype(np.int64)
    out = []
    for layer_i, _layer in enumerate(layers):
        sol = LayerSolution(meshes=[], potentials=[], power_densities=[],
                            disconnected_meshes=disconnected_meshes_by_layer[layer_i])
        for mesh_i, msh in enumerate(meshes):
            if mesh_index_to_layer_index[mesh_i] != layer_i:
                continue
            zf = mesh.ZeroForm(msh)
            zf.values = np.array(v[vindex.offsets[mesh_i]:vindex.offsets[mesh_i + 1]], dtype=DTYPE)
            tf = mesh.TwoForm(msh)
            if power_all is not None:
                tf.values = np.array(power_all[toff[mesh_i]:toff[mesh_i + 1]], dtype=DTYPE)
            sol.meshes.append(msh)
            sol.potentials.append(zf)
            sol.power_densities.append(tf)
        out.append(sol)
    return out


# --------------------------------------------------------------------------------------------
# orchestration
# --------------------------------------------------------------------------------------------


def _solve_partitioned(prob, meshes, mesh_index_to_layer_index, vindex, filtered_networks, node_indexer, partition,
                       ctx):
    """The solve of ``solve_meshed`` with the rows dealt to several GPUs (``distributed.py``): every rank lists the
    same stamps, assembles and solves its own rows, and all ranks end up with all potentials."""
    from . import distributed
    conductances = [prob.layers[mesh_index_to_layer_index[i]].conductance for i in range(len(meshes))]
    stamps, r = allocate_system(vindex, node_indexer)
    for network in filtered_networks:
        stamp_network_into_system(network, node_indexer, stamps, r)
    setup_ground_node(find_best_ground_node_index(prob, node_indexer), stamps, r)
    n_pot = len(vindex) + node_indexer.internal_node_count
    plan = distributed.build_problem_partition(meshes, conductances, list(mesh_index_to_layer_index), stamps, r, n_pot,
                                               partition.rank, partition.world)
    v_pot, res = distributed.solve_partitioned(plan, ctx, dist=partition.dist, team=partition.team, rtol=RTOL,
                                               gather=partition.gather)
    v = np.zeros(stamps.shape[0], dtype=DTYPE)
    if plan.reduction is not None:
        # sources or regulators: the whole solution vector came back -- potentials and the multiplier currents recovered
        # from the residual rows of the source-tied unknowns (distributed.solve_partitioned)
        v[:] = v_pot[:len(v)]
    else:
        v[:n_pot] = v_pot[:n_pot]
        # KCL over all potential rows: the mesh and resistor terms cancel, what is left is the ground current (row of
        # solver.py:558-560) = the net current the sources inject
        v[-1] = float(np.sum(r[:n_pot]))
    info = SolverInfo(ground_node_current=float(v[-1]), residual_norm=float(res.abs_residual),
                      iterations=int(res.iterations), rel_residual=float(res.rel_residual), solve_seconds=float(res.seconds))
    return v, info


def solve_meshed(prob, meshes, mesh_index_to_layer_index, *, filtered_networks=None,
                 disconnected_meshes_by_layer=None, partition=None) -> Solution:
    """Steps 4-11 of the reference's ``solve()`` (``solver.py:846-902``): everything after meshing.

    ``partition``: a :class:`padne_amd.distributed.Partition` -- the rows are dealt to the GPUs of the node (by layer, or
    by strips of layers when there are fewer layers than GPUs); every rank calls this with the same Problem and gets the
    same Solution."""
    meshes = [m if isinstance(m, mesh.Mesh) else mesh.Mesh.from_reference(m) for m in meshes]
    if filtered_networks is None:
        filtered_networks = list(prob.networks)
    if disconnected_meshes_by_layer is None:
        disconnected_meshes_by_layer = [[] for _ in prob.layers]
    log.info("Indexing vertices and connections")
    vindex = VertexIndexer.create(meshes)
    node_indexer = NodeIndexer.create(prob, meshes, mesh_index_to_layer_index, vindex, filtered_networks)
    if partition is not None and partition.world > 1:
        ctx = get_context()
        v, solver_info = _solve_partitioned(prob, meshes, mesh_index_to_layer_index, vindex, filtered_networks,
                                            node_indexer, partition, ctx)
        if not np.isclose(solver_info.ground_node_current, 0):
            warnings.warn(
                f"Ground node current is not zero ({solver_info.ground_node_current} A), this may indicate an issue "
                "with the problem being solved. Check for unterminated current loops or floating connected "
                "components. This may be harmless if the current is small, but it may indicate an "
                "ill-conditioned system.", SolverWarning)
        layer_solutions = produce_layer_solutions(prob.layers, vindex, meshes, mesh_index_to_layer_index, v,
                                                  disconnected_meshes_by_layer)
        return Solution(problem=prob, layer_solutions=layer_solutions, solver_info=solver_info)
    log.info("Assembling the global system")
    L, r = assemble_system(prob, meshes, mesh_index_to_layer_index, vindex, filtered_networks, node_indexer)
    log.info("Solving the system of equations")
    try:
        v, solver_info = solve_system(L, r)
    except BaseException:
        L.close()
        raise
    if not np.isclose(solver_info.ground_node_current, 0):
        warnings.warn(
            f"Ground node current is not zero ({solver_info.ground_node_current} A), this may indicate an issue "
            "with the problem being solved. Check for unterminated current loops or floating connected "
            "components. This may be harmless if the current is small, but it may indicate an "
            "ill-conditioned system.", SolverWarning)
    log.info("Producing the solution object")
    try:
        # the mesh is still on the device with the assembled system: the power densities need only the potentials
        layer_solutions = produce_layer_solutions(prob.layers, vindex, meshes, mesh_index_to_layer_index, v,
                                                  disconnected_meshes_by_layer, system=L)
    finally:
        L.close()
    return Solution(problem=prob, layer_solutions=layer_solutions, solver_info=solver_info)


def solve(prob, mesher_config: Optional[mesh.Mesher.Config] = None, *, mesher=None, partition=None) -> Solution:
    """``padne.solver.solve`` (``solver.py:815-902``).

    Meshing and the geometric connectivity pre-pass are out of scope (CGAL / shapely).  ``mesher``
    must offer ``poly_to_mesh(polygon, seed_points) -> Mesh`` (padne's own ``mesh.Mesher`` does, and
    so does :class:`padne_amd.structured.StructuredMesher` for rectangles and annuli); every
    polygon of every layer is meshed and treated as connected.
    """
    if mesher is None:
        mesher = mesh.Mesher(mesher_config)
    meshes, mesh_index_to_layer_index = [], []
    log.info("Meshing the connected components")
    for layer_i, layer in enumerate(prob.layers):
        seeds = [mesh.Point(c.point.x, c.point.y) for net in prob.networks for c in net.connections
                 if c.layer is layer or c.layer == layer]
        for geom in layer.geoms:
            meshes.append(mesher.poly_to_mesh(geom, seeds))
            mesh_index_to_layer_index.append(layer_i)
    return solve_meshed(prob, meshes, mesh_index_to_layer_index, partition=partition)
