"""Deterministic synthetic inputs for the BASELINE configs.

These stand in for the reference's CGAL mesher (``padne/mesh.py:662-795``,
``padne/cpp/_cgal.cpp`` -- out of scope) so that the hot path can be fed
benchmark-scale triangle soups: jittered structured triangulations with
7 non-zeros per interior row, stacked into layers and stitched by via resistor
rings the way ``kicad.process_via_spec`` does (``kicad.py:1497-1585``).

Everything here is plain index bookkeeping on numpy arrays; the arithmetic of
the path (cotangent weights, stamping merge, solve) happens on the device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

COPPER_CONDUCTIVITY = 5.95e4      # S/mm   (kicad.py:79)
DEFAULT_THICKNESS = 0.035         # mm
DEFAULT_SHEET_CONDUCTANCE = COPPER_CONDUCTIVITY * DEFAULT_THICKNESS  # 2082.5 S
DEFAULT_MAXIMUM_SIZE = 0.6        # mm     (mesh.py:672)


def jittered_grid(nx: int, ny: int, h: float = DEFAULT_MAXIMUM_SIZE, seed: int = 0,
                  jitter: float = 0.2, origin=(0.0, 0.0)):
    """nx*ny vertices, row-major, cells split by alternating diagonals, CCW triangles.

    Interior vertices are displaced by U(-jitter*h, +jitter*h) so that no
    cotangent weight is exactly zero (a right-angle corner would drop the
    diagonal edge, ``solver.py:187-190``) -> 7 nnz per interior row.
    """
    if nx < 2 or ny < 2:
        raise ValueError("grid needs at least 2x2 vertices")
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    xy = np.empty((nx * ny, 2), dtype=np.float64)
    xy[:, 0] = origin[0] + ix.reshape(-1) * h
    xy[:, 1] = origin[1] + iy.reshape(-1) * h
    if jitter:
        rng = np.random.default_rng(seed)
        d = rng.uniform(-jitter * h, jitter * h, size=(ny, nx, 2))
        d[0, :, :] = 0.0
        d[-1, :, :] = 0.0
        d[:, 0, :] = 0.0
        d[:, -1, :] = 0.0
        xy += d.reshape(-1, 2)
    cx, cy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    cx = cx.reshape(-1)
    cy = cy.reshape(-1)
    v00 = (cy * nx + cx).astype(np.int32)
    v10 = v00 + 1
    v01 = v00 + nx
    v11 = v01 + 1
    even = ((cx + cy) & 1) == 0
    tri = np.empty((cx.size, 2, 3), dtype=np.int32)
    # even cells: diagonal v00-v11 ; odd cells: diagonal v10-v01
    tri[:, 0, 0] = v00
    tri[:, 0, 1] = v10
    tri[:, 0, 2] = np.where(even, v11, v01)
    tri[:, 1, 0] = np.where(even, v00, v10)
    tri[:, 1, 1] = v11
    tri[:, 1, 2] = v01
    return xy, tri.reshape(-1, 3)


def annulus_mesh(r_in: float, r_out: float, n_r: int, n_theta: int):
    """Structured polar triangulation of an annulus (coaxial test, test_solver.py:597-751)."""
    rr = np.linspace(r_in, r_out, n_r)
    th = np.arange(n_theta) * (2 * math.pi / n_theta)
    R, T = np.meshgrid(rr, th, indexing="ij")
    xy = np.stack([R.reshape(-1) * np.cos(T.reshape(-1)), R.reshape(-1) * np.sin(T.reshape(-1))], axis=1)
    tris = []
    for i in range(n_r - 1):
        a = i * n_theta + np.arange(n_theta)
        b = i * n_theta + (np.arange(n_theta) + 1) % n_theta
        c = a + n_theta
        d = b + n_theta
        tris.append(np.stack([a, c, d], axis=1))   # CCW: inner, outer, outer-next
        tris.append(np.stack([a, d, b], axis=1))
    return xy.astype(np.float64), np.concatenate(tris).astype(np.int32)


def via_ring_resistance(length: float, drill: float = 0.3, plating: float = DEFAULT_THICKNESS,
                        conductivity: float = COPPER_CONDUCTIVITY, n_points: int = 16) -> float:
    """Per-resistor value of a via ring: hollow-cylinder R times the number of ring points.

    ``ViaSpec.compute_resistance`` (kicad.py:818-836) and the 1/N current split of
    ``process_via_spec`` (kicad.py:1546-1551).
    """
    ro = drill / 2 + plating
    ri = drill / 2
    area = math.pi * (ro * ro - ri * ri)
    return length / (conductivity * area) * n_points


@dataclass
class SyntheticSystem:
    """A meshed, indexed problem on flat arrays (what steps 4-6 of solve() produce).

    ``meshes``: list of (xy[n,2] f64, tri[t,3] i32, conductance, layer_index);
    lumped elements reference *global* unknown indices (VertexIndexer order:
    mesh after mesh, ``solver.py:221-229``; internal nodes after the vertices,
    ``solver.py:441-444``).
    """
    meshes: list
    n_internal: int = 0
    resistors: tuple = field(default_factory=lambda: (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)))
    current_sources: tuple = field(default_factory=lambda: (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)))
    voltage_sources: tuple = field(default_factory=lambda: (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)))
    ground: int = 0
    name: str = "synthetic"

    @property
    def n_vertices(self) -> int:
        if getattr(self, "_n_vertices", None) is not None:        # meshes live on the device (layered_system_on_device)
            return self._n_vertices
        return int(sum(m[0].shape[0] for m in self.meshes))

    @property
    def mesh_offsets(self) -> np.ndarray:
        if getattr(self, "_mesh_offsets", None) is not None:
            return self._mesh_offsets
        return np.concatenate([[0], np.cumsum([m[0].shape[0] for m in self.meshes])]).astype(np.int64)


def _nearest_vertex_on_grid(nx, ny, h, pts):
    """Nearest grid node of the *unjittered* lattice (connection snapping without a KD-tree)."""
    ix = np.clip(np.rint(pts[:, 0] / h), 0, nx - 1).astype(np.int64)
    iy = np.clip(np.rint(pts[:, 1] / h), 0, ny - 1).astype(np.int64)
    return iy * nx + ix


def layered_system(n_layers: int, nx: int, ny: int, *, h: float = DEFAULT_MAXIMUM_SIZE,
                   sigma: float = DEFAULT_SHEET_CONDUCTANCE, via_lattice: int = 32,
                   ring_points: int = 16, segment_length: float = 0.5,
                   current: float = 1.0, name: str | None = None, _no_meshes: bool = False) -> SyntheticSystem:
    """C2/C3/C4 of SURVEY.md section 8d.

    ``n_layers`` jittered nx*ny grids (seeds 0..n_layers-1); adjacent layers are
    stitched by ``via_lattice**2`` via rings of ``ring_points`` resistors; a 1 A
    source drives vertex (0.25,0.25)*L of layer 0 from (0.75,0.75)*L of the last
    layer; ground is vertex 0.
    """
    meshes = []
    for l in range(n_layers):
        if _no_meshes:
            break
        xy, tri = jittered_grid(nx, ny, h, seed=l)
        meshes.append((xy, tri, float(sigma), l))
    n_per = nx * ny
    Lx, Ly = (nx - 1) * h, (ny - 1) * h
    ra, rb, rr = [], [], []
    if n_layers > 1 and via_lattice > 0:
        cx = (np.arange(via_lattice) + 0.5) * Lx / via_lattice
        cy = (np.arange(via_lattice) + 0.5) * Ly / via_lattice
        CX, CY = np.meshgrid(cx, cy, indexing="xy")
        ang = np.arange(ring_points) * (2 * math.pi / ring_points)
        rad = 0.3 / 2
        px = (CX.reshape(-1, 1) + rad * np.cos(ang)[None, :]).reshape(-1)
        py = (CY.reshape(-1, 1) + rad * np.sin(ang)[None, :]).reshape(-1)
        snapped = _nearest_vertex_on_grid(nx, ny, h, np.stack([px, py], axis=1))
        rval = via_ring_resistance(segment_length, n_points=ring_points)
        for l in range(n_layers - 1):
            ra.append(snapped + l * n_per)
            rb.append(snapped + (l + 1) * n_per)
            rr.append(np.full(snapped.shape, rval))
    if ra:
        resistors = (np.concatenate(ra).astype(np.int64), np.concatenate(rb).astype(np.int64), np.concatenate(rr))
    else:
        resistors = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0))
    src = _nearest_vertex_on_grid(nx, ny, h, np.array([[0.25 * Lx, 0.25 * Ly]]))[0]
    snk = _nearest_vertex_on_grid(nx, ny, h, np.array([[0.75 * Lx, 0.75 * Ly]]))[0] + (n_layers - 1) * n_per
    cs = (np.array([src], np.int64), np.array([snk], np.int64), np.array([current]))
    return SyntheticSystem(meshes=meshes, resistors=resistors, current_sources=cs, ground=0,
                           name=name or f"{n_layers}-layer {nx}x{ny}")


def layered_system_on_device(ctx, n_layers: int, nx: int, ny: int, *, h: float = DEFAULT_MAXIMUM_SIZE,
                             sigma: float = DEFAULT_SHEET_CONDUCTANCE, name: str | None = None, **kw):
    """:func:`layered_system` with the meshes generated ON THE DEVICE (``padne_generate_grid_mesh``): the same vertices
    and triangles bit for bit, but no host array and no PCIe transfer -- at N = 10 M the numpy generation takes seconds
    and the upload is two thirds of the assembly.  Returns ``(system, xy_dev, tri_dev)``: ``system.meshes`` carries no
    arrays (``(None, None, sigma, layer)``), the lumped elements are the host-side index lists as before."""
    n_v, n_t = nx * ny, 2 * (nx - 1) * (ny - 1)
    xy = ctx.empty((n_layers * n_v, 2), np.float64)
    tri = ctx.empty((n_layers * n_t, 3), np.int32)
    for layer in range(n_layers):
        ctx.generate_grid_mesh(nx, ny, h, seed=layer, xy_out=xy, tri_out=tri, vertex_offset=layer * n_v,
                               tri_offset=layer * n_t)
    host = layered_system(n_layers, nx, ny, h=h, sigma=sigma, name=name, _no_meshes=True, **kw)
    host.meshes = [(None, None, float(sigma), layer) for layer in range(n_layers)]
    host._n_vertices = n_layers * n_v
    host._mesh_offsets = np.arange(n_layers + 1, dtype=np.int64) * n_v
    host._tri_offsets = np.arange(n_layers + 1, dtype=np.int64) * n_t
    return host, xy, tri


# named configurations of BASELINE.json
CONFIG_SHAPES = {"C2": (1, 1000, 1000, "C2 1-layer N=1M"), "C3": (4, 1118, 1118, "C3 4-layer N=5M"),
                 "C4": (8, 1118, 1118, "C4 8-layer N=10M"), "C5": (4, 1118, 1118, "C5 4-layer N=5M x 8 rhs")}


def config_on_device(ctx, name: str):
    """A named configuration with its meshes generated on the device: ``(system, xy_dev, tri_dev)``."""
    nl, nx, ny, label = CONFIG_SHAPES[name.upper()]
    return layered_system_on_device(ctx, nl, nx, ny, name=label)


def config(name: str) -> SyntheticSystem:
    name = name.upper()
    if name == "C2":
        return layered_system(1, 1000, 1000, name="C2 1-layer N=1M")
    if name == "C3":
        return layered_system(4, 1118, 1118, name="C3 4-layer N=5M")
    if name == "C4":
        return layered_system(8, 1118, 1118, name="C4 8-layer N=10M")
    if name == "C5":
        return layered_system(4, 1118, 1118, name="C5 4-layer N=5M x 8 rhs")
    raise KeyError(name)


def multi_rhs_pairs(system: SyntheticSystem, k: int = 8, seed: int = 5):
    """k distinct (source, sink) vertex pairs for the batched multi-RHS config C5."""
    rng = np.random.default_rng(seed)
    n = system.n_vertices
    pairs = rng.choice(n, size=(k, 2), replace=False)
    return pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
