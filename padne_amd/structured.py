"""Structured triangulations of simple shapes: a stand-in for the CGAL mesher in tests/benchmarks.

The reference meshes arbitrary polygons with CGAL (``mesh.py:662-795``, out of scope).  The
reference's synthetic end-to-end tests only need a rectangle (``tests/test_solver.py:461-595``) and
an annulus (``:597-751``); ``StructuredMesher.poly_to_mesh`` covers those two shapes so the same
tests can be replayed against the device path without CGAL or shapely.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import mesh, synthetic


@dataclass(frozen=True)
class Rect:
    x0: float
    y0: float
    x1: float
    y1: float

    def contains_xy(self, x, y) -> bool:
        return self.x0 <= x <= self.x1 and self.y0 <= y <= self.y1


@dataclass(frozen=True)
class Annulus:
    cx: float
    cy: float
    r_in: float
    r_out: float


@dataclass(frozen=True)
class Shapes:
    """Minimal MultiPolygon look-alike: ``.geoms`` is what ``problem.Layer`` reads."""
    geoms: tuple

    @classmethod
    def of(cls, *shapes) -> "Shapes":
        return cls(tuple(shapes))


class StructuredMesher:
    def __init__(self, config: mesh.Mesher.Config | None = None, jitter: float = 0.0, seed: int = 0):
        self.config = config if config is not None else mesh.Mesher.Config()
        self.jitter = jitter
        self.seed = seed

    def poly_to_mesh(self, poly, seed_points=()) -> mesh.Mesh:
        h = self.config.maximum_size or 0.6
        if isinstance(poly, Rect):
            nx = max(2, int(math.ceil((poly.x1 - poly.x0) / h)) + 1)
            ny = max(2, int(math.ceil((poly.y1 - poly.y0) / h)) + 1)
            xy, tri = synthetic.jittered_grid(nx, ny, 1.0, seed=self.seed, jitter=self.jitter)
            xy = xy.copy()
            xy[:, 0] = poly.x0 + xy[:, 0] * (poly.x1 - poly.x0) / (nx - 1)
            xy[:, 1] = poly.y0 + xy[:, 1] * (poly.y1 - poly.y0) / (ny - 1)
            return mesh.Mesh(xy, tri)
        if isinstance(poly, Annulus):
            n_r = max(2, int(math.ceil((poly.r_out - poly.r_in) / h)) + 1)
            n_t = max(8, int(math.ceil(2 * math.pi * poly.r_out / h)))
            xy, tri = synthetic.annulus_mesh(poly.r_in, poly.r_out, n_r, n_t)
            xy = xy + np.array([poly.cx, poly.cy])
            return mesh.Mesh(xy, tri)
        raise mesh.MeshingException(f"StructuredMesher cannot mesh {type(poly).__name__}")
