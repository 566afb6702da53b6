"""Load the hot-path modules of the *real* reference (atx/padne) in this container.

TEST INFRASTRUCTURE ONLY.  Nothing in ``padne_amd`` imports this file.  It only
works where ``/root/reference`` is mounted (the build container); the GPU box
has no reference, and nothing that runs there may call :func:`load_reference`.

What it does (SURVEY.md section 8c recipe): ``padne/__init__.py:2`` imports
``kicad`` which needs KiCad/pygerber/shapely, none of which exist here, so the
package cannot be imported normally.  The three files on the hot path
(``problem.py``, ``mesh.py``, ``solver.py``) only use shapely for type
annotations and for the geometric pre-pass that is out of scope, so we

* register inert stand-in *modules* for ``shapely`` / ``shapely.geometry`` /
  ``shapely.strtree`` and ``padne._cgal`` (any attribute resolves to a dummy
  class; nothing on the arithmetic path touches them),
* compile the three files from their source text **in memory** (nothing is
  written under /root/reference, no bytecode is produced, no source is copied
  into this repository),
* patch the one PEP-695 line (``mesh.py:190``, ``class IndexStore[T: ...]``)
  to the equivalent ``typing.Generic`` spelling, because this container runs
  Python 3.10.

The returned namespace gives the reference's own ``Mesh.from_triangle_soup``,
``HalfEdge.cotan``, ``laplace_operator``, ``stamp_network_into_system``,
``setup_ground_node``, ``solve_system``, ``compute_triangle_gradient``,
``compute_power_density``, ``produce_layer_solutions`` ... unmodified, which is
what ``tests/golden/make_golden.py`` runs to emit the golden vectors (the numpy
oracle is pinned against those: ``tests/test_oracle_golden.py``) and what
``tests/test_host_logic.py::test_problem_seam_with_the_reference_own_objects``
feeds to the product's host logic.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PADNE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "padne", "solver.py"))


class _Anything:
    """Dummy object: any attribute / call / subscript yields another dummy."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything

    def __call__(self, *a, **k):
        return _Anything()

    def __class_getitem__(cls, item):
        return cls


class _InertModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything


def _install_inert(name: str) -> types.ModuleType:
    mod = _InertModule(name)
    mod.__path__ = []  # behave as a package so "import a.b" works
    sys.modules[name] = mod
    return mod


def _exec_source(modname: str, relpath: str, package: str, patch=None) -> types.ModuleType:
    path = os.path.join(REFERENCE_ROOT, relpath)
    with open(path, "r", encoding="utf-8") as fh:
        text = fh.read()
    if patch is not None:
        text = patch(text)
    mod = types.ModuleType(modname)
    mod.__file__ = path
    mod.__package__ = package
    sys.modules[modname] = mod
    code = compile(text, path, "exec", dont_inherit=True)
    exec(code, mod.__dict__)
    return mod


def _patch_pep695(text: str) -> str:
    old = "class IndexStore[T: HasIndex]:"
    if old not in text:
        raise RuntimeError("reference mesh.py changed: PEP-695 line not found")
    new = ("import typing as _typing\n"
           "T = _typing.TypeVar('T')\n"
           "class IndexStore(_typing.Generic[T]):")
    return text.replace(old, new, 1)


_CACHE = None


def load_reference() -> types.SimpleNamespace:
    """Return ``SimpleNamespace(problem=..., mesh=..., solver=...)`` of the reference."""
    global _CACHE
    if _CACHE is not None:
        return _CACHE
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    import scipy.sparse.linalg  # noqa: F401  solver.py:773 relies on lazy submodule loading
    import scipy.spatial  # noqa: F401

    saved = {k: sys.modules.get(k) for k in
             ("shapely", "shapely.geometry", "shapely.strtree", "padne", "padne._cgal",
              "padne.problem", "padne.mesh", "padne.solver")}
    shp = _install_inert("shapely")
    shp.geometry = _install_inert("shapely.geometry")
    shp.strtree = _install_inert("shapely.strtree")
    pkg = types.ModuleType("padne")
    pkg.__path__ = []
    sys.modules["padne"] = pkg
    pkg._cgal = _install_inert("padne._cgal")
    try:
        pkg.problem = _exec_source("padne.problem", "padne/problem.py", "padne")
        pkg.mesh = _exec_source("padne.mesh", "padne/mesh.py", "padne", patch=_patch_pep695)
        pkg.solver = _exec_source("padne.solver", "padne/solver.py", "padne")
    finally:
        # keep "padne*" registered (dataclass machinery looks modules up by name)
        # but never leave a fake shapely behind for unrelated code
        for k in ("shapely", "shapely.geometry", "shapely.strtree"):
            if saved[k] is not None:
                sys.modules[k] = saved[k]
    _CACHE = types.SimpleNamespace(problem=pkg.problem, mesh=pkg.mesh, solver=pkg.solver)
    return _CACHE


class XY:
    """Duck-typed stand-in for ``shapely.geometry.Point`` (``solver.py:425`` reads .x/.y)."""

    def __init__(self, x: float, y: float):
        self.x = float(x)
        self.y = float(y)


class Geoms:
    """Duck-typed stand-in for a MultiPolygon (``problem.py:33`` reads .geoms)."""

    def __init__(self, n: int = 1):
        self.geoms = tuple(object() for _ in range(n))
