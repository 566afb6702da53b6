"""Input contract of the solver: layers, lumped-element networks, the Problem.

Mirrors the public surface of the reference's ``padne/problem.py:11-181`` (same
class names, field names, argument order and error behaviour) so that objects
built by padne's KiCad front-end can be handed over field by field, and so the
parity tests read like the reference's own.  There is no arithmetic here.

Differences that are deliberate:

* no dependency on shapely.  ``Layer.shape`` is any object with a ``.geoms``
  sequence (a shapely MultiPolygon qualifies); ``Connection.point`` is any
  object with ``.x`` / ``.y`` (a shapely Point qualifies, so does
  :class:`padne_amd.mesh.Point`).
* ``Network.nodes`` is numbered in first-appearance order of the terminals
  (element order, then terminal order) instead of the reference's
  ``list(set(...))`` (``problem.py:86-96``), whose order depends on object
  addresses.  The numbering is only used to order the *internal* unknowns
  (``solver.py:435-444``); any order is a valid permutation of the same system.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, ClassVar


@dataclass(frozen=True)
class Layer:
    """One copper layer.  ``conductance`` [S] = conductivity [S/mm] * thickness [mm]."""
    shape: Any
    name: str
    conductance: float
    geoms: tuple = field(init=False, repr=False)

    def __post_init__(self):
        object.__setattr__(self, "geoms", tuple(self.shape.geoms))


@dataclass(frozen=True, eq=False)
class NodeID:
    """Opaque identity of a network node (compared by identity, like the reference)."""


@dataclass(frozen=True)
class Connection:
    """Ties a network node to a point of a layer (snapped to the nearest mesh vertex)."""
    layer: Layer
    point: Any
    node_id: NodeID = field(default_factory=NodeID)


@dataclass(frozen=True)
class BaseLumped:
    _terminal_fields: ClassVar[tuple] = ()
    is_source: ClassVar[bool] = False
    extra_variable_count: ClassVar[int] = 0

    def __post_init__(self):
        assert self.terminals, "Lumped elements must have terminals"

    @property
    def terminals(self) -> list:
        return [getattr(self, name) for name in self._terminal_fields]


@dataclass(frozen=True)
class Resistor(BaseLumped):
    a: NodeID
    b: NodeID
    resistance: float
    _terminal_fields: ClassVar[tuple] = ("a", "b")

    def __post_init__(self):
        super().__post_init__()
        if self.resistance <= 0:
            raise ValueError(f"Resistance must be positive, got {self.resistance}")


@dataclass(frozen=True)
class VoltageSource(BaseLumped):
    p: NodeID
    n: NodeID
    voltage: float
    _terminal_fields: ClassVar[tuple] = ("p", "n")
    is_source: ClassVar[bool] = True
    extra_variable_count: ClassVar[int] = 1


@dataclass(frozen=True)
class CurrentSource(BaseLumped):
    f: NodeID
    t: NodeID
    current: float
    _terminal_fields: ClassVar[tuple] = ("f", "t")
    is_source: ClassVar[bool] = True


@dataclass(frozen=True)
class VoltageRegulator(BaseLumped):
    v_p: NodeID
    v_n: NodeID
    s_f: NodeID
    s_t: NodeID
    voltage: float
    gain: float
    _terminal_fields: ClassVar[tuple] = ("v_p", "v_n", "s_f", "s_t")
    is_source: ClassVar[bool] = True
    extra_variable_count: ClassVar[int] = 1


@dataclass(frozen=True)
class Network:
    connections: list
    elements: list
    nodes: dict = field(init=False)
    has_source: bool = field(init=False)

    def __post_init__(self):
        numbering: dict = {}
        for element in self.elements:
            for terminal in element.terminals:
                if not isinstance(terminal, NodeID):
                    raise TypeError("Terminal must be a NodeID")
                numbering.setdefault(terminal, len(numbering))
        object.__setattr__(self, "nodes", numbering)
        object.__setattr__(self, "has_source", any(e.is_source for e in self.elements))


@dataclass(frozen=True)
class Problem:
    layers: list
    networks: list
    project_name: str | None = None
