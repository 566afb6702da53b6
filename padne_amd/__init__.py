"""padne_amd: MI355X-native implementation of the padne solver hot path.

``padne_amd.solver`` mirrors ``padne.solver`` (assemble + solve + post-process); the arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI of ``include/padne_hip.h``.
"""
from . import mesh, problem  # noqa: F401

__all__ = ["mesh", "problem", "solver", "synthetic", "structured", "reduction", "distributed"]
__version__ = "0.1.0"
