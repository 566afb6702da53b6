"""Timing of padne_nearest_vertex (connection snapping on the device) for a few point / query counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from padne_amd import _hip
ctx = _hip.Context(0)
rng = np.random.default_rng(0)
for n, m in ((500000, 200), (500000, 2000), (5000000, 200)):
    pts = rng.uniform(0, 400, (n, 2)); q = rng.uniform(0, 400, (m, 2))
    ctx.nearest_vertex(pts, q)
    t = time.perf_counter()
    for _ in range(5): ctx.nearest_vertex(pts, q)
    print(n, m, (time.perf_counter() - t) / 5 * 1e3, "ms per call", flush=True)
