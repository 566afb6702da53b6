"""Warm wall time of the device-resident assembly of C4, call by call, with the first result still alive (what bench.py times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip, synthetic
ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, "C4")
N = sysm.n_vertices + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
sig = np.array([m[2] for m in sysm.meshes])
L0 = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals); ctx.synchronize()
ts = []
for k in range(12):
    ctx.synchronize(); t = time.perf_counter()
    L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals); ctx.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
    L.close()
print("assemble with the first result alive:", [round(t, 2) for t in ts])
