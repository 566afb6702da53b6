"""A/B of two builds of the library on one box: warm wall time of the device-resident assembly of a config.
python scripts/ab_lib.py LIB.so [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from padne_amd import _hip
_hip.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
from padne_amd import synthetic
ctx = _hip.Context(0)
name = sys.argv[2] if len(sys.argv) > 2 else "C4"
sysm, xy, tri = synthetic.config_on_device(ctx, name)
N = sysm.n_vertices + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
sig = np.array([m[2] for m in sysm.meshes])
ts = []
for k in range(12):
    ctx.synchronize(); t0 = time.perf_counter()
    L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
    ctx.synchronize(); ts.append(time.perf_counter() - t0); L.close()
print(f"{sys.argv[1]}: assembly {name} warm wall min {min(ts[2:])*1e3:.3f} ms mean {np.mean(ts[2:])*1e3:.3f} ms")
