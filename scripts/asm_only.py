"""Assembly of a named config from device-resident meshes, a few times (profiling target: rocprofv3 -- python3 scripts/asm_only.py C4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, name)
N = sysm.n_vertices + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
sig = np.array([m[2] for m in sysm.meshes])
for k in range(4):
    ctx.synchronize(); t = time.perf_counter()
    L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals); ctx.synchronize()
    print(f"assemble {name}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
    L.close()
# ... and the reduction of the assembled system to its potential block (index map resident on the device)
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
nv = sysm.n_vertices
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
dm = ctx.to_device(imap)
for k in range(3):
    ctx.synchronize(); t = time.perf_counter()
    A = L.reduce(dm, nv - 1, -1.0); ctx.synchronize()
    print(f"reduce {name}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
    A.close()
