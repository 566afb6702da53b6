import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["PADNE_VERBOSE"] = "amg"
from padne_amd import _hip
if len(sys.argv) > 1: _hip.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, bench
from padne_amd import synthetic
ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, sys.argv[2] if len(sys.argv) > 2 else "C2")
nv = sysm.n_vertices; N = nv + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
sig = np.array([m[2] for m in sysm.meshes])
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0)
keep = np.flatnonzero(imap[:nv] >= 0)
b = ctx.to_device(-rhs[keep]); x = ctx.empty(A.shape[0])
for k in range(3):
    try:
        r = A.solve_spd_dev(b, x, precond="amg", rebuild=True, max_iter=30, raise_on_fail=False)
    except Exception as e:
        print("solve:", e)
