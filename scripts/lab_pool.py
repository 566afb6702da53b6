import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PADNE_VERBOSE"] = sys.argv[2] if len(sys.argv) > 2 else "pool"
from padne_amd import _hip
import numpy as np, bench
from padne_amd import synthetic
ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, sys.argv[1] if len(sys.argv) > 1 else "C4")
nv = sysm.n_vertices; N = nv + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
sig = np.array([m[2] for m in sysm.meshes])
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0)
keep = np.flatnonzero(imap[:nv] >= 0)
b = ctx.to_device(-rhs[keep]); x = ctx.empty(A.shape[0])
for k in range(4):
    print("---- solve", k, file=sys.stderr, flush=True)
    r = A.solve_spd_dev(b, x, precond="amg", rebuild=True)
    print("setup ms", r.setup_seconds * 1e3, "iterations", r.iterations, file=sys.stderr, flush=True)
