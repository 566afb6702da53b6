import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, synthetic
ctx = _hip.Context(0)
sysm = synthetic.config("C4"); nv = sysm.n_vertices; N = nv + 1
xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0); L.close()
keep = np.flatnonzero(imap[:nv] >= 0)
b = ctx.to_device(-rhs[keep]); x = ctx.empty(A.shape[0])
for i in range(4):
    t = time.perf_counter(); r = A.solve_spd_dev(b, x, precond="amg", rebuild=True); w = time.perf_counter() - t
    print(f"rebuild: wall {w*1e3:.1f} ms  setup {r.setup_seconds*1e3:.1f}  solve {r.seconds*1e3:.1f}", flush=True)
for i in range(2):
    t = time.perf_counter(); r = A.solve_spd_dev(b, x, precond="amg"); w = time.perf_counter() - t
    print(f"cached : wall {w*1e3:.1f} ms  setup {r.setup_seconds*1e3:.1f}  solve {r.seconds*1e3:.1f}", flush=True)
